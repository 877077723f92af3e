"""Batch sources for the train / test loops.  The reference feeds the hot path from its LMDB `ImageReader`
(UNet/imagereader.py:77-355), which this build keeps out of scope; what the hot path needs from a reader is
only its OUTPUT CONTRACT (UNet/imagereader.py:298-312,353-355): images fp32 [C,H,W] z-scored per channel, labels int32
one-hot [H,W,K].  Two small sources implement that contract here so the CLIs run without lmdb:

  SyntheticReader  seeded N(0,1) tiles + piecewise-constant 8x8-block labels (the benchmark workload);
  TileFolderReader a folder of `<name>.npy` image tiles ([H,W] or [H,W,C], any dtype) with `<name>_mask.npy` class maps.

Both expose the reader surface train.py uses: startup(), shutdown(), get_image_count(), get_image_size() -> (H, W, C),
batches(batch_size) -> iterator of (images [B,C,H,W] fp32, labels [B,H,W,K] int32) that never ends (like the reference's
generator, UNet/imagereader.py:338-343).  Batches are pinned host tensors so the H2D copy is asynchronous.
`batches(batch_size, classmap=True, pin=False)` yields the uint8 class map [B,H,W] instead of the one-hot, unpinned: the
form `feed.DeviceFeed` stages itself and expands to the same one-hot on the device.

A reader with the reference's OWN surface -- per-sample `generator()` (UNet/imagereader.py:338-355), e.g. the kept LMDB
`ImageReader` -- plugs in through `from_sample_generator()` / directly as `train_model(train_reader=...)`.
"""
import os
import threading

import numpy as np
import torch


def zscore_normalize(chw):
    """Per-channel z-score; only mean-subtract when std <= 1 (reference UNet/imagereader.py:33-49)."""
    out = np.asarray(chw, dtype=np.float32).copy()
    for c in range(out.shape[0]):
        std, mean = float(np.std(out[c])), float(np.mean(out[c]))
        out[c] = (out[c] - mean) if std <= 1.0 else (out[c] - mean) / std
    return out


def one_hot(mask_hw, number_classes):
    m = np.asarray(mask_hw).astype(np.int64)
    if m.max(initial=0) >= number_classes or m.min(initial=0) < 0:
        raise IndexError("Number of classes specified differs from number of observed classes in data")
    return (m[..., None] == np.arange(number_classes)).astype(np.int32)


def _pin(t):
    return t.pin_memory() if torch.cuda.is_available() else t


class _Base:
    def startup(self):
        pass

    def shutdown(self):
        pass


class SyntheticReader(_Base):
    """`count` seeded N(0,1) tiles with block-random class maps, generated once (a dataset in memory, like a reader whose files sit in the
    page cache); every worker draws its own seeded index stream from it.  (Up to round 4 every batch was freshly drawn: ~50 ms of host RNG
    per 8 x 3 x 512 x 512 batch -- the benchmark of the feed measured torch.randn.)"""

    def __init__(self, count, height, width, channels, number_classes, seed=0):
        self.count, self.h, self.w, self.c, self.k, self.seed = count, height, width, channels, number_classes, seed
        self._pool = None
        self._lock = threading.Lock()

    def get_image_count(self):
        return self.count

    def get_image_size(self):
        return (self.h, self.w, self.c)

    def _tiles(self):
        with self._lock:
            if self._pool is None:
                g = torch.Generator().manual_seed(self.seed * 1000003 + 17)
                n = max(1, min(self.count, 256))
                img = torch.randn(n, self.c, self.h, self.w, generator=g)
                cls = torch.randint(0, self.k, (n, (self.h + 7) // 8, (self.w + 7) // 8), generator=g)
                cls = cls.repeat_interleave(8, 1).repeat_interleave(8, 2)[:, :self.h, :self.w].to(torch.uint8).contiguous()
                self._pool = (img, cls)
            return self._pool

    def batches(self, batch_size, classmap=False, pin=True, raw=False, worker=0, num_workers=1):
        # (raw: the synthetic tiles are N(0,1) either way; every worker draws its own index stream)
        g = torch.Generator().manual_seed(self.seed * 1000003 + worker)
        pin_ = _pin if pin else (lambda t: t)
        img_all, cls_all = self._tiles()
        while True:
            idx = torch.randint(0, img_all.shape[0], (batch_size,), generator=g)
            img, cls = img_all.index_select(0, idx), cls_all.index_select(0, idx)
            if classmap:
                yield pin_(img), pin_(cls)
            else:
                yield pin_(img), pin_(torch.nn.functional.one_hot(cls.long(), self.k).to(torch.int32))


class TileFolderReader(_Base):
    """Key selection follows the reference reader (UNet/imagereader.py:209-243):
      * shuffle=True: every sample is an independent uniform draw WITH replacement from the key list; with
        balance_classes a class is drawn uniformly from range(number_classes) (re-drawn while it has no example), then a
        uniform example among the tiles whose mask contains that class (:211-233; the reference reads the class list from the
        LMDB key suffix written by build_lmdb.py:123,178 -- here it is computed from the mask file);
      * shuffle=False: worker `w` of `num_workers` walks keys w, w+num_workers, ... modulo the key count (:239-241) and
        class balancing is ignored ("without shuffle you cannot balance classes", :238)."""

    def __init__(self, folder, number_classes, shuffle=False, seed=0, balance_classes=False):
        self.folder, self.k, self.shuffle, self.seed = folder, number_classes, shuffle, seed
        self.balance_classes = bool(balance_classes)
        self.names = sorted(f[:-4] for f in os.listdir(folder) if f.endswith(".npy") and not f.endswith("_mask.npy"))
        if not self.names:
            raise IOError("no <name>.npy tiles in " + folder)
        first = np.load(os.path.join(folder, self.names[0] + ".npy"))
        self.h, self.w = first.shape[:2]
        self.c = 1 if first.ndim == 2 else first.shape[2]
        if self.h % 16 or self.w % 16:
            raise IOError("Input Image tile size must be a multiple of 16")     # cf. UNet/imagereader.py:136-139
        self.keys = [[]]                                                         # per class: indices of the tiles containing it
        if self.balance_classes:
            for i, name in enumerate(self.names):
                for cls in np.unique(np.load(os.path.join(folder, name + "_mask.npy"))):
                    while len(self.keys) <= int(cls):
                        self.keys.append([])
                    self.keys[int(cls)].append(i)

    def get_image_count(self):
        return len(self.names)

    def get_image_size(self):
        return (self.h, self.w, self.c)

    def _load(self, name, classmap=False, raw=False):
        im = np.load(os.path.join(self.folder, name + ".npy"))
        im = im[..., None] if im.ndim == 2 else im
        mk = np.load(os.path.join(self.folder, name + "_mask.npy"))
        if classmap:
            if mk.max(initial=0) >= self.k or mk.min(initial=0) < 0:
                raise IndexError("Number of classes specified differs from number of observed classes in data")
            chw = im.transpose(2, 0, 1)
            return (np.ascontiguousarray(chw, dtype=np.float32) if raw else zscore_normalize(chw)), mk.astype(np.uint8)
        return zscore_normalize(im.transpose(2, 0, 1)), one_hot(mk, self.k)

    def key_sequence(self, worker=0, num_workers=1):
        """Infinite iterator of tile indices for one reader worker."""
        rng = np.random.default_rng([self.seed, worker])
        pos = worker % len(self.names)
        while True:
            if self.shuffle and self.balance_classes:
                while True:
                    label = int(rng.integers(0, self.k))
                    if label >= len(self.keys):
                        raise IndexError("Number of classes specified differs from number of observed classes in data")
                    if self.keys[label]:
                        break
                yield self.keys[label][int(rng.integers(0, len(self.keys[label])))]
            elif self.shuffle:
                yield int(rng.integers(0, len(self.names)))
            else:
                yield pos
                pos = (pos + num_workers) % len(self.names)

    def batches(self, batch_size, classmap=False, pin=True, raw=False, worker=0, num_workers=1):
        """raw=True (with classmap=True): un-normalised pixel values, for the device pipeline that augments before z-scoring"""
        pin_ = _pin if pin else (lambda t: t)
        keys = self.key_sequence(worker, num_workers)
        while True:
            imgs, labs = [], []
            for _ in range(batch_size):
                i, l = self._load(self.names[next(keys)], classmap, raw)
                imgs.append(i); labs.append(l)
            yield pin_(torch.as_tensor(np.stack(imgs))), pin_(torch.as_tensor(np.stack(labs)))


class SampleGeneratorReader(_Base):
    """Adapter for a reader with the REFERENCE's surface (UNet/imagereader.py:87-355): `startup()`, `shutdown()`,
    `get_image_size()` -> [H, W, C], `get_image_count()`, and `generator()` yielding one sample at a time as
    `(image fp32 [C,H,W] z-scored, label int32 one-hot [H,W,K])` until the reader is shut down (:338-343) -- what the reference wraps
    in `tf.data.Dataset.from_generator(...).batch(global_batch)` (:348-355, UNet/train.py:84-90).  `batches()` does that batching.

    Such a reader owns its workers (the reference forks `num_workers` processes that share one output queue), shuffling,
    class balancing and augmentation (constructor arguments, UNet/train.py:66-76): every `batches()` iterator of this adapter
    draws from that ONE sample stream under a lock (`worker` / `num_workers` only say how many consumers there are), and
    `augments_itself` tells the train loop not to augment a second time on the device.  `classmap=True` hands the labels over as
    the uint8 class map (`argmax` of an exact one-hot) for the device feed's on-device expansion."""
    augments_itself = True

    def __init__(self, reader):
        import threading
        self.reader = reader
        self.balance_classes = bool(getattr(reader, "balance_classes", False))    # what the wrapped reader was built with (UNet/imagereader.py:89-103)
        self._lock = threading.Lock()
        self._gen = None

    def startup(self):
        self.reader.startup()

    def shutdown(self):
        self.reader.shutdown()

    def get_image_count(self):
        return self.reader.get_image_count()

    def get_image_size(self):
        return tuple(self.reader.get_image_size())

    def _next_sample(self):
        with self._lock:
            if self._gen is None:
                self._gen = self.reader.generator()
            return next(self._gen)              # StopIteration once the reader was shut down

    def batches(self, batch_size, classmap=False, pin=True, raw=False, worker=0, num_workers=1):
        if raw:
            raise ValueError("a sample-generator reader normalises (and augments) inside its own workers: raw tiles are not available")
        pin_ = _pin if pin else (lambda t: t)
        h, w, c = self.get_image_size()
        while True:
            imgs, labs = [], []
            try:
                for _ in range(batch_size):
                    img, lab = self._next_sample()
                    img, lab = np.asarray(img), np.asarray(lab)
                    if img.dtype != np.float32 or img.shape != (c, h, w):
                        raise IOError("reader sample: expected a float32 image [C,H,W] = {}, got {} {}".format((c, h, w), img.dtype, img.shape))
                    if lab.dtype != np.int32 or lab.ndim != 3 or lab.shape[:2] != (h, w):
                        raise IOError("reader sample: expected an int32 one-hot label [H,W,K], got {} {}".format(lab.dtype, lab.shape))
                    imgs.append(img)
                    if classmap:
                        # the class map is the argmax of an EXACT one-hot: an all-zero or multi-hot pixel would silently become class 0 / the
                        # first of the set classes, and more than 256 classes do not fit the uint8 map
                        if lab.shape[2] > 256:
                            raise ValueError("the uint8 class-map hand-over holds at most 256 classes ({} given)".format(lab.shape[2]))
                        if lab.min() < 0 or lab.max() > 1 or not (lab.sum(-1) == 1).all():
                            raise ValueError("reader sample: the label is not an exact one-hot (every pixel needs exactly one class set)")
                        lab = lab.argmax(-1).astype(np.uint8)
                    labs.append(lab)
            except StopIteration:
                if not imgs:
                    return
            # (a final short batch is handed on, like Dataset.batch() without drop_remainder)
            yield pin_(torch.as_tensor(np.stack(imgs))), pin_(torch.as_tensor(np.stack(labs)))
            if len(imgs) < batch_size:
                return


def from_sample_generator(reader):
    """-> a batch source for train_model / DeviceFeed from a reference-style per-sample reader (see SampleGeneratorReader)."""
    return SampleGeneratorReader(reader)


def as_batch_reader(reader):
    """train_model accepts either this build's batch readers or a reference-style reader object (generator(), no batches())."""
    if not hasattr(reader, "batches") and hasattr(reader, "generator"):
        return from_sample_generator(reader)
    return reader


def round_robin(iterators):
    """One stream from several worker iterators (the synchronous hand-over's stand-in for the reference's shared queue)."""
    while True:
        for it in iterators:
            yield next(it)
