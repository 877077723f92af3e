"""Device feed: host logic on CPU, bit-identity of the device tensors on the GPU."""
import numpy as np
import pytest
import torch

from conftest import pkg

readers = pkg("readers")
feed = pkg("feed")


def _take(it, n):
    return [next(it) for _ in range(n)]


def test_feed_cpu_matches_reader_onehot_and_classmap():
    # same seeded reader three ways: direct one-hot batches, feed(one-hot), feed(class map -> one-hot): identical sequences
    mk = lambda: readers.SyntheticReader(8, 32, 48, 3, 4, seed=5)
    direct = _take(mk().batches(2, pin=False), 5)
    f1 = feed.DeviceFeed(mk().batches(2, pin=False), "cpu", depth=2)
    f2 = feed.DeviceFeed(mk().batches(2, classmap=True, pin=False), "cpu", depth=3, classmap=True, number_classes=4)
    for (i0, l0), (i1, l1), (i2, l2) in zip(direct, _take(f1, 5), _take(f2, 5)):
        assert torch.equal(i0, i1) and torch.equal(i0, i2)
        assert l1.dtype == torch.int32 and torch.equal(l0, l1) and torch.equal(l0, l2)
    f1.close(); f2.close()


def test_feed_cpu_finite_iterator_and_errors():
    def gen():
        for k in range(3):
            yield torch.full((1, 1, 16, 16), float(k)), torch.zeros(1, 16, 16, 2, dtype=torch.int32)
    f = feed.DeviceFeed(gen(), "cpu")
    assert [float(i[0, 0, 0, 0]) for i, _ in f] == [0.0, 1.0, 2.0]

    def bad():
        yield torch.zeros(1, 1, 16, 16), torch.zeros(1, 16, 16, 2, dtype=torch.int64)       # wrong label dtype
    with pytest.raises(AssertionError):
        next(feed.DeviceFeed(bad(), "cpu"))
    cm = feed.DeviceFeed(iter([(torch.zeros(1, 1, 16, 16), torch.full((1, 16, 16), 3, dtype=torch.uint8))]), "cpu",
                         classmap=True, number_classes=2)
    with pytest.raises(IndexError):              # the reader contract's error, UNet/imagereader.py:302-312
        next(cm)


def test_feed_cpu_several_reader_threads():
    # three finite sources drained by three staging threads: every batch arrives exactly once, then StopIteration
    def gen(base):
        for k in range(4):
            yield torch.full((1, 1, 16, 16), float(base + k)), torch.zeros(1, 16, 16, 2, dtype=torch.int32)
    f = feed.DeviceFeed([gen(0), gen(100), gen(200)], "cpu")
    got = sorted(float(i[0, 0, 0, 0]) for i, _ in f)
    assert got == sorted([b + k for b in (0, 100, 200) for k in range(4)])
    f.close()


def test_tile_folder_reader_classmap(tmp_path):
    rng = np.random.default_rng(0)
    for k in range(3):
        np.save(tmp_path / ("t%d.npy" % k), rng.integers(0, 4000, (32, 32)).astype(np.uint16))
        np.save(tmp_path / ("t%d_mask.npy" % k), rng.integers(0, 2, (32, 32)).astype(np.uint8))
    a = next(readers.TileFolderReader(str(tmp_path), 2).batches(3, pin=False))
    b = next(readers.TileFolderReader(str(tmp_path), 2).batches(3, classmap=True, pin=False))
    assert torch.equal(a[0], b[0]) and b[1].dtype == torch.uint8
    assert torch.equal(a[1], torch.nn.functional.one_hot(b[1].long(), 2).to(torch.int32))


@pytest.mark.gpu
def test_feed_gpu_bit_identical_and_slot_reuse():
    dev = torch.device("cuda", 0)
    mk = lambda: readers.SyntheticReader(8, 64, 64, 1, 3, seed=9)
    direct = _take(mk().batches(2, pin=False), 7)
    f = feed.DeviceFeed(mk().batches(2, classmap=True, pin=False), dev, depth=3, classmap=True, number_classes=3)
    for i0, l0 in direct:                        # 7 batches through 3 slots: every slot is recycled at least once
        i1, l1 = next(f)
        s = (i1 * 2.0).sum()                     # a consumer kernel on the caller's stream
        assert i1.device == dev and l1.dtype == torch.int32 and tuple(l1.shape) == (2, 64, 64, 3)
        assert torch.equal(i1.cpu(), i0) and torch.equal(l1.cpu(), l0) and torch.isfinite(s)
    assert f.out_of_range_labels() == 0
    f.close()
    g = feed.DeviceFeed(iter([(torch.zeros(1, 1, 16, 16), torch.full((1, 16, 16), 5, dtype=torch.uint8))]), dev,
                        classmap=True, number_classes=3)
    next(g)
    assert g.out_of_range_labels() == 256


class FakeReferenceReader:
    """Exactly the surface of the reference's ImageReader that train.py touches (UNet/imagereader.py:87,155-207,338-343; UNet/train.py:66-90):
    startup / shutdown / get_image_size / get_image_count / generator -- one sample per `next`: (float32 [C,H,W] z-scored image,
    int32 one-hot [H,W,K] label), ending after shutdown()."""

    def __init__(self, count, h, w, c, k, seed=0, limit=None):
        self.image_size = [h, w, c]
        self.count, self.k, self.seed, self.limit = count, k, seed, limit
        self.started = self.stopped = False
        self.handed_out = 0

    def startup(self):
        self.started = True

    def shutdown(self):
        self.stopped = True

    def get_image_size(self):
        return self.image_size

    def get_image_count(self):
        return self.count

    def sample(self, i):
        h, w, c = self.image_size
        rng = np.random.default_rng([self.seed, i])
        img = readers.zscore_normalize(rng.normal(3.0, 2.0, (c, h, w)).astype(np.float32))
        cls = rng.integers(0, self.k, (h, w))
        return img, (cls[..., None] == np.arange(self.k)).astype(np.int32)

    def generator(self):
        assert self.started
        while not self.stopped and (self.limit is None or self.handed_out < self.limit):
            self.handed_out += 1
            yield self.sample(self.handed_out - 1)


def test_reference_style_sample_reader_batches_like_dataset_batch():
    # Dataset.from_generator(reader.generator).batch(G) (UNet/imagereader.py:348-355, UNet/train.py:84-85): consecutive samples stacked,
    # a final short batch handed on, end of stream after the reader stops
    ref = FakeReferenceReader(10, 16, 32, 2, 3, seed=4, limit=7)
    rd = readers.from_sample_generator(ref)
    rd.startup()
    assert rd.get_image_size() == (16, 32, 2) and rd.get_image_count() == 10
    got = list(rd.batches(3, pin=False))
    assert [tuple(i.shape) for i, _ in got] == [(3, 2, 16, 32), (3, 2, 16, 32), (1, 2, 16, 32)]
    assert got[0][0].dtype == torch.float32 and got[0][1].dtype == torch.int32 and tuple(got[0][1].shape) == (3, 16, 32, 3)
    flat_i = torch.cat([i for i, _ in got]); flat_l = torch.cat([l for _, l in got])
    for s in range(7):
        img, lab = ref.sample(s)
        assert np.array_equal(flat_i[s].numpy(), img) and np.array_equal(flat_l[s].numpy(), lab)
    # class-map transport for the device feed: argmax of the one-hot, expanded back to the same one-hot by the feed
    ref2 = FakeReferenceReader(10, 16, 32, 2, 3, seed=4, limit=6)
    rd2 = readers.from_sample_generator(ref2); rd2.startup()
    f = feed.DeviceFeed(rd2.batches(2, classmap=True, pin=False), "cpu", classmap=True, number_classes=3)
    out = list(f)
    assert len(out) == 3
    for b, (i, l) in enumerate(out):
        for j in range(2):
            img, lab = ref2.sample(2 * b + j)
            assert np.array_equal(i[j].numpy(), img) and np.array_equal(l[j].numpy(), lab)
    f.close()
    with pytest.raises(ValueError):
        next(rd2.batches(2, classmap=True, raw=True))
    # several consumers share the reader's ONE sample stream: every sample exactly once
    ref3 = FakeReferenceReader(10, 16, 16, 1, 2, seed=1, limit=12)
    rd3 = readers.from_sample_generator(ref3); rd3.startup()
    its = [rd3.batches(2, pin=False, worker=w, num_workers=3) for w in range(3)]
    seen = []
    for it in its:
        seen += [i for i, _ in [next(it), next(it)]]
    want = sorted(float(ref3.sample(s)[0][0, 3, 5]) for s in range(12))
    assert sorted(float(x[j][0, 3, 5]) for x in seen for j in range(2)) == want
    # a sample that breaks the reader contract is refused, not silently cast
    class Bad(FakeReferenceReader):
        def sample(self, i):
            img, lab = super().sample(i)
            return img.astype(np.float64), lab
    bad = readers.from_sample_generator(Bad(4, 16, 16, 1, 2)); bad.startup()
    with pytest.raises(IOError):
        next(bad.batches(2, pin=False))
    # the uint8 class-map hand-over is the argmax of an EXACT one-hot: an all-zero or multi-hot pixel is refused (it would otherwise become
    # class 0 without anybody noticing); the same sample passes on the one-hot path
    class NotOneHot(FakeReferenceReader):
        def sample(self, i):
            img, lab = super().sample(i)
            lab[3, 5, :] = 0
            return img, lab
    noh = readers.from_sample_generator(NotOneHot(4, 16, 16, 1, 3)); noh.startup()
    with pytest.raises(ValueError):
        next(noh.batches(2, classmap=True, pin=False))
    ok = readers.from_sample_generator(NotOneHot(4, 16, 16, 1, 3)); ok.startup()
    assert tuple(next(ok.batches(2, pin=False))[1].shape) == (2, 16, 16, 3)
    # the adapter reports the wrapped reader's own class-balancing setting (UNet/imagereader.py:89-103), not a constant
    plain = FakeReferenceReader(4, 16, 16, 1, 2); assert readers.from_sample_generator(plain).balance_classes is False
    plain.balance_classes = True; assert readers.from_sample_generator(plain).balance_classes is True


@pytest.mark.gpu
def test_reference_style_sample_reader_through_the_device_feed():
    dev = torch.device("cuda", 0)
    ref = FakeReferenceReader(10, 32, 48, 3, 4, seed=2, limit=10)
    rd = readers.from_sample_generator(ref); rd.startup()
    f = feed.DeviceFeed(rd.batches(2, classmap=True, pin=False), dev, classmap=True, number_classes=4)
    n = 0
    for i, l in f:
        assert i.device == dev and l.dtype == torch.int32 and tuple(l.shape) == (2, 32, 48, 4)
        for j in range(2):
            img, lab = ref.sample(2 * n + j)
            assert np.array_equal(i[j].cpu().numpy(), img) and np.array_equal(l[j].cpu().numpy(), lab)
        n += 1
    assert n == 5 and f.out_of_range_labels() == 0
    f.close()
