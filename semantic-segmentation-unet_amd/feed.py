"""Device feed for the train / test loops: reader batches -> pinned staging ring -> asynchronous H2D on a copy stream.

The reference hands the hot path host batches through `tf.data.Dataset.from_generator(...).prefetch(...)` and
`experimental_distribute_dataset` (UNet/train.py:63-90): the framework stages and copies them while the previous step
runs.  This is that stage for the HIP engine:

  * a worker thread pulls batches from the reader generator (so z-scoring / decoding overlaps the GPU step), writes them
    into a ring of PINNED staging tensors and issues the H2D copies on a dedicated stream;
  * `next()` makes the caller's stream wait on the copy's event - no host synchronisation - and hands out device tensors
    that stay valid until the next-but-(depth-2) call (a slot is recycled only after an event recorded behind its last
    consumer kernel has completed);
  * label transport: either the reader's int32 one-hot [B,H,W,K] (the reference's contract, UNet/imagereader.py:302-312),
    or - `classmap=True` - the uint8 class map [B,H,W], expanded to the same one-hot on the device by
    `unet_labels_onehot` (4K x fewer bytes over PCIe, and no host-side one-hot); both give bit-identical device tensors.

`batches` may be a LIST of iterators (e.g. one seeded reader per worker, like the reference's `reader_count` processes,
UNet/imagereader.py:182-186): each gets its own staging thread - tensor ops release the GIL - and batches are handed out in
completion order.  On a machine without a GPU (`device.type == "cpu"`) the class degrades to plain prefetching threads so
the host logic is testable.
"""
import atexit
import ctypes
import queue
import threading

import torch


class DeviceFeed:
    def __init__(self, batches, device, depth=3, classmap=False, number_classes=None, onehot=True):
        assert depth >= 2
        self.dev = torch.device(device)
        self.cuda = self.dev.type == "cuda"
        self.depth, self.classmap, self.k = depth, classmap, number_classes
        self.onehot = onehot                      # False (classmap feeds only): hand out the uint8 class map itself
        if classmap:
            assert number_classes is not None and 0 < number_classes <= 256
        self._its = [iter(b) for b in batches] if isinstance(batches, (list, tuple)) else [iter(batches)]
        depth = max(depth, len(self._its) + 1)
        self.depth = depth
        self._live = len(self._its)
        self._lock = threading.Lock()
        self._ready = queue.Queue()
        self._free = queue.Queue()
        for s in range(depth):
            self._free.put(s)
        self._slots = [None] * depth              # per slot: dict(pin_img, pin_lab, dev_img, dev_lab, dev_onehot, ready, done)
        self._held = None
        self._stop = False
        self._err = None
        if self.cuda:
            self._copy = torch.cuda.Stream(device=self.dev)
            from . import _lib
            self._L = _lib.lib()                  # fails loudly when the HIP library is missing
            self._bad = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self._threads = [threading.Thread(target=self._work, args=(it,), daemon=True) for it in self._its]
        for t in self._threads:
            t.start()
        atexit.register(self.close)

    # ---- producer ------------------------------------------------------------------------------------------------------
    def _slot(self, s, img, lab):
        sl = self._slots[s]
        if sl is None or sl["pin_img"].shape != img.shape or sl["pin_lab"].shape != lab.shape or sl["pin_lab"].dtype != lab.dtype:
            sl = {"pin_img": torch.empty(img.shape, dtype=torch.float32), "pin_lab": torch.empty(lab.shape, dtype=lab.dtype)}
            if self.cuda:
                sl["pin_img"] = sl["pin_img"].pin_memory(); sl["pin_lab"] = sl["pin_lab"].pin_memory()
                sl["dev_img"] = torch.empty(img.shape, dtype=torch.float32, device=self.dev)
                sl["dev_lab"] = torch.empty(lab.shape, dtype=lab.dtype, device=self.dev)
                if self.classmap and self.onehot:
                    sl["dev_onehot"] = torch.empty(tuple(lab.shape) + (self.k,), dtype=torch.int32, device=self.dev)
                sl["ready"] = torch.cuda.Event(); sl["done"] = None
            self._slots[s] = sl
        return sl

    def _work(self, it):
        try:
            if self.cuda:
                torch.cuda.set_device(self.dev)
            while not self._stop:
                s = self._free.get()
                if s is None:
                    return
                try:
                    img, lab = next(it)
                except StopIteration:
                    self._free.put(s)
                    with self._lock:
                        self._live -= 1
                        last = self._live == 0
                    if last:
                        self._ready.put(None)
                    return
                img = torch.as_tensor(img, dtype=torch.float32)
                lab = torch.as_tensor(lab)
                if self.classmap:
                    assert lab.dtype == torch.uint8 and lab.dim() == 3, "classmap feed expects uint8 [B,H,W] labels"
                else:
                    assert lab.dtype == torch.int32 and lab.dim() == 4, "feed expects int32 one-hot [B,H,W,K] labels"
                sl = self._slot(s, img, lab)
                if self.cuda and sl["done"] is not None:
                    sl["done"].synchronize()      # the last kernels that read this slot's device tensors have finished
                sl["pin_img"].copy_(img); sl["pin_lab"].copy_(lab)
                if self.cuda:
                    with torch.cuda.stream(self._copy):
                        sl["dev_img"].copy_(sl["pin_img"], non_blocking=True)
                        sl["dev_lab"].copy_(sl["pin_lab"], non_blocking=True)
                        if self.classmap and self.onehot:
                            n = sl["dev_lab"].numel()
                            self._L.unet_labels_onehot(ctypes.c_void_p(sl["dev_lab"].data_ptr()), ctypes.c_void_p(sl["dev_onehot"].data_ptr()),
                                                       n, self.k, ctypes.c_void_p(self._bad.data_ptr()),
                                                       ctypes.c_void_p(self._copy.cuda_stream))
                        sl["ready"].record(self._copy)
                self._ready.put(s)
        except BaseException as e:                # surfaced on the consumer side
            self._err = e
            self._ready.put(None)

    # ---- consumer ------------------------------------------------------------------------------------------------------
    def __iter__(self):
        return self

    def __next__(self):
        if self._held is not None:                # everything enqueued so far on the caller's stream may read the held slot
            sl = self._slots[self._held]
            if self.cuda:
                sl["done"] = torch.cuda.Event(); sl["done"].record(torch.cuda.current_stream(self.dev))
            self._free.put(self._held)
            self._held = None
        s = self._ready.get()
        if s is None:
            if self._err is not None:
                raise self._err
            raise StopIteration
        self._held = s
        sl = self._slots[s]
        if not self.cuda:
            lab = sl["pin_lab"]
            if self.classmap and self.onehot:
                if int(lab.max()) >= self.k:
                    raise IndexError("Number of classes specified differs from number of observed classes in data")
                lab = torch.nn.functional.one_hot(lab.long(), self.k).to(torch.int32)
            return sl["pin_img"].clone(), lab.clone()
        torch.cuda.current_stream(self.dev).wait_event(sl["ready"])
        return sl["dev_img"], (sl["dev_onehot"] if self.classmap and self.onehot else sl["dev_lab"])

    def out_of_range_labels(self):
        """Number of class-map pixels >= number_classes seen so far (host sync; the reference raises IndexError per batch)."""
        return int(self._bad.item()) if self.cuda and self.classmap else 0

    def close(self):
        """Stop and join the worker (a thread still inside torch at interpreter exit aborts the process)."""
        if not self._threads:
            return
        self._stop = True
        for _ in self._threads:
            self._free.put(None)
        for t in self._threads:
            t.join(timeout=30)
        self._threads = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
