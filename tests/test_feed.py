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
