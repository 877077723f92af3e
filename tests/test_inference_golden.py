"""The inference driver's index work against fixtures made by the REFERENCE's own functions (tests/golden/make_inference_golden.py ran
UNet/inference.py:_inference_tiling, :_inference and UNet/imagereader.py:zscore_normalize in the build container): reflect padding to a
multiple of 16, zones of responsibility and halos of the tiled path (including the reference's far-halo behaviour at the image edge),
crops and pastes -- BIT-EXACT, with the same deterministic stand-in network on both sides (tests/fake_segmenter.py).  This pins SURVEY
8(f) rank 2's integer work, not the network arithmetic."""
import os

import numpy as np
import pytest

import fake_segmenter as fs
from conftest import pkg

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "inference_driver.npz"))


@pytest.mark.parametrize("case", fs.INFERENCE_CASES, ids=["%dx%dx%d r%d" % c[:4] for c in fs.INFERENCE_CASES])
def test_inference_functions_reproduce_the_reference_masks_bit_exactly(case):
    inf = pkg("inference")
    h, w, c, radius, seed = case
    key = "%dx%dx%d_r%d_s%d" % case
    img = fs.synthetic_image(h, w, c, seed)
    fake = fs.FakeSegmenter(radius)
    tiled = inf._inference_tiling(img.copy(), fake, inf.TILE_SIZE, predict=fs.predict_with(fake))
    # the same tiles, in the same order, as the reference handed to its model (N, C, H, W)
    assert np.array_equal(np.array(fake.calls, dtype=np.int32), G["calls_" + key])
    whole = inf._inference(img.copy(), fs.FakeSegmenter(radius), predict=fs.predict_with(fs.FakeSegmenter(radius)))
    for name, m in (("tiled_", tiled), ("whole_", whole)):
        assert m.dtype == np.int32 and m.shape == (h, w)
        rows, cols = fs.mask_digest(m)
        bad_r = np.nonzero(rows != G[name + "rows_" + key])[0]; bad_c = np.nonzero(cols != G[name + "cols_" + key])[0]
        assert bad_r.size == 0 and bad_c.size == 0, (name, "rows", bad_r[:8], "columns", bad_c[:8])
        if name + key in G.files:
            assert np.array_equal(m, G[name + key])


@pytest.mark.parametrize("i", range(len(fs.ZSCORE_CASES)))
def test_zscore_normalize_reproduces_the_reference_bit_exactly(i):
    readers = pkg("readers")
    shape, scale, offset, seed = fs.ZSCORE_CASES[i]
    x = fs.zscore_input(shape, scale, offset, seed)
    want = G["zscore_%d" % i]
    got = readers.zscore_normalize(x[None] if x.ndim == 2 else x)          # (the driver hands a 2-D image over as one channel)
    got = got[0] if x.ndim == 2 else got
    assert got.dtype == np.float32 and np.array_equal(got, want)
    if x.ndim == 3:                                                          # channels-last call of the reference == ours on the transposed view
        assert np.array_equal(got.transpose(1, 2, 0), G["zscore_hwc_%d" % i])
