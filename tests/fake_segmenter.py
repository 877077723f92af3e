"""A deterministic stand-in for the network in the inference-driver fixtures: exact integer arithmetic, a receptive field of 11 x 11
pixels, three classes.  Used by tests/golden/make_inference_golden.py (driving the REFERENCE's UNet/inference.py functions) and by
tests/test_inference_golden.py (driving this repository's) -- the fixtures pin the driver's pad / halo / crop / paste index work only,
not the network arithmetic."""
import numpy as np


def synthetic_image(h, w, c, seed):
    """float32 [h, w] (c == 0) or [h, w, c]: an integer hash mapped to [-4, 4) in steps of 1/64 -- identical on every numpy version"""
    cc = max(c, 1)
    i = np.arange(h, dtype=np.uint64)[:, None, None]
    j = np.arange(w, dtype=np.uint64)[None, :, None]
    k = np.arange(cc, dtype=np.uint64)[None, None, :]
    v = (i * np.uint64(2654435761) + j * np.uint64(40503) + k * np.uint64(977) + np.uint64(seed) * np.uint64(7919)) & np.uint64(0xFFFFFFFF)
    v = (v ^ (v >> np.uint64(13))) * np.uint64(1274126177) & np.uint64(0xFFFFFFFF)
    x = ((v >> np.uint64(23)).astype(np.int64) - 256).astype(np.float32) / np.float32(64.0)
    return x[:, :, 0] if c == 0 else x


def _box(q, r):
    """sum of q over the (2r+1) x (2r+1) window, zero outside (int64, exact)"""
    p = np.pad(q, ((r + 1, r), (r + 1, r)))
    s = p.cumsum(0).cumsum(1)
    n = 2 * r + 1
    return s[n:, n:] - s[:-n, n:] - s[n:, :-n] + s[:-n, :-n]


class FakeSegmenter:
    """the two methods UNet/inference.py:27-173 calls on its model object"""

    def __init__(self, radius):
        self.radius = radius
        self.calls = []

    def estimate_radius(self):
        return self.radius

    def get_keras_model(self):
        return self.forward

    def forward(self, batch):
        """NCHW float32 [1, C, H, W] -> scores [1, H, W, 3] (float64, integer-valued: np.argmax over them is exact)"""
        b = np.asarray(batch)
        assert b.ndim == 4 and b.shape[0] == 1
        self.calls.append(tuple(b.shape))
        q = np.rint(b[0].astype(np.float64) * 64.0).astype(np.int64)
        qs = sum((c + 1) * q[c] for c in range(q.shape[0]))
        s0 = _box(qs, 1) * 9
        s1 = _box(qs, 3) + 17
        s2 = -_box(qs, 5) // 2 + (np.indices(qs.shape).sum(0) % 7) * 5
        return np.stack([s0, s1, s2], -1).astype(np.float64)[None]


def predict_with(fake):
    """the `predict` hook of this repository's inference functions: HWC fp32 tile -> int32 [H, W] class map"""
    def predict(_unet, tile_hwc):
        sm = fake.forward(np.ascontiguousarray(tile_hwc.transpose(2, 0, 1))[None])
        return np.argmax(np.squeeze(sm), axis=-1).astype(np.int32)
    return predict


INFERENCE_CASES = [   # (h, w, c (0 = a 2-D image), radius, seed)
    (70, 83, 0, 96, 1), (70, 83, 3, 96, 2), (512, 512, 1, 96, 3),
    (1024, 1040, 0, 96, 4), (1090, 1040, 1, 112, 5), (1090, 1040, 3, 96, 6), (2100, 1500, 0, 112, 7), (2100, 1500, 2, 96, 8),
    (1700, 833, 1, 96, 9),
]
ZSCORE_CASES = [      # (shape, scale, offset, seed): std <= 1 and > 1, 2-D and 3-D, per-channel mixtures
    ((40, 56), 0.01, 3.0, 1), ((40, 56), 30.0, -7.0, 2), ((3, 40, 56), 0.01, 1.0, 3), ((3, 40, 56), 300.0, 5.0, 4), ((2, 33, 47), 1.0, 0.0, 5),
]


def zscore_input(shape, scale, offset, seed):
    if len(shape) == 2:
        return synthetic_image(shape[0], shape[1], 0, seed) * np.float32(scale) + np.float32(offset)
    x = synthetic_image(shape[1], shape[2], shape[0], seed).transpose(2, 0, 1).copy()
    x = x * np.float32(scale) + np.float32(offset)
    if shape[0] > 1:
        x[1] *= np.float32(0.001)          # one channel below the std <= 1 switch, the others above it (for the larger scales)
    return x


def mask_digest(mask):
    """(row hashes, column hashes) of a class map, uint32 each: a pixel that differs changes its row's and its column's entry (the big
    fixtures keep these instead of the masks themselves: 3 classes of pixel noise do not compress)"""
    m = np.asarray(mask).astype(np.uint64) + np.uint64(1)
    wj = (np.arange(m.shape[1], dtype=np.uint64) * np.uint64(2654435761) + np.uint64(12345)) & np.uint64(0xFFFFFFFF)
    wi = (np.arange(m.shape[0], dtype=np.uint64) * np.uint64(40503) + np.uint64(977)) & np.uint64(0xFFFFFFFF)
    rows = ((m * wj[None, :]) & np.uint64(0xFFFFFFFF)).sum(1) & np.uint64(0xFFFFFFFF)
    cols = ((m * wi[:, None]) & np.uint64(0xFFFFFFFF)).sum(0) & np.uint64(0xFFFFFFFF)
    return rows.astype(np.uint32), cols.astype(np.uint32)
