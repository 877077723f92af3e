"""Writes tests/golden/inference_driver.npz by running the REFERENCE's own functions -- UNet/inference.py:_inference_tiling / _inference
(:27-173) and UNet/imagereader.py:zscore_normalize (:33-66) -- on deterministic inputs with a deterministic stand-in for the network
(tests/fake_segmenter.py).  Run in the build container only (needs /root/reference):

    python tests/golden/make_inference_golden.py

`tensorflow`, `skimage`, `lmdb` and the generated protobuf module are replaced by import-only stand-ins: the three functions under test
are numpy code around an injected model object and call none of them beyond a pass-through `tf.convert_to_tensor`.  The fixture holds DATA
only (masks, z-scored arrays) and pins the inference driver's integer work -- reflect padding, zones of responsibility, halos, crops,
pasting -- NOT the network arithmetic."""
import contextlib
import io
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import fake_segmenter as fs                                         # noqa: E402

REF = "/root/reference/UNet"
tf = types.ModuleType("tensorflow"); tf.__version__ = "2.0.0"; tf.function = lambda f: f; tf.convert_to_tensor = lambda x: x
sys.modules["tensorflow"] = tf
for name in ("skimage", "skimage.io", "skimage.transform", "lmdb"):
    sys.modules[name] = types.ModuleType(name)
pb = types.ModuleType("isg_ai_pb2"); pb.ImageMaskPair = object; sys.modules["isg_ai_pb2"] = pb
sys.path.insert(0, REF)
import inference as ref_inference                                   # noqa: E402  (the reference's module)
import imagereader as ref_reader                                    # noqa: E402

out = {}
for (h, w, c, radius, seed) in fs.INFERENCE_CASES:
    img = fs.synthetic_image(h, w, c, seed)
    key = "%dx%dx%d_r%d_s%d" % (h, w, c, radius, seed)
    with contextlib.redirect_stdout(io.StringIO()):
        fake = fs.FakeSegmenter(radius)
        tiled = ref_inference._inference_tiling(img.copy(), fake, ref_inference.TILE_SIZE)
        calls = np.array(fake.calls, dtype=np.int32)
        whole = ref_inference._inference(img.copy(), fs.FakeSegmenter(radius))
    assert tiled.dtype == np.int32 and whole.dtype == np.int32 and tiled.shape == (h, w) and whole.shape == (h, w)
    for name, m in (("tiled_", tiled), ("whole_", whole)):
        out[name + "rows_" + key], out[name + "cols_" + key] = fs.mask_digest(m)
        if h * w <= 512 * 512:
            out[name + key] = m.astype(np.uint8)                     # the small masks in full
    out["calls_" + key] = calls                                      # the tile shapes the reference handed to the model, in order
    print(key, "tiles", len(calls), "classes", np.bincount(tiled.ravel(), minlength=3), "tiled != whole on", int((tiled != whole).sum()), "pixels")
for i, (shape, scale, offset, seed) in enumerate(fs.ZSCORE_CASES):
    x = fs.zscore_input(shape, scale, offset, seed)
    out["zscore_%d" % i] = ref_reader.zscore_normalize(x.copy())     # channels first (CHW) or a 2-D image
    if len(shape) == 3:
        out["zscore_hwc_%d" % i] = ref_reader.zscore_normalize(np.ascontiguousarray(x.transpose(1, 2, 0)), channels_first=False)
np.savez_compressed(os.path.join(HERE, "inference_driver.npz"), **out)
print("wrote", os.path.join(HERE, "inference_driver.npz"), "%.1f KB" % (os.path.getsize(os.path.join(HERE, "inference_driver.npz")) / 1e3))
