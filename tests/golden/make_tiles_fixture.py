"""Fixture: 16 of the reference's bundled 256x256 tiles (data/images/*.tif uint16, data/masks/*.tif {0,1}) as one .npz, so the
GPU box -- which has neither the reference tree nor a TIFF reader -- can run the train loop on real data (BASELINE config 1's
data).  Run with an interpreter that has scikit-image:

    /opt/conda/bin/python3.9 tests/golden/make_tiles_fixture.py /root/reference/data tests/golden/data_tiles.npz
"""
import glob, os, sys
import numpy as np
import skimage.io

src, out = sys.argv[1], sys.argv[2]
names = sorted(os.path.basename(f) for f in glob.glob(os.path.join(src, "images", "*.tif")))[::6][:16]
imgs = np.stack([skimage.io.imread(os.path.join(src, "images", n)) for n in names])
masks = np.stack([skimage.io.imread(os.path.join(src, "masks", n)) for n in names]).astype(np.uint8)
assert imgs.shape == (16, 256, 256) and imgs.dtype == np.uint16 and set(np.unique(masks)) <= {0, 1}
np.savez_compressed(out, images=imgs, masks=masks, names=np.array(names))
print("wrote", out, os.path.getsize(out) // 1024, "KiB; foreground fraction %.3f" % masks.mean())
