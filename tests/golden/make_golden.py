#!/usr/bin/env python3
"""Generates tests/golden/unet_c1k2_32.npz.  Run in the build container (reads /root/reference/data, which does not
exist on the GPU box; the committed .npz is what the tests read).

The reference ships NO expected outputs for this path (no tests) and TensorFlow cannot run here
(SURVEY.md 8(c)), so the expected values are produced by THIS repository's fp64 oracle (oracle/unet_numpy.py): the
fixture pins the oracle and the HIP path to each other and to a frozen snapshot -- it is not a TensorFlow golden
vector ("parity unpinned").  Inputs are real data: two 32x32 crops of the reference's bundled example tiles
(data/images/*.tif, uint16) with their masks (data/masks/*.tif, {0,1}), normalised per the reader contract
(reference UNet/imagereader.py:33-66 z-score per channel; :302-312 int32 one-hot labels).
"""
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import unet_numpy as on      # noqa: E402

REF = "/root/reference/data"
TILES = [("img_000580.tif", 64, 208), ("img_004422.tif", 16, 16)]
SEED = 20261003


def zscore(img):
    img = img.astype(np.float32)
    std, mv = np.std(img), np.mean(img)
    return (img - mv) if std <= 1.0 else (img - mv) / std


def main():
    imgs, labs = [], []
    for name, y, x in TILES:
        im = np.array(Image.open(os.path.join(REF, "images", name)))[y:y + 32, x:x + 32]
        mk = np.array(Image.open(os.path.join(REF, "masks", name)))[y:y + 32, x:x + 32].astype(np.int32)
        imgs.append(zscore(im)[None])                                  # CHW, C = 1
        labs.append((mk[..., None] == np.arange(2)).astype(np.int32))  # HWK one-hot
    images = np.stack(imgs).astype(np.float32)
    labels = np.stack(labs)
    rng = np.random.default_rng(SEED)
    masks = {"drop_4": rng.integers(0, 2, (2, 512, 4, 4)).astype(np.uint8),
             "drop_b": rng.integers(0, 2, (2, 1024, 2, 2)).astype(np.uint8)}
    prm = golden_params()
    o = on.OracleUNet(2, 2, 1, params=prm, dtype=np.float64)
    sm_eval, _ = o.forward(images, training=False)
    loss_eval, _ = o.test_step(images, labels)
    loss, sm_train, g, _, _ = o.loss_and_grads(images, labels, masks)
    out = dict(images=images, labels=labels, drop_4=masks["drop_4"], drop_b=masks["drop_b"], seed=np.int64(SEED),
               softmax_eval=sm_eval, mask_eval=np.argmax(sm_eval, -1).astype(np.int32), loss_eval=np.float64(loss_eval),
               softmax_train=sm_train, loss_train=np.float64(loss))
    for k, v in g.items():
        if v.size <= 4096:
            out["grad/" + k] = v
        else:
            out["gradnorm/" + k] = np.float64(np.linalg.norm(v))
            out["gradsum/" + k] = np.float64(v.sum())
    np.savez_compressed(os.path.join(HERE, "unet_c1k2_32.npz"), **out)
    print("wrote", os.path.join(HERE, "unet_c1k2_32.npz"), "loss", loss, "eval loss", loss_eval,
          "fg fraction", labels[..., 1].mean())


def golden_params():
    """Seeded weights (not stored: 124 MB): Keras-default init with perturbed bias/gamma/beta/moving stats."""
    rng = np.random.default_rng(SEED + 1)
    prm = on.init_params(1, 2, seed=SEED)
    for key in prm:
        if key.endswith(("bias", "beta")):
            prm[key] = rng.normal(0, 0.1, prm[key].shape).astype(np.float32)
        elif key.endswith("gamma"):
            prm[key] = rng.uniform(0.5, 1.5, prm[key].shape).astype(np.float32)
        elif key.endswith("moving_mean"):
            prm[key] = rng.normal(0.3, 0.1, prm[key].shape).astype(np.float32)
        elif key.endswith("moving_var"):
            prm[key] = rng.uniform(0.5, 1.5, prm[key].shape).astype(np.float32)
    return prm


if __name__ == "__main__":
    main()
