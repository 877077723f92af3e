#!/usr/bin/env python3
"""`inference` -- caller of the eval-mode forward + argmax (reference UNet/inference.py), on the HIP engine.

Flags are the reference's (UNet/inference.py:234-241).  Behaviour kept: image -> float32 -> z-score over the whole image
(:201-206); reflect-pad bottom/right to a multiple of 16 (:30-47,143-157); images larger than 1024 px go through the
tiled path: zones of responsibility of 1024 - 2*radius with a halo of `radius` clamped at the borders, eval forward per
tile, halo stripped, pasted (:54-129); mask = argmax over classes, first maximum wins (:107,166), cast to
uint8/uint16/int32 by its maximum (:215-220), written as the reference's imsave call asks (:221-227): a deflate-compressed BigTIFF
in 1024 x 1024 tiles (tifffile when importable, else the writer below).  The argmax runs on the device (`unet_argmax`).  Reading:
.npy, or anything Pillow opens (the reference uses scikit-image, absent here).
"""
import argparse
import os

import numpy as np
import torch

from . import model as unet_model_module
from .readers import zscore_normalize

TILE_SIZE = 1024
SIZE_FACTOR = unet_model_module.UNet.SIZE_FACTOR


def _pad_to_factor(img):
    if img.ndim not in (2, 3):
        raise IOError("Invalid number of dimensions for input image. Expecting HW or HWC dimension ordering.")
    if img.ndim == 2:
        img = img[:, :, None]
    pad_y = (-img.shape[0]) % SIZE_FACTOR
    pad_x = (-img.shape[1]) % SIZE_FACTOR
    if pad_x or pad_y:
        img = np.pad(img, ((0, pad_y), (0, pad_x), (0, 0)), mode="reflect")
    return img, pad_y, pad_x


def _predict(unet, tile_hwc):
    """HWC fp32 tile -> int32 [H,W] class map via the eval-mode forward and the device argmax."""
    x = torch.as_tensor(np.ascontiguousarray(tile_hwc.transpose(2, 0, 1))[None].astype(np.float32))
    e = unet.engine
    return e.argmax(e.forward(x, training=False))[0].cpu().numpy()


def _inference(img, unet, predict=_predict):
    """(`predict(unet, tile_hwc) -> int32 [H, W]`: the forward + argmax of one tile; the default runs the HIP engine.  The hook exists so that
    tests/test_inference_golden.py can drive this function's pad / crop index work with the same stand-in network as the fixtures made
    by the reference's own UNet/inference.py:139-173.)"""
    img, pad_y, pad_x = _pad_to_factor(img)
    mask = predict(unet, img)
    return mask[:mask.shape[0] - pad_y, :mask.shape[1] - pad_x]


def _inference_tiling(img, unet, tile_size, predict=_predict):
    img, pad_y, pad_x = _pad_to_factor(img)
    height, width = img.shape[:2]
    mask = np.zeros((height, width), dtype=np.int32)
    radius = unet.estimate_radius()
    assert tile_size % SIZE_FACTOR == 0 and radius % SIZE_FACTOR == 0
    zone = tile_size - 2 * radius
    assert zone >= radius
    for y0 in range(0, height, zone):
        for x0 in range(0, width, zone):
            y1, x1 = min(y0 + zone, height), min(x0 + zone, width)
            # halo of `radius` on every side that exists; the reference drops the far halo entirely when the padded
            # window would cross the image edge (UNet/inference.py:86-96), which is reproduced here
            ty0 = y0 - radius if y0 - radius >= 0 else 0
            tx0 = x0 - radius if x0 - radius >= 0 else 0
            ty1 = y0 + zone + radius if y0 + zone + radius <= height else height
            tx1 = x0 + zone + radius if x0 + zone + radius <= width else width
            pred = predict(unet, img[ty0:ty1, tx0:tx1])
            oy, ox = y0 - ty0, x0 - tx0
            mask[y0:y1, x0:x1] = pred[oy:oy + (y1 - y0), ox:ox + (x1 - x0)]
    return mask[:height - pad_y, :width - pad_x]


def _read(path):
    if path.endswith(".npy"):
        return np.load(path)
    from PIL import Image
    return np.array(Image.open(path))


def _write_bigtiff_tiled(path, mask, tile=1024, level=6):
    """The file `skimage.io.imsave(path, mask, compress=6, bigtiff=True, tile=(1024, 1024))` asks tifffile for (reference
    UNet/inference.py:221-222): a little-endian BigTIFF, one image, 1024 x 1024 tiles (edge tiles zero-padded to full size), every tile
    zlib-deflated at level 6 (Compression = 8, Adobe deflate), one sample per pixel, MinIsBlack."""
    import struct
    import zlib
    assert mask.ndim == 2 and mask.dtype in (np.uint8, np.uint16, np.int32)
    h, w = mask.shape
    ty, tx = (h + tile - 1) // tile, (w + tile - 1) // tile
    tiles = []
    for j in range(ty):
        for i in range(tx):
            t = np.zeros((tile, tile), mask.dtype)
            blk = mask[j * tile:(j + 1) * tile, i * tile:(i + 1) * tile]
            t[:blk.shape[0], :blk.shape[1]] = blk
            tiles.append(zlib.compress(t.astype(mask.dtype.newbyteorder("<")).tobytes(), level))
    n = len(tiles)
    # (tag, type, count, value): types 3 = SHORT, 4 = LONG, 16 = LONG8; tags in ascending order
    fmt = 2 if mask.dtype == np.int32 else 1
    entries = [(256, 4, 1, w), (257, 4, 1, h), (258, 3, 1, mask.dtype.itemsize * 8), (259, 3, 1, 8), (262, 3, 1, 1), (277, 3, 1, 1),
               (284, 3, 1, 1), (322, 4, 1, tile), (323, 4, 1, tile), (324, 16, n, None), (325, 16, n, None), (339, 3, 1, fmt)]
    ifd_off = 16
    ifd_len = 8 + 20 * len(entries) + 8
    arrays_off = ifd_off + ifd_len                    # tile offsets, then byte counts (only out of line when n > 1)
    data_off = arrays_off + (16 * n if n > 1 else 0)
    offs, cur = [], data_off
    for t in tiles:
        offs.append(cur)
        cur += len(t) + (len(t) & 1)                  # word-aligned tiles
    with open(path, "wb") as f:
        f.write(struct.pack("<2sHHHQ", b"II", 43, 8, 0, ifd_off))
        f.write(struct.pack("<Q", len(entries)))
        for tag, typ, cnt, val in entries:
            if tag == 324:
                val = offs[0] if n == 1 else arrays_off
            elif tag == 325:
                val = len(tiles[0]) if n == 1 else arrays_off + 8 * n
            f.write(struct.pack("<HHQQ", tag, typ, cnt, val))
        f.write(struct.pack("<Q", 0))
        if n > 1:
            f.write(struct.pack("<%dQ" % n, *offs))
            f.write(struct.pack("<%dQ" % n, *[len(t) for t in tiles]))
        for t in tiles:
            f.write(t)
            if len(t) & 1:
                f.write(b"\0")


def _write(path, mask, image_format="tif"):
    if path.endswith(".npy"):
        np.save(path, mask)
        return
    if "tif" in image_format:                         # UNet/inference.py:221-222
        try:
            import tifffile                           # what scikit-image's imsave delegates to
            tifffile.imwrite(path, mask, bigtiff=True, tile=(TILE_SIZE, TILE_SIZE), compression="zlib", compressionargs={"level": 6})
        except (ImportError, TypeError, ValueError):  # no tifffile, or one without the compression / compressionargs keywords: the writer below
            _write_bigtiff_tiled(path, mask, TILE_SIZE, 6)
        return
    from PIL import Image                             # UNet/inference.py:224-227 (compress where the format has it)
    try:
        Image.fromarray(mask).save(path, compress_level=6)
    except TypeError:
        Image.fromarray(mask).save(path)


def mask_dtype(max_label):
    """uint8 / uint16 / int32 by the largest label, the reference's rule (UNet/inference.py:215-220): 0..255 -> uint8,
    256..65535 -> uint16, anything else stays the arg-max's int32"""
    if 0 <= max_label <= 255:
        return np.uint8
    if 255 < max_label < 65536:
        return np.uint16
    return np.int32


def inference(checkpoint_filepath, image_folder, output_folder, number_classes, number_channels, image_format, compute_dtype=None):
    os.makedirs(output_folder, exist_ok=True)
    names = sorted(f for f in os.listdir(image_folder) if f.endswith("." + image_format))
    unet = unet_model_module.UNet(number_classes, 1, number_channels, 1e-4, compute_dtype=compute_dtype)
    unet.load_checkpoint(checkpoint_filepath)
    for i, name in enumerate(names):
        print("{}/{}".format(i, len(names)))
        img = _read(os.path.join(image_folder, name)).astype(np.float32)
        hwc = img[:, :, None] if img.ndim == 2 else img
        img = zscore_normalize(hwc.transpose(2, 0, 1)).transpose(1, 2, 0)
        if img.shape[0] > TILE_SIZE or img.shape[1] > TILE_SIZE:
            seg = _inference_tiling(img, unet, TILE_SIZE)
        else:
            seg = _inference(img, unet)
        seg = seg.astype(mask_dtype(int(seg.max())))
        _write(os.path.join(output_folder, name), seg, image_format)


def main(argv=None):
    ap = argparse.ArgumentParser(prog="inference", description="Script to detect stars with the selected unet model")
    ap.add_argument("--checkpoint_filepath", dest="checkpoint_filepath", type=str, required=True)
    ap.add_argument("--image_folder", dest="image_folder", type=str, required=True)
    ap.add_argument("--output_folder", dest="output_folder", type=str, required=True)
    ap.add_argument("--number_classes", dest="number_classes", type=int, required=True)
    ap.add_argument("--number_channels", dest="number_channels", type=int, required=True)
    ap.add_argument("--image_format", dest="image_format", type=str, default="tif")
    ap.add_argument("--compute_dtype", choices=["fp32", "bf16"], default=None,
                    help="opt-in extra: bf16 = mixed-precision contractions (fp32 is the reference's arithmetic and the default)")
    a = ap.parse_args(argv)
    inference(a.checkpoint_filepath, a.image_folder, a.output_folder, a.number_classes, a.number_channels, a.image_format,
              compute_dtype=a.compute_dtype)


if __name__ == "__main__":
    main()
