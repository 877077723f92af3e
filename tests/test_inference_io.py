"""Inference driver I/O on the host (reference UNet/inference.py:215-227): mask dtype by the largest label, and the mask file the
reference's `skimage.io.imsave(..., compress=6, bigtiff=True, tile=(1024, 1024))` call asks for."""
import struct
import zlib

import numpy as np
import pytest

from conftest import pkg

inf = pkg("inference")


def test_mask_dtype_follows_the_reference_rule():
    # UNet/inference.py:215-220: 0 <= max <= 255 -> uint8 (a 256-class mask, max label 255, is uint8); 255 < max < 65536 -> uint16;
    # max == 65536 matches neither cast and stays the arg-max's int32, like anything larger
    assert inf.mask_dtype(0) == np.uint8 and inf.mask_dtype(254) == np.uint8 and inf.mask_dtype(255) == np.uint8
    assert inf.mask_dtype(256) == np.uint16 and inf.mask_dtype(65535) == np.uint16
    assert inf.mask_dtype(65536) == np.int32 and inf.mask_dtype(70000) == np.int32


def _parse_bigtiff(path):
    b = open(path, "rb").read()
    order, magic, offsize, zero, ifd = struct.unpack_from("<2sHHHQ", b, 0)
    assert (order, magic, offsize, zero) == (b"II", 43, 8, 0)                 # little-endian BigTIFF
    (n,) = struct.unpack_from("<Q", b, ifd)
    tags = {}
    for i in range(n):
        tag, typ, cnt, val = struct.unpack_from("<HHQQ", b, ifd + 8 + 20 * i)
        tags[tag] = (typ, cnt, val)
    assert list(tags) == sorted(tags)                                         # the TIFF specification wants ascending tags
    return b, tags


@pytest.mark.parametrize("dtype,shape", [(np.uint8, (1500, 2100)), (np.uint16, (256, 256)), (np.int32, (1030, 5)), (np.uint8, (16, 16))])
def test_mask_tiff_is_a_tiled_deflate_bigtiff_and_round_trips(tmp_path, dtype, shape):
    rng = np.random.default_rng(0)
    mask = rng.integers(0, 200 if dtype == np.uint8 else 60000, shape).astype(dtype)
    path = str(tmp_path / "m.tif")
    inf._write_bigtiff_tiled(path, mask, 1024, 6)
    b, tags = _parse_bigtiff(path)
    assert tags[256][2] == shape[1] and tags[257][2] == shape[0] and tags[258][2] == mask.dtype.itemsize * 8
    assert tags[259][2] == 8 and tags[322][2] == 1024 and tags[323][2] == 1024                  # Adobe deflate, 1024 x 1024 tiles
    ty, tx = (shape[0] + 1023) // 1024, (shape[1] + 1023) // 1024
    n = ty * tx
    assert tags[324][1] == n and tags[325][1] == n
    offs = [tags[324][2]] if n == 1 else struct.unpack_from("<%dQ" % n, b, tags[324][2])
    cnts = [tags[325][2]] if n == 1 else struct.unpack_from("<%dQ" % n, b, tags[325][2])
    # decode by hand: every tile inflates to a full 1024 x 1024 block of the mask (zero padded at the edges)
    out = np.zeros((ty * 1024, tx * 1024), dtype)
    for t, (o, c) in enumerate(zip(offs, cnts)):
        blk = np.frombuffer(zlib.decompress(b[o:o + c]), dtype=np.dtype(dtype).newbyteorder("<")).reshape(1024, 1024)
        out[(t // tx) * 1024:(t // tx + 1) * 1024, (t % tx) * 1024:(t % tx + 1) * 1024] = blk
    assert np.array_equal(out[:shape[0], :shape[1]], mask) and not out[shape[0]:].any() and not out[:, shape[1]:].any()
    # ... and an independent reader (libtiff through Pillow) sees the same image
    Image = pytest.importorskip("PIL.Image")
    back = np.array(Image.open(path))
    assert back.dtype == dtype and np.array_equal(back, mask)
    # the driver's writer picks this format for 'tif' and leaves .npy alone
    inf._write(str(tmp_path / "n.tif"), mask, "tif")
    assert np.array_equal(np.array(Image.open(str(tmp_path / "n.tif"))), mask)
    inf._write(str(tmp_path / "n.npy"), mask, "npy")
    assert np.array_equal(np.load(str(tmp_path / "n.npy")), mask)
