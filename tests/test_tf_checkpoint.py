"""TensorBundle / object-graph checkpoint format (tf_checkpoint.py; reference UNet/train.py:96,184, UNet/model.py:81-83) on CPU.
TensorFlow is not installed, so nothing here can call it: the tests pin the published format's invariants -- known CRC-32C
vectors (RFC 3720) and TensorFlow's mask function, a hand-assembled index table parsed byte by byte, the 48-byte footer and
magic, prefix compression + restart points over several blocks, writer -> reader round trips with checksum verification and
corruption detection, and the Keras naming of this U-Net's 46 weighted layers."""
import os
import struct

import numpy as np
import pytest

from conftest import pkg


@pytest.fixture(scope="module")
def tfc():
    return pkg("tf_checkpoint")


def test_crc32c_known_vectors_and_mask(tfc):
    assert tfc.crc32c(b"123456789") == 0xE3069283                       # the CRC-32C check value
    assert tfc.crc32c(bytes(32)) == 0x8A9136AA                          # RFC 3720 B.4: 32 bytes of zeros
    assert tfc.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43                 # 32 bytes of 0xFF
    assert tfc.crc32c(bytes(range(32))) == 0x46DD794E                   # ascending
    assert tfc.crc32c(bytes(range(31, -1, -1))) == 0x113FDB5C           # descending
    assert tfc.crc32c(b"6789", tfc.crc32c(b"12345")) == 0xE3069283      # Extend
    a = np.arange(1000, dtype=np.float32)
    assert tfc.crc32c_array(a) == tfc.crc32c(a.tobytes())
    # crc32c::Mask(crc) = rotr(crc, 15) + 0xa282ead8; TensorFlow's own test: Mask(Value("foo")) != Value("foo"), Unmask inverts
    c = tfc.crc32c(b"foo")
    assert tfc.mask_crc(c) != c and tfc.unmask_crc(tfc.mask_crc(c)) == c
    assert tfc.mask_crc(0) == 0xa282ead8 and tfc.mask_crc(1 << 15) == 0xa282ead9


def test_varint_and_proto_roundtrip(tfc):
    for v in (0, 1, 127, 128, 300, 2 ** 31, 2 ** 63 - 1):
        b = tfc.put_varint(v)
        assert tfc.get_varint(b, 0) == (v, len(b))
    assert tfc.put_varint(300) == b"\xac\x02"
    e = tfc.encode_entry(1, (3, 3, 64, 128), 0, 4096, 294912, 0xDEADBEEF)
    # BundleEntryProto by hand: dtype=1 -> 08 01; shape -> 12 <len> (12 02 08 03)(12 02 08 03)(12 02 08 40)(12 03 08 80 01);
    # offset -> 20 80 20; size -> 28 80 80 12; crc32c (fixed32, field 6) -> 35 EF BE AD DE
    assert e == bytes.fromhex("0801" "1211" "12020803" "12020803" "12020840" "1203088001" "208020" "2880 8012".replace(" ", "")
                              + "35efbeadde")
    d = tfc.decode_entry(e)
    assert d == {"dtype": 1, "shape": (3, 3, 64, 128), "shard_id": 0, "offset": 4096, "size": 294912, "crc32c": 0xDEADBEEF, "slices": 0}
    assert tfc.decode_entry(tfc.encode_entry(9, (), 0, 0, 8, 5))["shape"] == ()
    h = tfc.encode_header(1)
    assert h == bytes.fromhex("0801" "1a02" "0801")
    assert tfc.decode_header(h) == {"num_shards": 1, "endianness": 0, "producer": 1}


def test_table_layout_by_hand(tfc, tmp_path):
    # two entries: "" -> "H", "ab" -> "xyz".  Data block: (0,0,1)"H" (0,2,3)"ab""xyz" + restart [0] + count 1.
    p = str(tmp_path / "t.index")
    tfc.write_table(p, [(b"", b"H"), (b"ab", b"xyz")])
    raw = open(p, "rb").read()
    block = bytes([0, 0, 1]) + b"H" + bytes([0, 2, 3]) + b"ab" + b"xyz" + struct.pack("<II", 0, 1)
    assert raw[:len(block)] == block
    assert raw[len(block)] == 0                                          # compression type: none
    crc = tfc.crc32c(b"\x00", tfc.crc32c(block))
    assert struct.unpack_from("<I", raw, len(block) + 1)[0] == tfc.mask_crc(crc)
    # footer: 48 bytes, magic last (little-endian 0xdb4775248b80fb57)
    assert raw[-8:] == bytes.fromhex("57fb808b247547db")
    meta_off = len(block) + 5
    assert raw[meta_off:meta_off + 8] == struct.pack("<II", 0, 1)        # empty metaindex block
    foot = raw[-48:]
    assert tfc.get_varint(foot, 0)[0] == meta_off and foot[1] == 8        # metaindex handle (offset, size 8)
    ioff = tfc.get_varint(foot, 2)[0]
    assert ioff == meta_off + 8 + 5
    # index block: one entry, key = short successor of "ab" = "b", value = BlockHandle(0, len(block))
    ib = bytes([0, 1, 2]) + b"b" + bytes([0, len(block)]) + struct.pack("<II", 0, 1)
    assert raw[ioff:ioff + len(ib)] == ib
    assert tfc.read_table(p) == [(b"", b"H"), (b"ab", b"xyz")]


def test_table_many_blocks_prefix_compression_and_corruption(tfc, tmp_path, monkeypatch):
    monkeypatch.setattr(tfc, "BLOCK_SIZE", 2048)                          # force several data blocks and separators
    rng = np.random.default_rng(0)
    items = [(b"", b"hdr")] + sorted((("model/layer_with_weights-%d/kernel/%04d" % (i % 46, i)).encode(), rng.bytes(int(rng.integers(1, 90))))
                                      for i in range(700))
    p = str(tmp_path / "big.index")
    tfc.write_table(p, items)
    assert tfc.read_table(p) == items
    raw = bytearray(open(p, "rb").read())
    assert len(raw) < sum(len(k) + len(v) for k, v in items)            # shared prefixes were compressed away
    raw[100] ^= 0x40
    open(p, "wb").write(raw)
    with pytest.raises(IOError):
        tfc.read_table(p)
    with pytest.raises(IOError):
        open(p, "wb").write(b"x" * 100); tfc.read_table(p)


def test_bundle_roundtrip_dtypes_strings_and_checksums(tfc, tmp_path):
    rng = np.random.default_rng(1)
    t = {"model/a/.ATTRIBUTES/VARIABLE_VALUE": rng.standard_normal((3, 3, 8, 16)).astype(np.float32),
         "model/b/.ATTRIBUTES/VARIABLE_VALUE": rng.standard_normal(7).astype(np.float64),
         "optimizer/iter/.ATTRIBUTES/VARIABLE_VALUE": np.array(1234567890123, np.int64),
         "optimizer/lr/.ATTRIBUTES/VARIABLE_VALUE": np.array(3e-4, np.float32),
         "_CHECKPOINTABLE_OBJECT_GRAPH": b"\x0a\x00graph-bytes\xff" * 50}
    stem = str(tmp_path / "checkpoint" / "ckpt")
    tfc.write_bundle(stem, t)
    assert sorted(os.listdir(os.path.dirname(stem))) == ["ckpt.data-00000-of-00001", "ckpt.index"]      # tf.train.Checkpoint.write's files
    b = tfc.read_bundle(stem)
    assert set(b) == set(t)
    for k, v in t.items():
        if isinstance(v, bytes):
            assert b[k] == v
        else:
            assert b[k].dtype == v.dtype and b[k].shape == v.shape and np.array_equal(b[k], v)
    assert tfc.list_bundle(stem)["model/a/.ATTRIBUTES/VARIABLE_VALUE"] == ("float32", (3, 3, 8, 16))
    assert tfc.list_bundle(stem)["optimizer/iter/.ATTRIBUTES/VARIABLE_VALUE"] == ("int64", ())
    # data shard = tensors back to back in key order; entry offsets / sizes say where
    entries = {k.decode(): tfc.decode_entry(v) for k, v in tfc.read_table(stem + ".index")[1:]}
    order = sorted(t, key=lambda s: s.encode())
    off = 0
    for k in order:
        assert entries[k]["offset"] == off
        off += entries[k]["size"]
    assert off == os.path.getsize(stem + ".data-00000-of-00001")
    e = entries["model/a/.ATTRIBUTES/VARIABLE_VALUE"]
    assert e["crc32c"] == tfc.mask_crc(tfc.crc32c(t["model/a/.ATTRIBUTES/VARIABLE_VALUE"].tobytes()))
    # string tensor layout: varint length, masked crc of the length (as uint32), bytes
    g = t["_CHECKPOINTABLE_OBJECT_GRAPH"]
    raw = open(stem + ".data-00000-of-00001", "rb").read()
    s0 = entries["_CHECKPOINTABLE_OBJECT_GRAPH"]["offset"]
    ln = tfc.put_varint(len(g))
    assert raw[s0:s0 + len(ln)] == ln
    assert struct.unpack_from("<I", raw, s0 + len(ln))[0] == tfc.mask_crc(tfc.crc32c(struct.pack("<I", len(g))))
    assert raw[s0 + len(ln) + 4:s0 + len(ln) + 4 + len(g)] == g
    # a flipped bit in the data shard is caught by the entry checksum
    rawb = bytearray(raw); rawb[entries["model/b/.ATTRIBUTES/VARIABLE_VALUE"]["offset"] + 3] ^= 1
    open(stem + ".data-00000-of-00001", "wb").write(rawb)
    with pytest.raises(IOError):
        tfc.read_bundle(stem)
    assert "model/a/.ATTRIBUTES/VARIABLE_VALUE" in tfc.read_bundle(stem, keys=lambda k: k.startswith("model/a"))


def test_checkpoints_are_readable_and_writable_without_the_hip_library(tfc, tmp_path, monkeypatch):
    # Inspecting / converting a checkpoint on a machine without the gfx950 build: the CRC falls back to the numpy table form.  Same
    # checksums as the native routine (RFC 3720 vectors, random data, continuation), and a bundle written with one is read by the other.
    assert tfc._crc32c_numpy(bytes(32)) == 0x8A9136AA and tfc._crc32c_numpy(bytes([0xFF] * 32)) == 0x62A8AB43
    assert tfc._crc32c_numpy(bytes(range(32))) == 0x46DD794E and tfc._crc32c_numpy(b"123456789") == 0xE3069283
    r = np.random.default_rng(0).integers(0, 256, 30011, dtype=np.uint8)
    assert tfc._crc32c_numpy(r) == tfc.crc32c_array(r) == tfc._crc32c_numpy(r[4097:], tfc._crc32c_numpy(r[:4097]))
    t = {"model/a/.ATTRIBUTES/VARIABLE_VALUE": np.random.default_rng(1).standard_normal((3, 3, 4, 8)).astype(np.float32),
         "_CHECKPOINTABLE_OBJECT_GRAPH": b"graph" * 9}
    native_stem, fallback_stem = str(tmp_path / "n" / "ckpt"), str(tmp_path / "f" / "ckpt")
    tfc.write_bundle(native_stem, t)
    monkeypatch.setattr(tfc, "_native", lambda: None)                  # "the library cannot be loaded"
    tfc.write_bundle(fallback_stem, t)
    for ext in (".index", ".data-00000-of-00001"):
        assert open(native_stem + ext, "rb").read() == open(fallback_stem + ext, "rb").read()
    b = tfc.read_bundle(native_stem)
    assert np.array_equal(b["model/a/.ATTRIBUTES/VARIABLE_VALUE"], t["model/a/.ATTRIBUTES/VARIABLE_VALUE"])


def test_unet_variable_naming_and_object_graph(tfc):
    eng = pkg("engine")
    layers = eng.layer_table(1, 2)
    all_layers, weighted = tfc.keras_graph(layers)
    names = [n for n, _, _ in all_layers]
    # UNet._build_model (UNet/model.py:85-146): 19 Conv2D (the 1x1 logits layer included), 4 Conv2DTranspose, 23 BatchNormalization,
    # 4 MaxPool2D, 2 Dropout, 4 Concatenate, Permute, Softmax, after the InputLayer
    assert names[0] == "input_1" and names[-2:] == ["permute", "softmax"]
    assert sum(n.startswith("conv2d_transpose") for n in names) == 4
    assert sum(n.startswith("conv2d") and not n.startswith("conv2d_transpose") for n in names) == 19
    assert sum(n.startswith("batch_normalization") for n in names) == 23
    assert sum(n.startswith("max_pooling2d") for n in names) == 4 and sum(n.startswith("dropout") for n in names) == 2
    assert sum(n.startswith("concatenate") for n in names) == 4
    assert len(weighted) == 46
    assert names[1:6] == ["conv2d", "batch_normalization", "conv2d_1", "batch_normalization_1", "max_pooling2d"]
    i4b = names.index("conv2d_7")                                           # conv_4b -> BN -> Dropout -> MaxPool (UNet/model.py:104-107)
    assert names[i4b:i4b + 4] == ["conv2d_7", "batch_normalization_7", "dropout", "max_pooling2d_3"]
    iu = names.index("conv2d_transpose")                                    # up_4 -> BN -> Concatenate (:116-117)
    assert names[iu:iu + 3] == ["conv2d_transpose", "batch_normalization_10", "concatenate"]
    vk = tfc.variable_keys(layers)
    assert len(vk) == 23 * 2 + 23 * 4
    d = {eng_name: stem for stem, eng_name, _ in vk}
    assert d["conv_1a/kernel"] == "model/layer_with_weights-0/kernel" and d["conv_1a/moving_var"] == "model/layer_with_weights-1/moving_variance"
    assert d["up_4/kernel"] == "model/layer_with_weights-20/kernel" and d["logits/beta"] == "model/layer_with_weights-45/beta"
    g = tfc.encode_object_graph(layers)
    nodes = tfc.decode_object_graph(g)
    root = nodes[0]["children"]
    assert set(root) == {"model", "optimizer"}
    mch = nodes[root["model"]]["children"]
    assert sum(k.startswith("layer_with_weights-") for k in mch) == 46 and sum(k.startswith("layer-") for k in mch) == len(all_layers)
    assert mch["layer_with_weights-0"] == mch["layer-1"]                     # same object under both names
    conv0 = nodes[mch["layer_with_weights-0"]]["children"]
    assert set(conv0) == {"kernel", "bias"}
    assert nodes[conv0["kernel"]]["attributes"] == {"VARIABLE_VALUE": "model/layer_with_weights-0/kernel/.ATTRIBUTES/VARIABLE_VALUE"}
    bn0 = nodes[mch["layer_with_weights-1"]]["children"]
    assert set(bn0) == {"gamma", "beta", "moving_mean", "moving_variance"}
    opt = nodes[root["optimizer"]]
    assert set(opt["children"]) == {"iter", "beta_1", "beta_2", "decay", "learning_rate"}
    assert len(opt["slots"]) == 2 * 92                                       # m and v for each of the 92 trainable tensors
    res = tfc.resolve_keys_through_object_graph(g, layers)
    assert res["bott_b/kernel"] == "model/layer_with_weights-18/kernel/.ATTRIBUTES/VARIABLE_VALUE"
    assert res["optimizer/m/bott_b/kernel"] == "model/layer_with_weights-18/kernel/.OPTIMIZER_SLOT/optimizer/m/.ATTRIBUTES/VARIABLE_VALUE"
    assert res["optimizer/iter"] == "optimizer/iter/.ATTRIBUTES/VARIABLE_VALUE"
    assert len(res) == 138 + 184 + 5
