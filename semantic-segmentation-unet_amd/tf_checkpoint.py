"""Reader / writer for the checkpoint format the reference saves and restores: `tf.train.Checkpoint(optimizer=..., model=...)`
(UNet/train.py:96,184 `checkpoint.write(<out>/checkpoint/ckpt)`; UNet/model.py:81-83 `checkpoint.restore(path).expect_partial()`),
i.e. a TensorFlow **TensorBundle** (`ckpt.index` + `ckpt.data-00000-of-00001`) whose keys follow the object graph of the
checkpointed Python objects.  TensorFlow is not installed here, so this module restates
the published on-disk format from its specification; it is **unverified against TensorFlow itself** -- what the tests pin is the
writer -> reader round trip, the format's own checksums and the hand-checked layout of small examples.

Format (tensorflow/core/util/tensor_bundle, tensorflow/core/lib/io/{table_builder,format,block_builder}.cc, core/protobuf/
tensor_bundle.proto, trackable_object_graph.proto):
  * `<prefix>.data-SSSSS-of-NNNNN`: the tensors' raw little-endian bytes back to back;
  * `<prefix>.index`: an immutable sorted string table (the LevelDB table format): data blocks of prefix-compressed
    (shared, non_shared, value_len varint32; key delta; value) entries with restart points, each block followed by a 1-byte
    compression type (0) and a masked CRC-32C; an (empty) metaindex block, an index block of (separator key -> BlockHandle) and a
    48-byte footer (two BlockHandles padded to 40 bytes + magic 0xdb4775248b80fb57).  Key "" holds `BundleHeaderProto`
    (num_shards, endianness, version.producer = 1); every other key holds a `BundleEntryProto` (dtype, shape, shard_id, offset,
    size, masked crc32c of the bytes);
  * tensor keys are object-graph paths: `model/layer_with_weights-<N>/{kernel,bias,gamma,beta,moving_mean,moving_variance}/
    .ATTRIBUTES/VARIABLE_VALUE`, with N counting the Keras layers that own weights in graph order -- for this U-Net (UNet/model.py:
    85-146) layer i of the 23 conv / transposed-conv layers is N = 2i and its BatchNormalization is N = 2i + 1; optimizer
    hyper-parameters `optimizer/{iter,beta_1,beta_2,decay,learning_rate}/...`; Adam slots `model/layer_with_weights-<N>/<var>/
    .OPTIMIZER_SLOT/optimizer/{m,v}/.ATTRIBUTES/VARIABLE_VALUE`; and the serialized `TrackableObjectGraph` under
    `_CHECKPOINTABLE_OBJECT_GRAPH` (a DT_STRING scalar), which TensorFlow's restore walks edge by edge.
Variable layouts are Keras' and therefore this build's own (Conv2D HWIO, Conv2DTranspose [kh,kw,Cout,Cin]): no transposition.
"""
import os
import struct

import numpy as np


HEADER_KEY = b""
OBJECT_GRAPH_KEY = b"_CHECKPOINTABLE_OBJECT_GRAPH"
TABLE_MAGIC = 0xdb4775248b80fb57
BLOCK_RESTART_INTERVAL = 16          # table::Options defaults (lib/io/table_options.h)
BLOCK_SIZE = 262144
DT = {"float32": 1, "float64": 2, "int32": 3, "uint8": 4, "int16": 5, "int8": 6, "string": 7, "int64": 9, "bool": 10}     # types.proto
DT_INV = {v: k for k, v in DT.items()}
MASK_DELTA = 0xa282ead8


# ----------------------------------------------------------------------------------------------------------- crc32c / varints
# CRC-32C (Castagnoli, reflected polynomial 0x82F63B78) comes from the native library when it loads (hardware-speed: the data shard is
# 124 MB); on a machine without the HIP build -- inspecting or converting a checkpoint elsewhere -- the numpy table-driven form below
# takes over (slicing-by-8 tables, a Python loop over 8-byte words: a few MB/s -- instant for an index, about half a minute for the
# data shard; `verify=False` skips it on reads), so this module needs nothing but numpy.
_TABLES = None


def _crc_tables():
    global _TABLES
    if _TABLES is None:
        t = np.zeros((8, 256), np.uint32)
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
            t[0, i] = c
        for k in range(1, 8):
            t[k] = (t[k - 1] >> np.uint32(8)) ^ t[0][t[k - 1] & np.uint32(0xFF)]
        _TABLES = t
    return _TABLES


def _crc32c_numpy(buf, init=0):
    """CRC-32C of a bytes-like / uint8 array, continuing from `init` (same convention as the native unet_crc32c_extend)."""
    t = _crc_tables()
    a = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf.reshape(-1).view(np.uint8)
    crc = (~init) & 0xFFFFFFFF
    n = len(a)
    i = 0
    t0 = [int(v) for v in t[0]]
    if n >= 64:
        # the CRC of a long message is a sequential recurrence; process it 8 bytes at a time with the slicing-by-8 tables
        m = (n // 8) * 8
        words = a[:m].reshape(-1, 8)
        tl = [tt.astype(np.uint64) for tt in t]
        c = np.uint64(crc)
        for w in words:                                   # (python-level loop over 8-byte words: fine for indexes, slow but correct for shards)
            lo = (int(w[0]) | int(w[1]) << 8 | int(w[2]) << 16 | int(w[3]) << 24) ^ int(c)
            c = (int(tl[7][lo & 0xFF]) ^ int(tl[6][(lo >> 8) & 0xFF]) ^ int(tl[5][(lo >> 16) & 0xFF]) ^ int(tl[4][(lo >> 24) & 0xFF])
                 ^ int(tl[3][int(w[4])]) ^ int(tl[2][int(w[5])]) ^ int(tl[1][int(w[6])]) ^ int(tl[0][int(w[7])]))
        crc = int(c)
        i = m
    for b in a[i:]:
        crc = t0[(crc ^ int(b)) & 0xFF] ^ (crc >> 8)
    return (~crc) & 0xFFFFFFFF


_NATIVE = [False, None]          # [resolved?, function or None]


def _native():
    """the native CRC routine, or None where the HIP library is not there (no build): resolved ONCE.  A library that is there but stale or of
    another ABI version is an error (UnetHipError propagates) -- falling back silently would hide it behind a CRC that takes half a minute
    per shard."""
    if not _NATIVE[0]:
        fn = None
        try:
            from . import _lib                         # (lazy: importing it pulls in torch; this module otherwise needs numpy only)
            if os.path.exists(_lib.LIB_PATH):
                fn = _lib.lib().unet_crc32c_extend      # UnetHipError (stale build / ABI mismatch) is NOT caught
        except (ImportError, OSError):
            fn = None
        if fn is None:
            import warnings
            warnings.warn("libunet_hip.so is not available: checkpoint CRC-32C falls back to the numpy routine (slow on 100 MB shards)")
        _NATIVE[0], _NATIVE[1] = True, fn
    return _NATIVE[1]


def crc32c(data, init=0):
    b = bytes(data)
    f = _native()
    return int(f(init, b, len(b))) if f is not None else _crc32c_numpy(b, init)


def crc32c_array(arr, init=0):
    """CRC of a C-contiguous numpy array without copying it."""
    a = np.ascontiguousarray(arr)
    f = _native()
    return int(f(init, a.ctypes.data, a.nbytes)) if f is not None else _crc32c_numpy(a, init)


def mask_crc(c):
    """crc32c::Mask: rotate right by 15 and add a constant (a CRC of data that itself contains CRCs stays well-behaved)."""
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + MASK_DELTA) & 0xFFFFFFFF


def unmask_crc(m):
    rot = (m - MASK_DELTA) & 0xFFFFFFFF
    return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


def put_varint(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def get_varint(buf, pos):
    shift = val = 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7


# ----------------------------------------------------------------------------------------------------------- protobuf wire format
def pb_field(num, wire, payload):
    return put_varint((num << 3) | wire) + payload


def pb_varint(num, v):
    return pb_field(num, 0, put_varint(v & 0xFFFFFFFFFFFFFFFF))          # negative int64 -> 10-byte two's complement


def pb_bytes(num, b):
    return pb_field(num, 2, put_varint(len(b)) + bytes(b))


def pb_fixed32(num, v):
    return pb_field(num, 5, struct.pack("<I", v))


def pb_parse(buf):
    """-> list of (field number, wire type, value): varint -> int, length-delimited -> bytes, fixed32/64 -> int."""
    out, pos, n = [], 0, len(buf)
    while pos < n:
        tag, pos = get_varint(buf, pos)
        num, wire = tag >> 3, tag & 7
        if wire == 0:
            v, pos = get_varint(buf, pos)
        elif wire == 2:
            ln, pos = get_varint(buf, pos)
            v = bytes(buf[pos:pos + ln]); pos += ln
        elif wire == 5:
            v = struct.unpack_from("<I", buf, pos)[0]; pos += 4
        elif wire == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]; pos += 8
        else:
            raise ValueError("unsupported protobuf wire type %d" % wire)
        out.append((num, wire, v))
    return out


def _signed64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def encode_shape(shape):
    """TensorShapeProto: repeated Dim dim = 2 { int64 size = 1 }"""
    return b"".join(pb_bytes(2, pb_varint(1, int(d))) for d in shape)


def decode_shape(buf):
    dims = []
    for num, _, v in pb_parse(buf):
        if num == 2:
            size = 0
            for n2, _, v2 in pb_parse(v):
                if n2 == 1:
                    size = _signed64(v2)
            dims.append(size)
        elif num == 3 and v:
            raise ValueError("tensor of unknown rank in a checkpoint")
    return tuple(dims)


def encode_entry(dtype, shape, shard_id, offset, size, crc_masked):
    """BundleEntryProto (proto3: zero-valued scalars are omitted; the shape message is always present)."""
    out = pb_varint(1, dtype) + pb_bytes(2, encode_shape(shape))
    if shard_id:
        out += pb_varint(3, shard_id)
    if offset:
        out += pb_varint(4, offset)
    if size:
        out += pb_varint(5, size)
    if crc_masked:
        out += pb_fixed32(6, crc_masked)
    return out


def decode_entry(buf):
    e = {"dtype": 0, "shape": (), "shard_id": 0, "offset": 0, "size": 0, "crc32c": 0, "slices": 0}
    for num, _, v in pb_parse(buf):
        if num == 1:
            e["dtype"] = v
        elif num == 2:
            e["shape"] = decode_shape(v)
        elif num == 3:
            e["shard_id"] = v
        elif num == 4:
            e["offset"] = _signed64(v)
        elif num == 5:
            e["size"] = _signed64(v)
        elif num == 6:
            e["crc32c"] = v
        elif num == 7:
            e["slices"] += 1
    return e


def encode_header(num_shards=1):
    """BundleHeaderProto: num_shards = 1; endianness = 2 (LITTLE = 0, omitted); version = 3 { producer = 1 }"""
    return pb_varint(1, num_shards) + pb_bytes(3, pb_varint(1, 1))


def decode_header(buf):
    h = {"num_shards": 0, "endianness": 0, "producer": 0}
    for num, _, v in pb_parse(buf):
        if num == 1:
            h["num_shards"] = v
        elif num == 2:
            h["endianness"] = v
        elif num == 3:
            for n2, _, v2 in pb_parse(v):
                if n2 == 1:
                    h["producer"] = v2
    return h


# ----------------------------------------------------------------------------------------------------------- sorted string table
def _block_bytes(entries):
    """BlockBuilder: prefix-compressed entries + restart array + restart count."""
    out, restarts, last = bytearray(), [], b""
    for i, (k, v) in enumerate(entries):
        shared = 0
        if i % BLOCK_RESTART_INTERVAL == 0:
            restarts.append(len(out))
        else:
            m = min(len(last), len(k))
            while shared < m and last[shared] == k[shared]:
                shared += 1
        out += put_varint(shared) + put_varint(len(k) - shared) + put_varint(len(v)) + k[shared:] + v
        last = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def _shortest_separator(a, b):
    """BytewiseComparator::FindShortestSeparator: a short key k with a <= k < b (index-block keys)."""
    m = min(len(a), len(b))
    d = 0
    while d < m and a[d] == b[d]:
        d += 1
    if d < m and a[d] < 0xFF and a[d] + 1 < b[d]:
        return a[:d] + bytes([a[d] + 1])
    return a


def _short_successor(a):
    """BytewiseComparator::FindShortSuccessor: a short key >= a (the last index entry)."""
    for i, c in enumerate(a):
        if c != 0xFF:
            return a[:i] + bytes([c + 1])
    return a


def write_table(path, items):
    """items: list of (key bytes, value bytes) sorted by key, keys unique."""
    assert all(items[i][0] < items[i + 1][0] for i in range(len(items) - 1)), "table keys must be strictly increasing"
    out = bytearray()
    index = []

    def emit(block):
        off = len(out)
        trailer = b"\x00"                                                  # kNoCompression
        crc = mask_crc(crc32c(trailer, crc32c(block)))
        out.extend(block); out.extend(trailer); out.extend(struct.pack("<I", crc))
        return put_varint(off) + put_varint(len(block))                    # BlockHandle

    cur, cur_size, pending = [], 0, None
    for k, v in items:
        if pending is not None:                                             # the previous block was flushed: index it now
            index.append((_shortest_separator(pending[0], k), pending[1]))
            pending = None
        cur.append((k, v))
        cur_size += len(k) + len(v) + 3
        if cur_size >= BLOCK_SIZE:
            pending = (cur[-1][0], emit(_block_bytes(cur)))
            cur, cur_size = [], 0
    if cur:
        pending = (cur[-1][0], emit(_block_bytes(cur)))
    if pending is not None:
        index.append((_short_successor(pending[0]), pending[1]))
    meta_handle = emit(_block_bytes([]))
    index_handle = emit(_block_bytes(index))
    footer = meta_handle + index_handle
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<II", TABLE_MAGIC & 0xFFFFFFFF, TABLE_MAGIC >> 32)
    out.extend(footer)
    with open(path, "wb") as f:
        f.write(out)


def _read_block(buf, off, size, verify=True):
    block = buf[off:off + size]
    ctype = buf[off + size]
    stored = struct.unpack_from("<I", buf, off + size + 1)[0]
    if verify and unmask_crc(stored) != crc32c(buf[off + size:off + size + 1], crc32c(block)):
        raise IOError("checkpoint index: block checksum mismatch at offset %d" % off)
    if ctype != 0:
        raise IOError("checkpoint index: compressed table blocks (type %d) are not supported" % ctype)
    return block


def _block_entries(block):
    n_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * n_restarts
    pos, key, out = 0, b"", []
    while pos < end:
        shared, pos = get_varint(block, pos)
        non_shared, pos = get_varint(block, pos)
        vlen, pos = get_varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared]); pos += non_shared
        out.append((key, bytes(block[pos:pos + vlen]))); pos += vlen
    return out


def read_table(path, verify=True):
    """-> list of (key, value) in key order."""
    buf = open(path, "rb").read()
    if len(buf) < 48:
        raise IOError("%s is too short to be a TensorBundle index" % path)
    lo, hi = struct.unpack_from("<II", buf, len(buf) - 8)
    if (hi << 32 | lo) != TABLE_MAGIC:
        raise IOError("%s is not a TensorBundle index (bad table magic)" % path)
    pos = len(buf) - 48
    _, pos = get_varint(buf, pos); _, pos = get_varint(buf, pos)            # metaindex handle (unused)
    ioff, pos = get_varint(buf, pos); isize, pos = get_varint(buf, pos)
    out = []
    for _, handle in _block_entries(_read_block(buf, ioff, isize, verify)):
        boff, p2 = get_varint(handle, 0); bsize, _ = get_varint(handle, p2)
        out.extend(_block_entries(_read_block(buf, boff, bsize, verify)))
    return out


# ----------------------------------------------------------------------------------------------------------- bundle
def _shard_name(prefix, i, n):
    return "%s.data-%05d-of-%05d" % (prefix, i, n)


def _encode_string_tensor(strings):
    """WriteStringTensor: [varint64 len]* [fixed32 masked crc of the lengths (each as uint32)] [bytes]*; returns (payload, crc)
    where crc runs over the lengths (as uint32 LE), the 4 checksum bytes and the string bytes."""
    lens = b"".join(put_varint(len(s)) for s in strings)
    c = 0
    for s in strings:
        c = crc32c(struct.pack("<I", len(s)), c)
    cks = struct.pack("<I", mask_crc(c))
    c = crc32c(cks, c)
    for s in strings:
        c = crc32c(s, c)
    return lens + cks + b"".join(strings), c


def _decode_string_tensor(raw, count):
    pos, lens = 0, []
    for _ in range(count):
        ln, pos = get_varint(raw, pos)
        lens.append(ln)
    pos += 4
    out = []
    for ln in lens:
        out.append(bytes(raw[pos:pos + ln])); pos += ln
    return out


def write_bundle(prefix, tensors):
    """tensors: {key (str): numpy array | bytes (a DT_STRING scalar)}.  Writes <prefix>.index and one data shard."""
    d = os.path.dirname(prefix)
    if d:
        os.makedirs(d, exist_ok=True)
    items = [(HEADER_KEY, encode_header(1))]
    off = 0
    tmp = _shard_name(prefix, 0, 1) + ".tmp"
    with open(tmp, "wb") as f:
        for key in sorted(tensors, key=lambda s: s.encode()):
            v = tensors[key]
            if isinstance(v, (bytes, bytearray)):
                payload, c = _encode_string_tensor([bytes(v)])
                dtype, shape = DT["string"], ()
                f.write(payload)
                size = len(payload)
            else:
                a = np.asarray(v)
                a = np.ascontiguousarray(a) if a.ndim else a               # (ascontiguousarray would turn a scalar into shape (1,))
                if a.dtype.byteorder == ">":
                    a = a.astype(a.dtype.newbyteorder("<"))
                dtype, shape = DT[a.dtype.name], a.shape
                c = crc32c_array(a)
                f.write(a.tobytes() if not a.ndim else a.data)
                size = a.nbytes
            items.append((key.encode(), encode_entry(dtype, shape, 0, off, size, mask_crc(c))))
            off += size
    os.replace(tmp, _shard_name(prefix, 0, 1))
    write_table(prefix + ".index.tmp", items)
    os.replace(prefix + ".index.tmp", prefix + ".index")


def read_bundle(prefix, keys=None, verify=True):
    """-> {key (str): numpy array | bytes}.  keys: optional predicate or collection restricting what is loaded."""
    items = read_table(prefix + ".index", verify)
    if not items or items[0][0] != HEADER_KEY:
        raise IOError("%s.index has no bundle header entry" % prefix)
    hdr = decode_header(items[0][1])
    if hdr["endianness"] != 0:
        raise IOError("big-endian checkpoints are not supported")
    want = (lambda k: True) if keys is None else (keys if callable(keys) else (lambda k, s=set(keys): k in s))
    shards, out = {}, {}
    for kb, vb in items[1:]:
        key = kb.decode()
        if not want(key):
            continue
        e = decode_entry(vb)
        if e["slices"]:
            raise IOError("partitioned variable %s: sliced checkpoint entries are not supported" % key)
        if e["shard_id"] not in shards:
            shards[e["shard_id"]] = np.memmap(_shard_name(prefix, e["shard_id"], hdr["num_shards"]), dtype=np.uint8, mode="r")
        raw = shards[e["shard_id"]][e["offset"]:e["offset"] + e["size"]]
        if len(raw) != e["size"]:
            raise IOError("checkpoint data shard is truncated at %s" % key)
        name = DT_INV.get(e["dtype"])
        if name is None:
            raise IOError("%s: unsupported checkpoint dtype %d" % (key, e["dtype"]))
        if name == "string":
            count = int(np.prod(e["shape"])) if e["shape"] else 1
            raw_b = bytes(raw)
            strings = _decode_string_tensor(raw_b, count)
            if verify:
                _, c = _encode_string_tensor(strings)
                if mask_crc(c) != e["crc32c"]:
                    raise IOError("checksum mismatch in checkpoint entry " + key)
            out[key] = strings[0] if not e["shape"] else strings
            continue
        if verify and mask_crc(crc32c_array(np.asarray(raw))) != e["crc32c"]:
            raise IOError("checksum mismatch in checkpoint entry " + key)
        out[key] = np.frombuffer(raw, dtype=np.dtype(name).newbyteorder("<")).reshape(e["shape"]).copy()
    return out


def list_bundle(prefix):
    """-> {key: (dtype name, shape)} without touching the data shards."""
    out = {}
    for kb, vb in read_table(prefix + ".index")[1:]:
        e = decode_entry(vb)
        out[kb.decode()] = (DT_INV.get(e["dtype"], str(e["dtype"])), e["shape"])
    return out


# ----------------------------------------------------------------------------------------------------------- object graph
ATTR = "/.ATTRIBUTES/VARIABLE_VALUE"
CONV_VARS = ("kernel", "bias")
BN_VARS = ("gamma", "beta", "moving_mean", "moving_variance")
OPT_HYPER = ("iter", "beta_1", "beta_2", "decay", "learning_rate")


def keras_graph(layers):
    """The Keras layer sequence of UNet._build_model (UNet/model.py:85-146) -> (all_layers, weighted):
    all_layers = [(keras name, engine layer name or None, 'conv' | 'bn' | None)] in model.layers order (InputLayer first),
    weighted   = indices into all_layers of the layers that own weights (= the layer_with_weights-N numbering)."""
    all_layers = [("input_1", None, None)]
    n_conv = n_convt = n_bn = n_pool = n_drop = n_cat = 0

    def suffix(base, n):
        return base if n == 0 else "%s_%d" % (base, n)

    for name, kind, _, _ in layers:
        if kind == "deconv":
            all_layers.append((suffix("conv2d_transpose", n_convt), name, "conv")); n_convt += 1
        else:
            all_layers.append((suffix("conv2d", n_conv), name, "conv")); n_conv += 1
        all_layers.append((suffix("batch_normalization", n_bn), name, "bn")); n_bn += 1
        if kind == "deconv":
            all_layers.append((suffix("concatenate", n_cat), None, None)); n_cat += 1
        if name in ("conv_4b", "bott_b"):
            all_layers.append((suffix("dropout", n_drop), None, None)); n_drop += 1
        if name in ("conv_1b", "conv_2b", "conv_3b", "conv_4b"):
            all_layers.append((suffix("max_pooling2d", n_pool), None, None)); n_pool += 1
    all_layers += [("permute", None, None), ("softmax", None, None)]
    weighted = [i for i, (_, eng, _) in enumerate(all_layers) if eng is not None]
    return all_layers, weighted


def variable_keys(layers):
    """-> list of (checkpoint key stem, engine tensor name, keras full name) for every model variable; a stem + ATTR is the
    variable's key, stem + '/.OPTIMIZER_SLOT/optimizer/{m,v}' + ATTR its Adam slots."""
    all_layers, weighted = keras_graph(layers)
    out = []
    for n, li in enumerate(weighted):
        kname, eng, what = all_layers[li]
        for var in (CONV_VARS if what == "conv" else BN_VARS):
            eng_name = eng + "/" + {"moving_variance": "moving_var"}.get(var, var)
            out.append(("model/layer_with_weights-%d/%s" % (n, var), eng_name, "%s/%s" % (kname, var)))
    return out


def encode_object_graph(layers):
    """TrackableObjectGraph for Checkpoint(optimizer=Adam, model=<functional Model>): node 0 = root with children `model`,
    `optimizer`; the model node references every layer as `layer-<i>` and the weighted ones also as `layer_with_weights-<n>`;
    layer nodes reference their variables; variable nodes carry the VARIABLE_VALUE attribute (name, full_name, checkpoint_key);
    the optimizer node references its hyper-parameter variables and lists the slot variables."""
    all_layers, weighted = keras_graph(layers)
    nodes = []                      # each: dict(children=[(node_id, local_name)], attrs=[(name, full_name, key)], slots=[(orig, slot, node)])

    def new_node():
        nodes.append({"children": [], "attrs": [], "slots": []})
        return len(nodes) - 1

    root, model, opt = new_node(), new_node(), new_node()
    nodes[root]["children"] += [(model, "model"), (opt, "optimizer")]
    layer_node = []
    for i, _ in enumerate(all_layers):
        ln = new_node()
        layer_node.append(ln)
        nodes[model]["children"].append((ln, "layer-%d" % i))
    for n, li in enumerate(weighted):
        nodes[model]["children"].append((layer_node[li], "layer_with_weights-%d" % n))
    trainable = []
    for n, li in enumerate(weighted):
        kname, _, what = all_layers[li]
        for var in (CONV_VARS if what == "conv" else BN_VARS):
            vn = new_node()
            nodes[layer_node[li]]["children"].append((vn, var))
            stem = "model/layer_with_weights-%d/%s" % (n, var)
            nodes[vn]["attrs"].append(("VARIABLE_VALUE", "%s/%s" % (kname, var), stem + ATTR))
            if not var.startswith("moving"):
                trainable.append((vn, stem, "%s/%s" % (kname, var)))
    for h in OPT_HYPER:
        vn = new_node()
        nodes[opt]["children"].append((vn, h))
        nodes[vn]["attrs"].append(("VARIABLE_VALUE", "Adam/" + h, "optimizer/%s%s" % (h, ATTR)))
    for vn, stem, full in trainable:
        for slot in ("m", "v"):
            sn = new_node()
            nodes[sn]["attrs"].append(("VARIABLE_VALUE", "Adam/%s/%s" % (full, slot), "%s/.OPTIMIZER_SLOT/optimizer/%s%s" % (stem, slot, ATTR)))
            nodes[opt]["slots"].append((vn, slot, sn))
    out = b""
    for nd in nodes:
        body = b""
        for nid, local in nd["children"]:
            body += pb_bytes(1, (pb_varint(1, nid) if nid else b"") + pb_bytes(2, local.encode()))
        for name, full, key in nd["attrs"]:
            body += pb_bytes(2, pb_bytes(1, name.encode()) + pb_bytes(2, full.encode()) + pb_bytes(3, key.encode()))
        for orig, slot, sn in nd["slots"]:
            body += pb_bytes(3, pb_varint(1, orig) + pb_bytes(2, slot.encode()) + pb_varint(3, sn))
        out += pb_bytes(1, body)
    return out


def decode_object_graph(buf):
    """-> list of nodes: dict(children={local_name: node_id}, attributes={name: checkpoint_key}, slots=[(orig, slot, node)])"""
    nodes = []
    for num, _, v in pb_parse(buf):
        if num != 1:
            continue
        nd = {"children": {}, "attributes": {}, "slots": []}
        for n2, _, v2 in pb_parse(v):
            f = {a: c for a, _, c in pb_parse(v2)}
            if n2 == 1:
                nd["children"][f.get(2, b"").decode()] = f.get(1, 0)
            elif n2 == 2:
                nd["attributes"][f.get(1, b"").decode()] = f.get(3, b"").decode()
            elif n2 == 3:
                nd["slots"].append((f.get(1, 0), f.get(2, b"").decode(), f.get(3, 0)))
        nodes.append(nd)
    return nodes


def resolve_keys_through_object_graph(graph_bytes, layers):
    """Walk the checkpoint's object graph the way TensorFlow's restore does -- root -> `model` -> `layer_with_weights-<n>` ->
    variable -> VARIABLE_VALUE, root -> `optimizer` -> hyper-parameters / slot variables -- and return
    {engine name or 'optimizer/...': checkpoint key}.  Works for bundles whose key strings differ from the canonical ones."""
    nodes = decode_object_graph(graph_bytes)
    root = nodes[0]["children"]
    out = {}
    var_node = {}
    if "model" in root:
        mch = nodes[root["model"]]["children"]
        all_layers, weighted = keras_graph(layers)
        for n, li in enumerate(weighted):
            _, eng, what = all_layers[li]
            ln = mch.get("layer_with_weights-%d" % n)
            if ln is None:
                continue
            for var in (CONV_VARS if what == "conv" else BN_VARS):
                vn = nodes[ln]["children"].get(var)
                if vn is not None and "VARIABLE_VALUE" in nodes[vn]["attributes"]:
                    eng_name = eng + "/" + {"moving_variance": "moving_var"}.get(var, var)
                    out[eng_name] = nodes[vn]["attributes"]["VARIABLE_VALUE"]
                    var_node[vn] = eng_name
    if "optimizer" in root:
        on = nodes[root["optimizer"]]
        for h in OPT_HYPER:
            vn = on["children"].get(h)
            if vn is not None and "VARIABLE_VALUE" in nodes[vn]["attributes"]:
                out["optimizer/" + h] = nodes[vn]["attributes"]["VARIABLE_VALUE"]
        for orig, slot, sn in on["slots"]:
            if orig in var_node and "VARIABLE_VALUE" in nodes[sn]["attributes"]:
                out["optimizer/%s/%s" % (slot, var_node[orig])] = nodes[sn]["attributes"]["VARIABLE_VALUE"]
    return out
