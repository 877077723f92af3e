"""Builds csrc/*.hip into the in-tree C-ABI library `libunet_hip.so` for gfx950 (hipcc cross-compiles without a GPU).

Staleness is decided by CONTENT, not by mtimes: every object file and the library carry a stamp = sha256 over the compiler
flags, the source text and the text of the headers it includes (`<name>.o.stamp`, `libunet_hip.so.stamp`).  A change of FLAGS,
a touched-but-identical file, or a pushed `.so` that is newer than edited sources can therefore neither force nor hide a
rebuild; `library_is_current()` lets a loader check that the binary it maps corresponds to the sources next to it."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libunet_hip.so")
SOURCES = ["conv_igemm.hip", "conv_wgrad.hip", "conv_direct.hip", "winograd.hip", "winograd_x6.hip", "convt_stream.hip", "convt_x6.hip", "conv_bf16.hip", "augment.hip",
           "norm.hip", "misc.hip", "crc32c.hip"]
HEADERS = ["common.h", "wino_epilogue.h"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
# per-source additions.  winograd_x6.hip: the split steps in the MFMA gaps are scalar fp32 subtractions on purpose -- SLP vectorisation turns the
# pairs (v0 - lo(h), v1 - hi(h)) of the round-to-nearest split into v_pk_add_f32, which stalls a one-wave-per-SIMD MFMA stream (+17 cycles per gap,
# profiles/r05_bf16_mfma_gap_costs.txt); the packed column stage of the end-of-tile code is written with vector types and is not affected
EXTRA_FLAGS = {"winograd_x6.hip": ["-fno-slp-vectorize"]}
LINK_FLAGS = ["--offload-arch=gfx950", "-shared"]


def _sha(*chunks):
    h = hashlib.sha256()
    for c in chunks:
        h.update(c if isinstance(c, bytes) else c.encode())
        h.update(b"\0")
    return h.hexdigest()


def source_stamp(src):
    hdr = [open(os.path.join(CSRC, h), "rb").read() for h in HEADERS]
    return _sha(" ".join(FLAGS + EXTRA_FLAGS.get(src, [])), open(os.path.join(CSRC, src), "rb").read(), *hdr)


def library_stamp():
    return _sha(" ".join(LINK_FLAGS), *[source_stamp(s) for s in SOURCES])


def _read(path):
    try:
        return open(path).read().strip()
    except OSError:
        return None


def library_is_current():
    """True when libunet_hip.so exists and was built from exactly the sources + flags in this tree."""
    return os.path.exists(LIB_PATH) and _read(LIB_PATH + ".stamp") == library_stamp()


def build_library(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        stamp = source_stamp(s)
        if force or not os.path.exists(obj) or _read(obj + ".stamp") != stamp:
            cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(s, []) + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
            open(obj + ".stamp", "w").write(stamp)
        objs.append(obj)
    if force or not library_is_current():
        cmd = [hipcc] + LINK_FLAGS + ["-o", LIB_PATH] + objs
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        open(LIB_PATH + ".stamp", "w").write(library_stamp())
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
