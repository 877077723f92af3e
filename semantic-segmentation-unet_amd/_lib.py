"""ctypes binding of the C-ABI library (include/unet_hip.h).  Prototypes are parsed from the header itself so the
Python side can never drift from the declared ABI.  There is NO fallback: if the HIP library is missing the product
path raises (build it with `python __graft_entry__.py` or `semantic-segmentation-unet_amd/_build.py`)."""
import ctypes
import os
import re

import torch  # noqa: F401  -- FIRST: PyTorch ships its own libamdhip64; loaded before this library, the dynamic linker binds
#               libunet_hip.so's HIP dependency to that same runtime.  The other order puts two HIP runtimes into the process
#               and the first kernel launch on a torch stream fails with hipErrorNoDevice (seen: build() then smoke() in one process).

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(HERE), "include", "unet_hip.h")
LIB_PATH = os.environ.get("UNET_HIP_LIB") or os.path.join(HERE, "csrc", "libunet_hip.so")     # override: diagnostics only

_DECL = re.compile(r"\b(int|size_t|uint32_t)\s+(unet_\w+)\s*\(([^)]*)\)\s*;", re.S)


def _ctype(decl):
    d = decl.strip()
    if d == "void" or not d:
        return None
    if "*" in d:
        return ctypes.c_void_p
    base = d.rsplit(" ", 1)[0].replace("const", "").strip() if " " in d else d
    return {"int": ctypes.c_int, "long": ctypes.c_long, "size_t": ctypes.c_size_t, "float": ctypes.c_float,
            "uint32_t": ctypes.c_uint32}[base]


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes])} for every function the header declares."""
    text = re.sub(r"/\*.*?\*/", "", open(path).read(), flags=re.S)
    out = {}
    for ret, name, args in _DECL.findall(text):
        at = [t for t in (_ctype(a) for a in args.split(",")) if t is not None]
        out[name] = ({"int": ctypes.c_int, "size_t": ctypes.c_size_t, "uint32_t": ctypes.c_uint32}[ret], at)
    return out


class UnetHipError(RuntimeError):
    pass


def header_abi_version(path=HEADER):
    m = re.search(r"#define\s+UNET_HIP_ABI_VERSION\s+(\d+)", open(path).read())
    if not m:
        raise UnetHipError("include/unet_hip.h does not define UNET_HIP_ABI_VERSION")
    return int(m.group(1))


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise UnetHipError("HIP extension %s is missing; run build() -- there is no CPU fallback" % LIB_PATH)
        if "UNET_HIP_LIB" not in os.environ:
            from . import _build
            if not _build.library_is_current():     # a binary that does not correspond to the sources next to it: refuse, loudly
                raise UnetHipError("%s was not built from the sources/flags in this tree (content stamp mismatch); "
                                   "run build() (python __graft_entry__.py)" % LIB_PATH)
        self.cdll = ctypes.CDLL(LIB_PATH)
        # ALWAYS (also for a UNET_HIP_LIB diagnostic build): the binary must speak the header's ABI version -- functions are bound by
        # name, so an older library would silently take arguments in the wrong slots
        want = header_abi_version()
        self.cdll.unet_hip_abi_version.restype = ctypes.c_int
        got = self.cdll.unet_hip_abi_version()
        if got != want:
            raise UnetHipError("%s speaks ABI version %d, include/unet_hip.h declares %d: rebuild it" % (LIB_PATH, got, want))
        self.protos = parse_header()
        for name, (res, args) in self.protos.items():
            fn = getattr(self.cdll, name)          # AttributeError = header/library mismatch: fail loudly
            fn.restype = res
            fn.argtypes = args
            if res is ctypes.c_int and name != "unet_hip_abi_version" and not name.endswith(("_supported", "_rows", "_rows_wg")):   # predicates / counts
                setattr(self, name, self._checked(name, fn))
            else:
                setattr(self, name, fn)

    @staticmethod
    def _checked(name, fn):
        def call(*a):
            rc = fn(*a)
            if rc != 0:
                raise UnetHipError("%s failed with status %d (%s)" % (
                    name, rc, {-1: "bad argument", -2: "workspace too small"}.get(rc, "hipError_t")))
        call.__name__ = name
        return call


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib
