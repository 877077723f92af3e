"""Diagnostic: s_memtime phases of the BF16x6 forward kernel's units (build: scripts/build_variant.sh x6tl winograd_x6.hip "-DUNET_X6_ABLATE=8";
run with UNET_HIP_LIB=.../libunet_hip_x6tl.so).  Prints, per unit of workgroup 0 / wave 0: operand wait, MFMA stream, DMA/LDS wait, barrier."""
import ctypes, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
L = importlib.import_module("semantic-segmentation-unet_amd._lib").lib()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
ST = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for shape in [(8, 64, 64, 512, 512), (8, 512, 512, 64, 64), (8, 32, 32, 1024, 1024), (8, 256, 256, 128, 128)]:
    n, h, w, ci, co = shape
    x = torch.randn(n, h, w, ci, device="cuda"); wt = torch.randn(3, 3, ci, co, device="cuda") / float(np.sqrt(9 * ci))
    u = torch.empty(L.unet_winograd_x6_weight_bytes(ci, co), dtype=torch.uint8, device="cuda")
    L.unet_winograd_weight_transform_x6(P(wt), P(u), ci, co, 0, ST())
    y = torch.empty(n, h, w, co, device="cuda")
    for _ in range(3):
        L.unet_conv3x3_fwd_winograd_x6(P(x), ci, None, P(u), None, P(y), co, n, h, w, ci, co, 1, None, 0, ST())
    torch.cuda.synchronize()
    out = (ctypes.c_longlong * 16)()
    L.cdll.unet_debug_x6_timeline(out)
    t = list(out)
    if "prev" in dir():
        d = [a - b for a, b in zip(t, prev)]
    else:
        d = t
    prev = t
    units = max(d[8], 1) / 2.0          # units per group in the MFMA role (each group has the role in half of the units)
    print("%-26s per unit (cycles): group0 mfma stream %5.0f + wait %5.0f | transform %5.0f + wait %5.0f || group1 mfma %5.0f + %5.0f | transform %5.0f + %5.0f" % (
        str(shape), d[0] / units, d[1] / units, d[2] / units, d[3] / units, d[4] / units, d[5] / units, d[6] / units, d[7] / units), flush=True)
