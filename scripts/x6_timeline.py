"""Diagnostic: s_memtime phases of the BF16x6 forward kernel's units (build: scripts/build_variant.sh x6tl winograd_x6.hip "-DUNET_X6_ABLATE=8";
run with UNET_HIP_LIB=.../libunet_hip_x6tl.so).  Prints, per chunk of workgroup 0 / wave 0: stream + wait cycles of the four periods, the barrier; per tile: chunk loop and epilogue."""
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
    ch, tiles = max(t[9], 1), max(t[12], 1)
    print("%-26s chunks %4d tiles %3d | per chunk: " % (str(shape), ch, tiles) + " ".join("P%d %5.0f+%-5.0f" % (j, t[2 * j] / ch, t[2 * j + 1] / ch) for j in range(4))
          + " barrier %5.0f | per tile: loop %7.0f epilogue %6.0f (column stage + sends %5.0f, barrier %5.0f, combine %5.0f, finish %5.0f)" % (
              t[8] / ch, t[10] / tiles, t[11] / tiles, t[13] / tiles, t[14] / tiles, t[15] / tiles, (t[11] - t[13] - t[14] - t[15]) / tiles), flush=True)
