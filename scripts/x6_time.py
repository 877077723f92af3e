"""ms per launch of the BF16x6 forward kernel at a few BASELINE layer shapes (diagnostic builds via UNET_HIP_LIB): python scripts/x6_time.py"""
import ctypes, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
L = importlib.import_module("semantic-segmentation-unet_amd._lib").lib()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
ST = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
out = []
for shape in [(8, 64, 64, 512, 512), (8, 512, 512, 64, 64), (8, 512, 512, 128, 64), (8, 256, 256, 128, 128), (8, 32, 32, 1024, 1024)]:
    n, h, w, ci, co = shape
    x = torch.randn(n, h, w, ci, device="cuda"); wt = torch.randn(3, 3, ci, co, device="cuda") / float(np.sqrt(9 * ci))
    u = torch.empty(L.unet_winograd_x6_weight_bytes(ci, co), dtype=torch.uint8, device="cuda")
    L.unet_winograd_weight_transform_x6(P(wt), P(u), ci, co, 0, ST())
    y = torch.empty(n, h, w, co, device="cuda")
    f = lambda: L.unet_conv3x3_fwd_winograd_x6(P(x), ci, None, P(u), None, P(y), co, n, h, w, ci, co, 1, None, 0, ST())
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    out.append("%s %.3f" % (shape, e0.elapsed_time(e1) / 20))
print(" | ".join(out), flush=True)
