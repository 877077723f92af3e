"""Probe: the BF16x6 forward kernel under a profiler, with progress lines (python scripts/x6_prof_probe.py <logfile>)."""
import ctypes, importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
log = open(sys.argv[1], "a")
def say(*a):
    print(*a, file=log, flush=True)
say("start"); L = importlib.import_module("semantic-segmentation-unet_amd._lib").lib(); say("lib loaded")
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
ST = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for shape in [(2, 16, 16, 64, 64), (8, 64, 64, 512, 512)]:
    n, h, w, ci, co = shape
    x = torch.randn(n, h, w, ci, device="cuda"); wt = torch.randn(3, 3, ci, co, device="cuda") / float(np.sqrt(9 * ci))
    u = torch.empty(L.unet_winograd_x6_weight_bytes(ci, co), dtype=torch.uint8, device="cuda")
    L.unet_winograd_weight_transform_x6(P(wt), P(u), ci, co, 0, ST()); torch.cuda.synchronize(); say(shape, "weights done")
    y = torch.empty(n, h, w, co, device="cuda")
    t0 = time.time()
    L.unet_conv3x3_fwd_winograd_x6(P(x), ci, None, P(u), None, P(y), co, n, h, w, ci, co, 1, None, 0, ST()); torch.cuda.synchronize()
    say(shape, "x6 forward done in %.3f s, mean |y| %.4f" % (time.time() - t0, float(y.abs().mean())))
say("end")
