"""GPU micro-benchmark: the HBM-bound convolutions that are not GEMM-shaped (first 3x3 layer, 1x1 class map) at config 2."""
import ctypes, os, sys, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
L = importlib.import_module("semantic-segmentation-unet_amd._lib").lib()
P = lambda t: ctypes.c_void_p(t.data_ptr())
ST = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
B, H, C, K = 8, 512, int(os.environ.get("C", 1)), int(os.environ.get("K", 2))
npx = B * H * H
x1 = torch.randn(B, H, H, C, device="cuda"); w1 = torch.randn(3, 3, C, 64, device="cuda"); b1 = torch.randn(64, device="cuda")
y = torch.empty(B, H, H, 64, device="cuda"); dz = torch.randn(B, H, H, 64, device="cuda"); dw1 = torch.empty_like(w1)
nb = L.unet_conv3x3_wgrad_direct_workspace(B, H, H, C, 64); ws = torch.empty(nb + 256, dtype=torch.uint8, device="cuda")
big = npx * 64 * 4.0
t = timeit(lambda: L.unet_conv3x3_fwd_direct(P(x1), C, P(w1), P(b1), P(y), 64, B, H, H, C, 64, 1, ST()))
print("conv3x3 direct fwd   %dch->64   %6.3f ms  %5.2f TB/s (output write)" % (C, t, big / t / 1e9))
t = timeit(lambda: L.unet_conv3x3_wgrad_direct(P(x1), C, P(dz), 64, 0, P(dw1), B, H, H, C, 64, P(ws), nb, ST()))
print("conv3x3 direct wgrad %dch->64   %6.3f ms  %5.2f TB/s (dz read x Cin)" % (C, t, C * big / t / 1e9))
wk = torch.randn(64, K, device="cuda"); bk = torch.randn(K, device="cuda"); z = torch.empty(B, H, H, K, device="cuda"); dzk = torch.randn(B, H, H, K, device="cuda")
dx = torch.empty(B, H, H, 64, device="cuda"); dwk = torch.empty_like(wk)
nb2 = L.unet_conv1x1_wgrad_workspace(npx, 64, K); ws2 = torch.empty(nb2 + 256, dtype=torch.uint8, device="cuda")
t = timeit(lambda: L.unet_conv1x1_fwd(P(y), 64, 0, P(wk), P(bk), P(z), K, npx, 64, K, 1, ST()))
print("conv1x1 fwd   64->%d           %6.3f ms  %5.2f TB/s (input read)" % (K, t, big / t / 1e9))
t = timeit(lambda: L.unet_conv1x1_dgrad(P(dzk), K, P(wk), P(dx), 64, 0, npx, 64, K, ST()))
print("conv1x1 dgrad 64->%d           %6.3f ms  %5.2f TB/s (dx write)" % (K, t, big / t / 1e9))
t = timeit(lambda: L.unet_conv1x1_wgrad(P(y), 64, 0, P(dzk), K, P(dwk), npx, 64, K, P(ws2), nb2, ST()))
print("conv1x1 wgrad 64->%d           %6.3f ms  %5.2f TB/s (input read)" % (K, t, big / t / 1e9))
