"""Diagnostic: phase timeline of the fused Winograd weight-gradient kernel (needs a -DUNET_ABLATE=7 build via UNET_HIP_LIB)."""
import ctypes, os, sys, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
L = importlib.import_module("semantic-segmentation-unet_amd._lib").lib()
P = lambda t: ctypes.c_void_p(t.data_ptr())
ST = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, h, ci, co in [("bott_b", 32, 1024, 1024), ("4b", 64, 512, 512), ("2b", 256, 128, 128)]:
    B = 8
    x = torch.randn(B, h, h, ci, device="cuda"); dz = torch.randn(B, h, h, co, device="cuda"); dw = torch.empty(3, 3, ci, co, device="cuda")
    nb = L.unet_conv3x3_wgrad_winograd_fused_workspace(B, h, h, ci, co, 0)
    ws = torch.zeros(nb + 256, dtype=torch.uint8, device="cuda")
    for _ in range(3):
        L.unet_conv3x3_wgrad_winograd_fused(P(x), ci, P(dz), co, P(dw), B, h, h, ci, co, 0, P(ws), nb, ST())
    torch.cuda.synchronize()
    t = ws[nb:nb + 40].view(torch.int64).tolist()
    n = max(t[4], 1)
    print("%-7s iterations %4d | cycles per iteration: LDS-wait %6.0f  transforms + DMA offsets (VALU block) %6.0f  vmcnt+barrier %6.0f  64 MFMAs with reads, DMAs and scalar stages behind them %6.0f | total %6.0f"
          % (name, n, t[0] / n, t[1] / n, t[2] / n, t[3] / max(n - 1, 1), (t[0] + t[1] + t[2]) / n + t[3] / max(n - 1, 1)))
