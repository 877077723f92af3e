"""Diagnostic: prologue / chunk loop / epilogue cycles of one workgroup of the fused Winograd forward kernel
(needs a -DUNET_ABLATE=8 build passed via UNET_HIP_LIB)."""
import ctypes, os, sys, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
L = importlib.import_module("semantic-segmentation-unet_amd._lib").lib()
raw = ctypes.CDLL(os.environ["UNET_HIP_LIB"])
P = lambda t: ctypes.c_void_p(t.data_ptr())
ST = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, h, ci, co in [("1b", 512, 64, 64), ("2b", 256, 128, 128), ("4b", 64, 512, 512), ("dec_1a", 512, 128, 64)]:
    B = 8
    x = torch.randn(B, h, h, ci, device="cuda"); w = torch.randn(3, 3, ci, co, device="cuda") * 0.05; b = torch.randn(co, device="cuda")
    out = torch.empty(B, h, h, co, device="cuda"); Uc = torch.empty(16 * ci * co, device="cuda")
    L.unet_winograd_weight_transform(P(w), P(Uc), ci, co, 2, ST())
    for _ in range(3):
        L.unet_conv3x3_fwd_winograd_fused(P(x), ci, None, P(Uc), P(b), P(out), co, B, h, h, ci, co, 1, None, 0, ST())
    torch.cuda.synchronize()
    t = (ctypes.c_longlong * 8)()
    raw.unet_debug_wf_timeline(t)
    n = max(t[3], 1)
    print("%-7s chunks %4d | cycles: prologue-or-gap %6d  loop %7d (%5.0f per chunk)  epilogue %6d | loop share %.2f"
          % (name, n, t[0], t[1], t[1] / n, t[2], t[1] / float(t[0] + t[1] + t[2])))
    print("        per chunk: DMA issue %5.0f | operand reads + transform + V writes %5.0f | 64 MFMAs + reads %5.0f | wait + barrier %5.0f"
          % (t[4] / n, t[5] / n, t[6] / n, t[7] / n))
