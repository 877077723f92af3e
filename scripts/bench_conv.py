"""GPU micro-benchmark: per-layer TFLOP/s of the fp32-MFMA conv kernels at the BASELINE config-2 layer shapes."""
import ctypes, os, sys, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
L = importlib.import_module("semantic-segmentation-unet_amd._lib").lib()
P = lambda t: ctypes.c_void_p(t.data_ptr())
ST = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B = int(os.environ.get("B", 8))
shapes = [("1b", 512, 64, 64), ("2a", 256, 64, 128), ("2b", 256, 128, 128), ("3b", 128, 256, 256), ("4b", 64, 512, 512),
          ("bott_b", 32, 1024, 1024), ("dec_4a", 64, 1024, 512), ("dec_1a", 512, 128, 64)]
which = sys.argv[1:] or ["fwd", "dgrad", "wgrad"]
reps = 5
def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
tot = {k: [0.0, 0.0] for k in which}
for name, h, ci, co in shapes:
    x = torch.randn(B, h, h, ci, device="cuda"); dz = torch.randn(B, h, h, co, device="cuda")
    w = torch.randn(3, 3, ci, co, device="cuda") * 0.05; b = torch.randn(co, device="cuda")
    out = torch.empty(B, h, h, co, device="cuda"); dx = torch.empty(B, h, h, ci, device="cuda"); dw = torch.empty_like(w)
    nb = L.unet_conv3x3_wgrad_mfma_workspace(B, h, h, ci, co); ws = torch.empty(nb + 256, dtype=torch.uint8, device="cuda")
    fl = 2.0 * 9 * B * h * h * ci * co
    wok = wwok = False          # (the unfused Winograd pipeline was removed in round 5: git history has it)
    Uc = torch.empty(16 * ci * co, device="cuda"); Ucd = torch.empty(16 * ci * co, device="cuda")
    L.unet_winograd_weight_transform(P(w), P(Uc), ci, co, 2, ST()); L.unet_winograd_weight_transform(P(w), P(Ucd), ci, co, 3, ST())
    nbq = L.unet_conv3x3_wgrad_winograd_fused_workspace(B, h, h, ci, co, 0); wsq = torch.empty(nbq + 256, dtype=torch.uint8, device="cuda")
    fns = {"fwgrad": lambda: L.unet_conv3x3_wgrad_winograd_fused(P(x), ci, P(dz), co, P(dw), B, h, h, ci, co, 0, P(wsq), nbq, ST()),
           "ffwd": lambda: L.unet_conv3x3_fwd_winograd_fused(P(x), ci, None, P(Uc), P(b), P(out), co, B, h, h, ci, co, 1, None, 0, ST()),
           "fdgrad": lambda: L.unet_conv3x3_dgrad_winograd_fused(P(dz), co, P(Ucd), P(dx), ci, B, h, h, ci, co, None, 0, 0, 0, None, 0, ST()),
           "fwd": lambda: L.unet_conv3x3_fwd_mfma(P(x), ci, P(w), P(b), P(out), co, B, h, h, ci, co, 1, ST()),
           "dgrad": lambda: L.unet_conv3x3_dgrad_mfma(P(dz), co, P(w), P(dx), ci, B, h, h, ci, co, ST()),
           "wgrad": lambda: L.unet_conv3x3_wgrad_mfma(P(x), ci, P(dz), co, P(dw), B, h, h, ci, co, P(ws), nb, ST())}
    line = "%-7s h%4d %4d->%4d " % (name, h, ci, co)
    for k in which:
        if k in ("wfwd", "wdgrad") and not wok:
            continue
        if k == "wwgrad" and not wwok:
            continue
        ms = timeit(fns[k]); tot[k][0] += ms; tot[k][1] += fl
        line += " %s %7.3f ms %6.1f TF |" % (k, ms, fl / ms / 1e9)
    print(line, flush=True)
    del x, dz, out, dx, ws
print("TOTAL " + " ".join("%s %.2f ms %.1f TF" % (k, v[0], v[1] / v[0] / 1e9) for k, v in tot.items()))
