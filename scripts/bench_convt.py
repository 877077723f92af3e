"""GPU micro-benchmark: the 2x2/stride-2 transposed-conv kernels at the BASELINE config-2 up-sampling shapes."""
import ctypes, os, sys, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
L = importlib.import_module("semantic-segmentation-unet_amd._lib").lib()
P = lambda t: ctypes.c_void_p(t.data_ptr())
ST = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B = 8
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
tot = [0.0, 0.0, 0.0]
for name, h, ci, co in [("up4", 32, 1024, 512), ("up3", 64, 512, 256), ("up2", 128, 256, 128), ("up1", 256, 128, 64)]:
    x = torch.randn(B, h, h, ci, device="cuda"); w = torch.randn(2, 2, co, ci, device="cuda") * 0.05; b = torch.randn(co, device="cuda")
    out = torch.empty(B, 2 * h, 2 * h, co, device="cuda"); dz = torch.randn(B, 2 * h, 2 * h, co, device="cuda")
    dx = torch.empty_like(x); dw = torch.empty_like(w)
    nb = L.unet_convT2x2_wgrad_workspace(B, h, h, ci, co); ws = torch.empty(nb + 256, dtype=torch.uint8, device="cuda")
    fl = 2.0 * 4 * B * h * h * ci * co
    t0 = timeit(lambda: L.unet_convT2x2_fwd(P(x), ci, P(w), P(b), P(out), co, B, h, h, ci, co, ST()))
    t1 = timeit(lambda: L.unet_convT2x2_dgrad(P(dz), co, P(w), P(dx), ci, B, h, h, ci, co, ST()))
    t2 = timeit(lambda: L.unet_convT2x2_wgrad(P(x), ci, P(dz), co, P(dw), B, h, h, ci, co, P(ws), nb, ST()))
    if L.unet_convT2x2_fwd_stream_supported(B, h, h, ci, co) == 1:
        t3 = timeit(lambda: L.unet_convT2x2_fwd_stream(P(x), ci, P(w), P(b), P(out), co, B, h, h, ci, co, ST()))
        print("        fwd (stream kernel) %6.3f ms %6.1f TF" % (t3, fl / t3 / 1e9))
    if L.unet_convT2x2_bf16_supported(B, h, h, ci, co) == 1:
        nbp = L.unet_convT2x2_bf16_packed_bytes(ci, co)
        wp = torch.empty(nbp, dtype=torch.uint8, device="cuda"); wpd = torch.empty(nbp, dtype=torch.uint8, device="cuda")
        L.unet_convT2x2_bf16_pack_weights(P(w), P(wp), ci, co, 0, ST()); L.unet_convT2x2_bf16_pack_weights(P(w), P(wpd), ci, co, 1, ST())
        t5 = timeit(lambda: L.unet_convT2x2_fwd_bf16(P(x), ci, 0, P(wp), P(b), P(out), co, 0, B, h, h, ci, co, None, 0, ST()))
        t6 = timeit(lambda: L.unet_convT2x2_dgrad_bf16(P(dz), co, 0, P(wpd), P(dx), ci, 0, B, h, h, ci, co, None, 0, 0, None, 0, ST()))
        t7 = float("nan")
        if L.unet_convT2x2_wgrad_bf16_supported(B, h, h, ci, co) == 1:
            nbw2 = L.unet_convT2x2_wgrad_bf16_workspace(B, h, h, ci, co); wsw2 = torch.empty(nbw2 + 256, dtype=torch.uint8, device="cuda")
            t7 = timeit(lambda: L.unet_convT2x2_wgrad_bf16(P(x), ci, 0, P(dz), co, 0, P(dw), B, h, h, ci, co, P(wsw2), nbw2, ST()))
        print("        bf16 kernels: fwd %6.3f ms %6.1f TF | dgrad %6.3f ms %6.1f TF | wgrad %6.3f ms %6.1f TF" % (t5, fl / t5 / 1e9, t6, fl / t6 / 1e9, t7, fl / t7 / 1e9))
    tot[0] += t0; tot[1] += t1; tot[2] += t2
    print("%-4s h%4d %4d->%4d  fwd %6.3f ms %6.1f TF | dgrad %6.3f ms %6.1f TF | wgrad %6.3f ms %6.1f TF" % (name, h, ci, co, t0, fl / t0 / 1e9, t1, fl / t1 / 1e9, t2, fl / t2 / 1e9), flush=True)
print("TOTAL fwd %.2f ms  dgrad %.2f ms  wgrad %.2f ms" % tuple(tot))
