"""GPU micro-benchmark: the bf16 matrix-core 3x3 convolution at the BASELINE 512x512 batch-8 layer shapes (forward kernel;
the data gradient is the same kernel).  Prints time, TFLOP/s and the fp32 activation bytes moved per second."""
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
tot = 0.0
for name, h, ci, co in [("1b", 512, 64, 64), ("2a", 256, 64, 128), ("2b", 256, 128, 128), ("3b", 128, 256, 256), ("4b", 64, 512, 512),
                        ("bott_a", 32, 512, 1024), ("bott_b", 32, 1024, 1024), ("dec4a", 64, 1024, 512), ("dec1a", 512, 128, 64)]:
    x = torch.randn(B, h, h, ci, device="cuda"); w = torch.randn(3, 3, ci, co, device="cuda") * 0.05; b = torch.randn(co, device="cuda")
    out = torch.empty(B, h, h, co, device="cuda")
    wp = torch.empty(L.unet_conv3x3_bf16_packed_bytes(ci, co), dtype=torch.uint8, device="cuda")
    tp = timeit(lambda: L.unet_conv3x3_bf16_pack_weights(P(w), P(wp), ci, co, 0, ST()))
    t = timeit(lambda: L.unet_conv3x3_fwd_bf16(P(x), ci, 0, None, None, P(wp), P(b), P(out), co, 0, B, h, h, ci, co, 1, None, 0, ST()))
    fl = 2.0 * 9 * B * h * h * ci * co
    dz = torch.randn(B, h, h, co, device="cuda"); dw = torch.empty_like(w)
    nbw = L.unet_conv3x3_wgrad_bf16_workspace(B, h, h, ci, co); wsw = torch.empty(nbw + 256, dtype=torch.uint8, device="cuda")
    tw = timeit(lambda: L.unet_conv3x3_wgrad_bf16(P(x), ci, 0, P(dz), co, 0, P(dw), B, h, h, ci, co, P(wsw), nbw, ST()))
    print("        wgrad %7.3f ms %7.1f TF" % (tw, fl / tw / 1e9))
    by = 4.0 * B * h * h * (ci + co)
    tot += t
    print("%-7s h%4d %4d->%4d  %7.3f ms %7.1f TF  %6.2f TB/s (activations in+out) | pack %6.3f ms" % (name, h, ci, co, t, fl / t / 1e9, by / t / 1e9, tp), flush=True)
print("TOTAL %.2f ms" % tot)
