"""GPU micro-benchmark: the bf16 3x3 kernels on the full-resolution (few-channel) layers with bf16-stored operands, the forms the
training step launches: forward + BatchNorm sums (bf16 in / bf16 out), data gradient + producer BatchNorm-backward sums (bf16 dz in,
bf16 dx out, bf16 saved activation).  UNET_HIP_LIB selects a diagnostic build (scripts/build_variant.sh)."""
import ctypes, os, sys, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
L = importlib.import_module("semantic-segmentation-unet_amd._lib").lib()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
ST = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B = 8
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
bf = torch.bfloat16
LAYERS = [("1b", 512, 64, 64), ("dec1a", 512, 128, 64), ("2b", 256, 128, 128), ("dec2a", 256, 256, 128), ("3b", 128, 256, 256), ("4b", 64, 512, 512)]
if "--all" in sys.argv:          # every MFMA 3x3 layer of the 512^2 network (x count in the step)
    LAYERS = [("1b/dec1b x2", 512, 64, 64), ("2a", 256, 64, 128), ("2b/dec2b x2", 256, 128, 128), ("3a", 128, 128, 256), ("3b/dec3b x2", 128, 256, 256),
              ("4a", 64, 256, 512), ("4b/dec4b x2", 64, 512, 512), ("bott_a", 32, 512, 1024), ("bott_b", 32, 1024, 1024), ("dec4a", 64, 1024, 512),
              ("dec3a", 128, 512, 256), ("dec2a", 256, 256, 128), ("dec1a", 512, 128, 64)]
for name, h, ci, co in LAYERS:
    x = torch.randn(B, h, h, ci, device="cuda").to(bf); w = torch.randn(3, 3, ci, co, device="cuda") * 0.05; b = torch.randn(co, device="cuda")
    out = torch.empty(B, h, h, co, device="cuda", dtype=bf)
    wp = torch.empty(L.unet_conv3x3_bf16_packed_bytes(ci, co), dtype=torch.uint8, device="cuda"); wpd = torch.empty_like(wp)
    L.unet_conv3x3_bf16_pack_weights(P(w), P(wp), ci, co, 0, ST()); L.unet_conv3x3_bf16_pack_weights(P(w), P(wpd), ci, co, 1, ST())
    rows = L.unet_conv3x3_bf16_stats_rows(B, h, h, ci, co)
    part = torch.empty((co // 64) * rows * 128, device="cuda")
    tf = timeit(lambda: L.unet_conv3x3_fwd_bf16(P(x), ci, 1, None, None, P(wp), P(b), P(out), co, 1, B, h, h, ci, co, 1, P(part), part.numel() * 4, ST()))
    tf0 = timeit(lambda: L.unet_conv3x3_fwd_bf16(P(x), ci, 1, None, None, P(wp), P(b), P(out), co, 1, B, h, h, ci, co, 1, None, 0, ST()))
    sc = torch.rand(ci, device="cuda") + 0.5; sh = torch.randn(ci, device="cuda"); y16 = torch.empty_like(x)
    tn = timeit(lambda: L.unet_conv3x3_fwd_bf16(P(x), ci, 1, P(sc), P(sh), P(wp), P(b), P(out), co, 1, B, h, h, ci, co, 1, P(part), part.numel() * 4, ST()))
    ta = timeit(lambda: L.unet_bn_apply_any(P(x), ci, 1, P(sc), P(sh), P(y16), ci, 1, None, 0, None, B, h, h, ci, ST()))
    print("        BatchNorm apply on load: fwd+stats %6.3f ms  vs  separate apply %6.3f + fwd %6.3f = %6.3f ms" % (tn, ta, tf, ta + tf))
    dz = torch.randn(B, h, h, co, device="cuda").to(bf); dx = torch.empty(B, h, h, ci, device="cuda", dtype=bf)
    rp = torch.randn(B, h, h, ci, device="cuda").to(bf)
    rows2 = L.unet_conv3x3_bf16_stats_rows(B, h, h, co, ci)
    part2 = torch.empty((ci // 64) * rows2 * 128, device="cuda")
    td = timeit(lambda: L.unet_conv3x3_dgrad_bf16(P(dz), co, 1, P(wpd), P(dx), ci, 1, B, h, h, ci, co, P(rp), ci, 1, 0, ci, P(part2), part2.numel() * 4, ST()))
    td0 = timeit(lambda: L.unet_conv3x3_dgrad_bf16(P(dz), co, 1, P(wpd), P(dx), ci, 1, B, h, h, ci, co, None, 0, 0, 0, 0, None, 0, ST()))
    nbw = L.unet_conv3x3_wgrad_bf16_workspace(B, h, h, ci, co); wsw = torch.empty(nbw + 256, dtype=torch.uint8, device="cuda"); dw = torch.empty_like(w)
    tw = timeit(lambda: L.unet_conv3x3_wgrad_bf16(P(x), ci, 1, P(dz), co, 1, P(dw), B, h, h, ci, co, P(wsw), nbw, ST()))
    fl = 2.0 * 9 * B * h * h * ci * co
    print("        wgrad (bf16 x, bf16 dz) %6.3f ms (%5.0f TF)" % (tw, fl / tw / 1e9))
    byf = 2.0 * B * h * h * (ci + co)
    print("%-6s h%4d %4d->%4d | fwd+stats %6.3f ms (%5.0f TF, %4.2f TB/s) plain %6.3f | dgrad+bnbwd %6.3f ms (%5.0f TF, %4.2f TB/s) plain %6.3f"
          % (name, h, ci, co, tf, fl / tf / 1e9, byf / tf / 1e9, tf0, td, fl / td / 1e9, (byf + 2.0 * B * h * h * ci) / td / 1e9, td0), flush=True)
