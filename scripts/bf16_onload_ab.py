"""A/B for BatchNorm-apply on load in the bf16 route (review item 2a), per single-consumer layer of BASELINE config 4 (512x512, batch 8):
  today     = BatchNorm apply pass (r -> y, bf16) + persistent DMA-staged forward on y + DMA-staged weight gradient on y
  on load   = forward with in_scale / in_shift on r (the register-staged per-tile kernel: the persistent kernel stages by LDS-DMA, which cannot
              apply a scale) + a weight gradient that normalises while staging -- no such kernel exists; its LOWER bound is today's
              register-staged weight gradient (no FMA added), timed from a -DUNET_WGRAD_BF16_NO_DMA build
usage: python scripts/bf16_onload_ab.py          (regular library: columns apply / fwd stream / fwd on-load / wgrad dma)
       UNET_HIP_LIB=.../libunet_hip_wgreg.so python scripts/bf16_onload_ab.py wgreg    (adds nothing but prints the register-staged wgrad column)"""
import ctypes, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
L = importlib.import_module("semantic-segmentation-unet_amd._lib").lib()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
ST = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B, BF = 8, torch.bfloat16
wgreg = len(sys.argv) > 1 and sys.argv[1] == "wgreg"


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


# consumer layers whose input is one producer's BatchNorm output and nothing else (plan.CONSUMER): (name, H, Cin, Cout)
LAYERS = [("conv_1b", 512, 64, 64), ("conv_2b", 256, 128, 128), ("conv_3b", 128, 256, 256), ("conv_4b", 64, 512, 512), ("bott_b", 32, 1024, 1024),
          ("dec_4b", 64, 512, 512), ("dec_3b", 128, 256, 256), ("dec_2b", 256, 128, 128), ("dec_1b", 512, 64, 64)]
print("# ms per launch, one MI355X, batch 8; %s" % ("register-staged weight gradient build" if wgreg else "regular build"))
print("%-8s %5s %5s | %8s %10s %11s | %9s" % ("layer", "H", "C", "bn_apply", "fwd stream", "fwd on-load", "wgrad reg" if wgreg else "wgrad dma"))
tot = [0.0] * 4
for name, h, ci, co in LAYERS:
    g = torch.Generator(device="cuda").manual_seed(h + ci)
    r = torch.randn(B, h, h, ci, device="cuda", generator=g).to(BF); y = torch.empty_like(r)
    sc = torch.rand(ci, device="cuda", generator=g) + 0.5; sh = torch.randn(ci, device="cuda", generator=g)
    w = torch.randn(3, 3, ci, co, device="cuda", generator=g) * 0.05; b = torch.randn(co, device="cuda", generator=g)
    out = torch.empty(B, h, h, co, device="cuda", dtype=BF)
    wp = torch.empty(L.unet_conv3x3_bf16_packed_bytes(ci, co), dtype=torch.uint8, device="cuda")
    L.unet_conv3x3_bf16_pack_weights(P(w), P(wp), ci, co, 0, ST())
    rows = L.unet_conv3x3_bf16_stats_rows(B, h, h, ci, co)
    part = torch.empty((co // 64) * rows * 128, device="cuda")
    t_apply = timeit(lambda: L.unet_bn_apply_any(P(r), ci, 1, P(sc), P(sh), P(y), ci, 1, None, 0, None, B, h, h, ci, ST()))
    t_stream = timeit(lambda: L.unet_conv3x3_fwd_bf16(P(y), ci, 1, None, None, P(wp), P(b), P(out), co, 1, B, h, h, ci, co, 1, P(part), part.numel() * 4, ST()))
    t_onload = timeit(lambda: L.unet_conv3x3_fwd_bf16(P(r), ci, 1, P(sc), P(sh), P(wp), P(b), P(out), co, 1, B, h, h, ci, co, 1, P(part), part.numel() * 4, ST()))
    dz = torch.randn(B, h, h, co, device="cuda", generator=g).to(BF); dw = torch.empty_like(w)
    nbw = L.unet_conv3x3_wgrad_bf16_workspace(B, h, h, ci, co); ws = torch.empty(nbw + 256, dtype=torch.uint8, device="cuda")
    t_wg = timeit(lambda: L.unet_conv3x3_wgrad_bf16(P(y), ci, 1, P(dz), co, 1, P(dw), B, h, h, ci, co, P(ws), nbw, ST()))
    for i, t in enumerate((t_apply, t_stream, t_onload, t_wg)):
        tot[i] += t
    print("%-8s %5d %5d | %8.3f %10.3f %11.3f | %9.3f" % (name, h, ci, t_apply, t_stream, t_onload, t_wg), flush=True)
print("%-8s %11s | %8.3f %10.3f %11.3f | %9.3f" % ("TOTAL", "", *tot))
print("today: apply + stream = %.3f ms; on load: %.3f ms (forward only; the weight gradient column of the other build is added in profiles/r04_bf16_apply_on_load_ab.txt)" % (tot[0] + tot[1], tot[2]))
