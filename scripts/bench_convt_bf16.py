"""GPU micro-benchmark: the bf16 transposed-conv kernels as the mixed-precision step calls them (bf16-stored operands, BatchNorm sums fused) at the
BASELINE up-sampling shapes, with each launch's floors: HBM bytes at 5.5 TB/s and matrix time at half the bf16 peak."""
import ctypes, os, sys, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
L = importlib.import_module("semantic-segmentation-unet_amd._lib").lib()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
ST = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B = 8; bf = torch.bfloat16
def timeit(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
tot = [0.0, 0.0, 0.0]
for name, h, ci, co in [("up4", 32, 1024, 512), ("up3", 64, 512, 256), ("up2", 128, 256, 128), ("up1", 256, 128, 64)]:
    x = torch.randn(B, h, h, ci, device="cuda").to(bf); w = torch.randn(2, 2, co, ci, device="cuda") * 0.05; b = torch.randn(co, device="cuda")
    out = torch.empty(B, 2 * h, 2 * h, co, device="cuda", dtype=bf); dz = torch.randn(B, 2 * h, 2 * h, co, device="cuda").to(bf)
    dx = torch.empty_like(x); dw = torch.empty_like(w); r = torch.randn(B, h, h, ci, device="cuda").to(bf)
    nbp = L.unet_convT2x2_bf16_packed_bytes(ci, co)
    wp = torch.empty(nbp, dtype=torch.uint8, device="cuda"); wpd = torch.empty(nbp, dtype=torch.uint8, device="cuda")
    L.unet_convT2x2_bf16_pack_weights(P(w), P(wp), ci, co, 0, ST()); L.unet_convT2x2_bf16_pack_weights(P(w), P(wpd), ci, co, 1, ST())
    rows_f = L.unet_convT2x2_bf16_stats_rows(B, h, h, ci, co, 0); rows_d = L.unet_convT2x2_bf16_stats_rows(B, h, h, ci, co, 1)
    pf = torch.empty((co // 64) * rows_f * 128, device="cuda"); pd = torch.empty((ci // 64) * rows_d * 128, device="cuda")
    fl = 2.0 * 4 * B * h * h * ci * co
    tf = timeit(lambda: L.unet_convT2x2_fwd_bf16(P(x), ci, 1, P(wp), P(b), P(out), co, 1, B, h, h, ci, co, P(pf), pf.numel() * 4, ST()))
    td = timeit(lambda: L.unet_convT2x2_dgrad_bf16(P(dz), co, 1, P(wpd), P(dx), ci, 1, B, h, h, ci, co, P(r), ci, 1, P(pd), pd.numel() * 4, ST()))
    nbw = L.unet_convT2x2_wgrad_bf16_workspace(B, h, h, ci, co); wsw = torch.empty(nbw + 256, dtype=torch.uint8, device="cuda")
    tw = timeit(lambda: L.unet_convT2x2_wgrad_bf16(P(x), ci, 1, P(dz), co, 1, P(dw), B, h, h, ci, co, P(wsw), nbw, ST()))
    bx, bo = x.numel() * 2, out.numel() * 2
    floor = lambda by: max(by / 5.5e12, fl / 1.25e15) * 1e3
    print("%-4s %4d^2 %4d->%4d | fwd+sums %6.3f ms (floor %5.3f) | dgrad+sums %6.3f ms (floor %5.3f) | wgrad %6.3f ms (floor %5.3f) | lib=%s"
          % (name, h, ci, co, tf, floor(bx + bo), td, floor(2 * bx + bo), tw, floor(bx + bo), os.path.basename(os.environ.get("UNET_HIP_LIB", "default"))), flush=True)
    tot[0] += tf; tot[1] += td; tot[2] += tw
print("TOTAL fwd %.3f  dgrad %.3f  wgrad %.3f ms" % tuple(tot))
