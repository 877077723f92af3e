"""BF16x6 transposed-conv forward / data gradient (csrc/convt_x6.hip) against the native fp32-MFMA kernels on one MI355X: error against
torch's fp64 transposed convolution and ms per launch at the BASELINE config-2 up-sampling layers.  usage: python scripts/convt_x6_check.py"""
import ctypes, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
L = importlib.import_module("semantic-segmentation-unet_amd._lib").lib()
DEV = "cuda"
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
ST = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def errs(a, r):
    d = a.double() - r
    return float(d.abs().max() / r.abs().max()), float(d.pow(2).mean().sqrt() / r.pow(2).mean().sqrt())


print("device:", torch.cuda.get_device_name(0), flush=True)
for (n, h, ci, co, timed) in [(2, 8, 128, 64, False), (1, 16, 256, 128, False), (4, 8, 128, 192, False),
                              (8, 32, 1024, 512, True), (8, 64, 512, 256, True), (8, 128, 256, 128, True), (8, 256, 128, 64, True)]:
    w_ = h
    g = torch.Generator(device=DEV).manual_seed(h + ci)
    x = torch.randn(n, h, w_, ci, device=DEV, generator=g)
    wT = torch.randn(2, 2, co, ci, device=DEV, generator=g) / float(np.sqrt(ci))
    b = torch.randn(co, device=DEV, generator=g)
    dz = torch.randn(n, 2 * h, 2 * w_, co, device=DEV, generator=g)
    ref = torch.nn.functional.conv_transpose2d(x.double().permute(0, 3, 1, 2), wT.double().permute(3, 2, 0, 1), b.double(), stride=2).permute(0, 2, 3, 1)
    refd = torch.nn.functional.conv2d(dz.double().permute(0, 3, 1, 2), wT.double().permute(3, 2, 0, 1), None, stride=2).permute(0, 2, 3, 1)
    assert L.unet_convT2x2_x6_supported(n, h, w_, ci, co) == 1
    nb = L.unet_convT2x2_x6_weight_bytes(ci, co)
    W6 = torch.empty(nb, dtype=torch.uint8, device=DEV); W6d = torch.empty(nb, dtype=torch.uint8, device=DEV)
    L.unet_convT2x2_weight_transform_x6(P(wT), P(W6), ci, co, 0, ST()); L.unet_convT2x2_weight_transform_x6(P(wT), P(W6d), ci, co, 1, ST())
    rows = L.unet_convT2x2_x6_stats_rows(n, h, w_, ci, co)
    part = torch.full(((co // 64) * rows * 128,), float("nan"), device=DEV)
    y6 = torch.full((n, 2 * h, 2 * w_, co), 7.0, device=DEV); yn = torch.full_like(y6, 7.0)
    d6 = torch.full((n, h, w_, ci), 7.0, device=DEV); dn = torch.full_like(d6, 7.0)
    f6 = lambda: L.unet_convT2x2_fwd_x6(P(x), ci, P(W6), P(b), P(y6), co, n, h, w_, ci, co, P(part), part.numel() * 4, ST())
    fn = lambda: L.unet_convT2x2_fwd_stream(P(x), ci, P(wT), P(b), P(yn), co, n, h, w_, ci, co, ST()) if L.unet_convT2x2_fwd_stream_supported(n, h, w_, ci, co) else L.unet_convT2x2_fwd(P(x), ci, P(wT), P(b), P(yn), co, n, h, w_, ci, co, ST())
    g6 = lambda: L.unet_convT2x2_dgrad_x6(P(dz), co, P(W6d), P(d6), ci, n, h, w_, ci, co, ST())
    gn = lambda: L.unet_convT2x2_dgrad(P(dz), co, P(wT), P(dn), ci, n, h, w_, ci, co, ST())
    nbw = L.unet_convT2x2_wgrad_x6_workspace(n, h, w_, ci, co); ws6 = torch.empty(nbw + 256, dtype=torch.uint8, device=DEV)
    nbn = L.unet_convT2x2_wgrad_workspace(n, h, w_, ci, co); wsn = torch.empty(nbn + 256, dtype=torch.uint8, device=DEV)
    w6 = torch.full((2, 2, co, ci), 7.0, device=DEV); wn = torch.full_like(w6, 7.0)
    h6 = lambda: L.unet_convT2x2_wgrad_x6(P(x), ci, P(dz), co, P(w6), n, h, w_, ci, co, P(ws6), nbw, ST())
    hn = lambda: L.unet_convT2x2_wgrad(P(x), ci, P(dz), co, P(wn), n, h, w_, ci, co, P(wsn), nbn, ST())
    f6(); fn(); g6(); gn(); h6(); hn(); torch.cuda.synchronize()
    refw = torch.einsum("nyaxbk,nyxc->abkc", dz.double().reshape(n, h, 2, w_, 2, co), x.double())
    sums = part.view(co // 64, rows, 64, 2).double().sum(1).reshape(co, 2)
    rs = ref.reshape(-1, co).sum(0); rq = ref.reshape(-1, co).pow(2).sum(0)
    se = float((sums[:, 0] - rs).abs().max() / rs.abs().max()), float((sums[:, 1] - rq).abs().max() / rq.abs().max())
    row = "%-24s fwd x6 max %.2e rms %.2e | native max %.2e rms %.2e || dgrad x6 max %.2e rms %.2e | native max %.2e rms %.2e || wgrad x6 max %.2e rms %.2e | native max %.2e rms %.2e || sums %.1e %.1e" % (
        (str((n, h, w_, ci, co)),) + errs(y6, ref) + errs(yn, ref) + errs(d6, refd) + errs(dn, refd) + errs(w6, refw) + errs(wn, refw) + se)
    if timed:
        fs = lambda: L.unet_convT2x2_fwd_stream_stats(P(x), ci, P(wT), P(b), P(yn), co, n, h, w_, ci, co, P(torch.empty((co // 64) * max(1, L.unet_convT2x2_fwd_stream_stats_rows(n, h, w_, ci, co)) * 128, device=DEV)), (co // 64) * L.unet_convT2x2_fwd_stream_stats_rows(n, h, w_, ci, co) * 512, ST())
        pn = torch.empty((co // 64) * L.unet_convT2x2_fwd_stream_stats_rows(n, h, w_, ci, co) * 128, device=DEV)
        fs = lambda: L.unet_convT2x2_fwd_stream_stats(P(x), ci, P(wT), P(b), P(yn), co, n, h, w_, ci, co, P(pn), pn.numel() * 4, ST())
        row += " || ms fwd+sums x6 %.3f native %.3f | dgrad x6 %.3f native %.3f | wgrad x6 %.3f native %.3f" % (timeit(f6), timeit(fs), timeit(g6), timeit(gn), timeit(h6), timeit(hn))
    print(row, flush=True)
