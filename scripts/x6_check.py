"""Development check of the BF16x6 Winograd kernels (csrc/winograd_x6.hip) on one MI355X: error against torch's fp64 convolution next to
the native fp32-MFMA kernels' error on the same inputs, and launch times at the BASELINE config-2 layer shapes.
usage: python scripts/x6_check.py [quick|full]"""
import ctypes
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
L = importlib.import_module("semantic-segmentation-unet_amd._lib").lib()
DEV = "cuda"
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
ST = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def u6(w, mode):
    ci, co = w.shape[2], w.shape[3]
    u = torch.empty(L.unet_winograd_x6_weight_bytes(ci, co), dtype=torch.uint8, device=DEV)
    L.unet_winograd_weight_transform_x6(P(w), P(u), ci, co, mode, ST())
    return u


def uc(w, mode):
    ci, co = w.shape[2], w.shape[3]
    u = torch.empty(16 * ci * co, device=DEV)
    L.unet_winograd_weight_transform(P(w), P(u), ci, co, mode, ST())
    return u


def ref_fwd(x, w, b, relu):
    y = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), None if b is None else b.double(), padding=1)
    if relu:
        y = torch.relu(y)
    return y.permute(0, 2, 3, 1)


def ref_dgrad(dz, w):
    wt = torch.flip(w.double(), (0, 1)).permute(2, 3, 0, 1)             # [ci][co][a][b] of the rotated filter
    return torch.nn.functional.conv2d(dz.double().permute(0, 3, 1, 2), wt, None, padding=1).permute(0, 2, 3, 1)


def errs(a, r):
    d = (a.double() - r)
    return float(d.abs().max() / r.abs().max()), float(d.pow(2).mean().sqrt() / r.pow(2).mean().sqrt())


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def check(shape, seed=0, scale=1.0, time_it=False):
    n, h, w_, ci, co = shape
    g = torch.Generator(device=DEV).manual_seed(seed + ci + co + h)
    x = torch.randn(n, h, w_, ci, device=DEV, generator=g) * scale
    dz = torch.randn(n, h, w_, co, device=DEV, generator=g)
    wt = torch.randn(3, 3, ci, co, device=DEV, generator=g) / float(np.sqrt(9 * ci))
    b = torch.randn(co, device=DEV, generator=g)
    rf = ref_fwd(x, wt, b, True)
    rd = ref_dgrad(dz, wt)
    y6 = torch.full((n, h, w_, co), 7.0, device=DEV); yn = torch.full_like(y6, 7.0)
    d6 = torch.full((n, h, w_, ci), 7.0, device=DEV); dn = torch.full_like(d6, 7.0)
    U6, U6d, Uc, Ucd = u6(wt, 0), u6(wt, 1), uc(wt, 2), uc(wt, 3)
    f6 = lambda: L.unet_conv3x3_fwd_winograd_x6(P(x), ci, None, P(U6), P(b), P(y6), co, n, h, w_, ci, co, 1, None, 0, ST())
    fn = lambda: L.unet_conv3x3_fwd_winograd_fused(P(x), ci, None, P(Uc), P(b), P(yn), co, n, h, w_, ci, co, 1, None, 0, ST())
    g6 = lambda: L.unet_conv3x3_dgrad_winograd_x6(P(dz), co, P(U6d), P(d6), ci, n, h, w_, ci, co, None, 0, 0, 0, None, 0, ST())
    gn = lambda: L.unet_conv3x3_dgrad_winograd_fused(P(dz), co, P(Ucd), P(dn), ci, n, h, w_, ci, co, None, 0, 0, 0, None, 0, ST())
    f6(); fn(); g6(); gn(); torch.cuda.synchronize()
    row = "%-24s fwd x6 max %.2e rms %.2e | native max %.2e rms %.2e || dgrad x6 max %.2e rms %.2e | native max %.2e rms %.2e" % (
        (str(shape),) + errs(y6, rf) + errs(yn, rf) + errs(d6, rd) + errs(dn, rd))
    if time_it:
        row += " || ms fwd x6 %.3f native %.3f dgrad x6 %.3f native %.3f" % (timeit(f6), timeit(fn), timeit(g6), timeit(gn))
    print(row, flush=True)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "quick"
    print("device:", torch.cuda.get_device_name(0), flush=True)
    for shp in [(2, 16, 16, 64, 64), (1, 16, 32, 128, 128), (2, 32, 48, 64, 128), (1, 20, 36, 256, 128), (5, 104, 136, 64, 64)]:
        check(shp)
    check((2, 16, 16, 64, 64), scale=1e4)
    if mode == "full":
        for shp in [(8, 512, 512, 64, 64), (8, 512, 512, 128, 64), (8, 256, 256, 128, 128), (8, 256, 256, 256, 128), (8, 128, 128, 256, 256),
                    (8, 128, 128, 512, 256), (8, 64, 64, 512, 512), (8, 64, 64, 1024, 512), (8, 32, 32, 512, 1024), (8, 32, 32, 1024, 1024)]:
            check(shp, time_it=True)
