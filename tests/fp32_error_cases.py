"""Shared by tests/test_gpu_fp32_errors.py and scripts/make_fp32_error_table.py: every fp32 contraction kernel of the C ABI run on seeded
inputs and measured against torch's fp64 convolution of the same operands -- max |err| / max |ref| and rms(err) / rms(ref).

The committed table (tests/golden/fp32_kernel_errors.json) holds the values measured on an MI355X when the table was made; the test asserts
every kernel stays within 4 x of its own entry.  A bound derived from the kernel's own rounding noise (1e-7 .. 1e-6 here) instead of a
generic 2e-5 is what lets the suite tell fp32 arithmetic from "almost fp32" (a three-product bf16 emulation sits ~6 x above these rows)."""
import ctypes

import numpy as np
import torch

DEV = "cuda"
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
ST = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ws(n):
    return torch.empty(int(n) + 256, dtype=torch.uint8, device=DEV)


def ref_conv(x, w, b=None, relu=False):
    y = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), None if b is None else b.double(), padding=1)
    return (torch.relu(y) if relu else y).permute(0, 2, 3, 1)


def ref_dgrad(dz, w):
    wt = torch.flip(w.double(), (0, 1)).permute(2, 3, 0, 1)
    return torch.nn.functional.conv2d(dz.double().permute(0, 3, 1, 2), wt, None, padding=1).permute(0, 2, 3, 1)


def ref_wgrad(x, dz):
    n, h, w, ci = x.shape
    xp = torch.nn.functional.pad(x.double().permute(0, 3, 1, 2), (1, 1, 1, 1)); dzn = dz.double().permute(0, 3, 1, 2)
    return torch.stack([torch.stack([torch.einsum("ncyx,nkyx->ck", xp[:, :, a:a + h, b:b + w], dzn) for b in range(3)]) for a in range(3)])


def ref_convt(x, w, b):        # w [2][2][co][ci]
    y = torch.nn.functional.conv_transpose2d(x.double().permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), b.double(), stride=2)
    return y.permute(0, 2, 3, 1)


def errs(a, r):
    d = a.double() - r
    return float(d.abs().max() / r.abs().max()), float(d.pow(2).mean().sqrt() / r.pow(2).mean().sqrt())


def inputs(shape, seed, kind="normal"):
    """x [n,h,w,ci], dz [n,h,w,co], w [3,3,ci,co], b [co].  kind: 'normal' | 'raw16' (un-normalised 16-bit pixel values, all positive) |
    'mixed' (eight decades of magnitude inside every 16-channel chunk) | 'edges' (mantissas on and next to the bf16 piece boundaries)"""
    n, h, w_, ci, co = shape
    g = torch.Generator(device=DEV).manual_seed(seed)
    x = torch.randn(n, h, w_, ci, device=DEV, generator=g)
    dz = torch.randn(n, h, w_, co, device=DEV, generator=g)
    wt = torch.randn(3, 3, ci, co, device=DEV, generator=g) / float(np.sqrt(9 * ci))
    b = torch.randn(co, device=DEV, generator=g)
    if kind == "raw16":
        x = torch.rand(n, h, w_, ci, device=DEV, generator=g) * 65535.0
        dz = dz * 1e4
    elif kind == "mixed":
        s = torch.pow(10.0, (torch.arange(ci, device=DEV) % 8).float() - 4.0)
        x = x * s
        dz = dz * torch.pow(10.0, (torch.arange(co, device=DEV) % 8).float() - 4.0)
    elif kind == "edges":
        def edgy(t):
            pats = torch.tensor([0x007fff, 0x008000, 0x008001, 0x7f8000, 0x7f7fff, 0x7fffff, 0x00ffff, 0x010000, 0x000001, 0x7f0000, 0x00ff80, 0x3f807f],
                                dtype=torch.int32, device=DEV)
            bits = t.contiguous().view(torch.int32)
            idx = torch.randint(0, pats.numel(), t.shape, device=DEV, generator=g)
            return ((bits & ~0x7fffff) | pats[idx]).view(torch.float32)
        x, dz, wt = edgy(x), edgy(dz), edgy(wt)
    return x, dz, wt, b


def x6_weights(L, w, mode):
    ci, co = w.shape[2], w.shape[3]
    u = torch.empty(L.unet_winograd_x6_weight_bytes(ci, co), dtype=torch.uint8, device=DEV)
    L.unet_winograd_weight_transform_x6(P(w), P(u), ci, co, mode, ST())
    return u


def native_weights(L, w, mode):
    ci, co = w.shape[2], w.shape[3]
    u = torch.empty(16 * ci * co, device=DEV)
    L.unet_winograd_weight_transform(P(w), P(u), ci, co, mode, ST())
    return u


def run_case(L, family, shape, seed, kind="normal"):
    """-> (max_rel, rms_rel) of one kernel family on one shape"""
    n, h, w_, ci, co = shape
    x, dz, wt, b = inputs(shape, seed, kind)
    if family in ("wino_fwd", "x6_fwd", "mfma_fwd"):
        out = torch.empty(n, h, w_, co, device=DEV)
        if family == "wino_fwd":
            L.unet_conv3x3_fwd_winograd_fused(P(x), ci, None, P(native_weights(L, wt, 2)), P(b), P(out), co, n, h, w_, ci, co, 1, None, 0, ST())
        elif family == "x6_fwd":
            L.unet_conv3x3_fwd_winograd_x6(P(x), ci, None, P(x6_weights(L, wt, 0)), P(b), P(out), co, n, h, w_, ci, co, 1, None, 0, ST())
        else:
            L.unet_conv3x3_fwd_mfma(P(x), ci, P(wt), P(b), P(out), co, n, h, w_, ci, co, 1, ST())
        return errs(out, ref_conv(x, wt, b, True))
    if family in ("wino_dgrad", "x6_dgrad", "mfma_dgrad"):
        dx = torch.empty(n, h, w_, ci, device=DEV)
        if family == "wino_dgrad":
            L.unet_conv3x3_dgrad_winograd_fused(P(dz), co, P(native_weights(L, wt, 3)), P(dx), ci, n, h, w_, ci, co, None, 0, 0, 0, None, 0, ST())
        elif family == "x6_dgrad":
            L.unet_conv3x3_dgrad_winograd_x6(P(dz), co, P(x6_weights(L, wt, 1)), P(dx), ci, n, h, w_, ci, co, None, 0, 0, 0, None, 0, ST())
        else:
            L.unet_conv3x3_dgrad_mfma(P(dz), co, P(wt), P(dx), ci, n, h, w_, ci, co, ST())
        return errs(dx, ref_dgrad(dz, wt))
    if family in ("wino_wgrad", "mfma_wgrad"):
        dw = torch.empty(3, 3, ci, co, device=DEV)
        if family == "wino_wgrad":
            nb = L.unet_conv3x3_wgrad_winograd_fused_workspace(n, h, w_, ci, co, 0); ws = _ws(nb)
            L.unet_conv3x3_wgrad_winograd_fused(P(x), ci, P(dz), co, P(dw), n, h, w_, ci, co, 0, P(ws), nb, ST())
        else:
            nb = L.unet_conv3x3_wgrad_mfma_workspace(n, h, w_, ci, co); ws = _ws(nb)
            L.unet_conv3x3_wgrad_mfma(P(x), ci, P(dz), co, P(dw), n, h, w_, ci, co, P(ws), nb, ST())
        return errs(dw, ref_wgrad(x, dz))
    if family in ("convt_fwd", "convt_fwd_stream", "convt_dgrad", "convt_wgrad", "convt_x6_fwd", "convt_x6_dgrad", "convt_x6_wgrad"):
        g = torch.Generator(device=DEV).manual_seed(seed + 1)
        wT = torch.randn(2, 2, co, ci, device=DEV, generator=g) / float(np.sqrt(ci))
        if family in ("convt_fwd", "convt_fwd_stream"):
            out = torch.empty(n, 2 * h, 2 * w_, co, device=DEV)
            fn = L.unet_convT2x2_fwd if family == "convt_fwd" else L.unet_convT2x2_fwd_stream
            fn(P(x), ci, P(wT), P(b), P(out), co, n, h, w_, ci, co, ST())
            return errs(out, ref_convt(x, wT, b))
        dzT = torch.randn(n, 2 * h, 2 * w_, co, device=DEV, generator=g)
        if family in ("convt_x6_fwd", "convt_x6_dgrad"):
            mode = 0 if family == "convt_x6_fwd" else 1
            u = torch.empty(L.unet_convT2x2_x6_weight_bytes(ci, co), dtype=torch.uint8, device=DEV)
            L.unet_convT2x2_weight_transform_x6(P(wT), P(u), ci, co, mode, ST())
            if mode == 0:
                out = torch.empty(n, 2 * h, 2 * w_, co, device=DEV)
                L.unet_convT2x2_fwd_x6(P(x), ci, P(u), P(b), P(out), co, n, h, w_, ci, co, None, 0, ST())
                return errs(out, ref_convt(x, wT, b))
            dx = torch.empty(n, h, w_, ci, device=DEV)
            L.unet_convT2x2_dgrad_x6(P(dzT), co, P(u), P(dx), ci, n, h, w_, ci, co, ST())
            ref = torch.nn.functional.conv2d(dzT.double().permute(0, 3, 1, 2), wT.double().permute(3, 2, 0, 1), None, stride=2).permute(0, 2, 3, 1)
            return errs(dx, ref)
        if family == "convt_dgrad":
            dx = torch.empty(n, h, w_, ci, device=DEV)
            L.unet_convT2x2_dgrad(P(dzT), co, P(wT), P(dx), ci, n, h, w_, ci, co, ST())
            ref = torch.nn.functional.conv2d(dzT.double().permute(0, 3, 1, 2), wT.double().permute(3, 2, 0, 1), None, stride=2).permute(0, 2, 3, 1)
            return errs(dx, ref)
        dw = torch.empty(2, 2, co, ci, device=DEV)
        if family == "convt_x6_wgrad":
            nb = L.unet_convT2x2_wgrad_x6_workspace(n, h, w_, ci, co); ws = _ws(nb)
            L.unet_convT2x2_wgrad_x6(P(x), ci, P(dzT), co, P(dw), n, h, w_, ci, co, P(ws), nb, ST())
        else:
            nb = L.unet_convT2x2_wgrad_workspace(n, h, w_, ci, co); ws = _ws(nb)
            L.unet_convT2x2_wgrad(P(x), ci, P(dzT), co, P(dw), n, h, w_, ci, co, P(ws), nb, ST())
        d6 = dzT.double().reshape(n, h, 2, w_, 2, co)
        ref = torch.einsum("nyaxbk,nyxc->abkc", d6, x.double())
        return errs(dw, ref)
    raise KeyError(family)


# (family, shape N H W Cin Cout, seed): small enough for the whole table in a few seconds, with ragged tile grids and multi-tile workgroups
CASES = [
    ("wino_fwd", (2, 16, 16, 64, 64), 1), ("wino_fwd", (1, 20, 36, 256, 128), 2), ("wino_fwd", (5, 104, 136, 64, 64), 3),
    ("wino_dgrad", (2, 16, 16, 64, 64), 1), ("wino_dgrad", (1, 20, 36, 256, 128), 2), ("wino_dgrad", (5, 104, 136, 64, 64), 3),
    ("wino_wgrad", (2, 16, 16, 64, 64), 1), ("wino_wgrad", (3, 8, 24, 64, 192), 2), ("wino_wgrad", (2, 32, 48, 128, 128), 3),
    ("x6_fwd", (2, 16, 16, 64, 64), 1), ("x6_fwd", (1, 20, 36, 256, 128), 2), ("x6_fwd", (5, 104, 136, 64, 64), 3),
    ("x6_dgrad", (2, 16, 16, 64, 64), 1), ("x6_dgrad", (1, 20, 36, 256, 128), 2), ("x6_dgrad", (5, 104, 136, 64, 64), 3),
    # (round 5: more reduce-channel counts in both directions)
    ("x6_fwd", (2, 24, 40, 512, 64), 4), ("x6_dgrad", (2, 16, 24, 64, 256), 4), ("x6_dgrad", (1, 32, 16, 128, 512), 5),
    ("wino_fwd", (2, 24, 40, 512, 64), 4), ("wino_dgrad", (2, 16, 24, 64, 256), 4), ("wino_dgrad", (1, 32, 16, 128, 512), 5),
    ("mfma_fwd", (2, 8, 32, 64, 64), 1), ("mfma_fwd", (1, 5, 33, 64, 192), 2),
    ("mfma_dgrad", (2, 8, 32, 64, 64), 1), ("mfma_wgrad", (2, 8, 32, 64, 64), 1),
    ("convt_fwd", (2, 4, 32, 128, 64), 1), ("convt_fwd_stream", (2, 8, 16, 128, 128), 1), ("convt_fwd_stream", (4, 8, 16, 64, 192), 2),
    ("convt_dgrad", (2, 4, 32, 128, 64), 1), ("convt_wgrad", (2, 4, 32, 128, 64), 1),
    ("convt_x6_fwd", (2, 4, 32, 128, 64), 1), ("convt_x6_fwd", (2, 8, 16, 256, 128), 2), ("convt_x6_dgrad", (2, 4, 32, 128, 64), 1), ("convt_x6_dgrad", (2, 8, 16, 256, 128), 2),
    ("convt_x6_wgrad", (2, 4, 32, 128, 64), 1), ("convt_x6_wgrad", (2, 8, 16, 256, 128), 2),
]


def case_key(family, shape, seed):
    return "%s %s seed %d" % (family, "x".join(map(str, shape)), seed)
