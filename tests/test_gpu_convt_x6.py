"""BF16x6 transposed-conv forward / data gradient (csrc/convt_x6.hip; reference layer UNet._deconv_layer, UNet/model.py:39-46) through the
C ABI against torch's fp64 transposed convolution: fp32-grade error (at or below the native fp32-MFMA kernels'), exact three-piece weight
operands, BatchNorm sums, channel-sliced outputs (the zero-copy concat writes the upper half of a [N,2H,2W,2C] buffer), ragged widths."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import pkg

pytestmark = pytest.mark.gpu
DEV = "cuda"
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
ST = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def make(shape, seed, scale=1.0):
    n, h, w, ci, co = shape
    g = torch.Generator(device=DEV).manual_seed(seed)
    x = torch.randn(n, h, w, ci, device=DEV, generator=g) * scale
    wT = torch.randn(2, 2, co, ci, device=DEV, generator=g) / float(np.sqrt(ci))
    b = torch.randn(co, device=DEV, generator=g)
    dz = torch.randn(n, 2 * h, 2 * w, co, device=DEV, generator=g)
    return x, wT, b, dz


def ref_fwd(x, wT, b):
    return torch.nn.functional.conv_transpose2d(x.double().permute(0, 3, 1, 2), wT.double().permute(3, 2, 0, 1), b.double(), stride=2).permute(0, 2, 3, 1)


def ref_dgrad(dz, wT):
    return torch.nn.functional.conv2d(dz.double().permute(0, 3, 1, 2), wT.double().permute(3, 2, 0, 1), None, stride=2).permute(0, 2, 3, 1)


def operands(L, wT):
    co, ci = wT.shape[2], wT.shape[3]
    nb = L.unet_convT2x2_x6_weight_bytes(ci, co)
    u = [torch.empty(nb, dtype=torch.uint8, device=DEV) for _ in range(2)]
    for mode in (0, 1):
        L.unet_convT2x2_weight_transform_x6(P(wT), P(u[mode]), ci, co, mode, ST())
    return u


def rel(a, r):
    d = a.double() - r
    return float(d.abs().max() / r.abs().max()), float(d.pow(2).mean().sqrt() / r.pow(2).mean().sqrt())


BASELINE_SHAPES = [(8, 32, 32, 1024, 512), (8, 64, 64, 512, 256), (8, 128, 128, 256, 128), (8, 256, 256, 128, 64)]      # up_4 .. up_1 of BASELINE config 2


@pytest.mark.parametrize("shape", [(2, 8, 8, 128, 64), (1, 16, 16, 256, 128), (4, 8, 8, 128, 192), (2, 8, 24, 128, 64), (1, 12, 32, 384, 64)] + BASELINE_SHAPES)
def test_forward_and_data_gradient_are_fp32_grade(shape):
    L = pkg("_lib").lib()
    n, h, w, ci, co = shape
    assert L.unet_convT2x2_x6_supported(*shape) == 1
    x, wT, b, dz = make(shape, 7 + h + ci)
    W6, W6d = operands(L, wT)
    rows = L.unet_convT2x2_x6_stats_rows(*shape)
    assert rows == 4 * (n * h * w // 128)
    part = torch.full(((co // 64) * rows * 128,), float("nan"), device=DEV)
    y = torch.full((n, 2 * h, 2 * w, co), float("nan"), device=DEV); y2 = torch.full_like(y, float("nan"))
    dx = torch.full((n, h, w, ci), float("nan"), device=DEV); yn = torch.empty_like(y); dn = torch.empty_like(dx)
    L.unet_convT2x2_fwd_x6(P(x), ci, P(W6), P(b), P(y), co, n, h, w, ci, co, P(part), part.numel() * 4, ST())
    L.unet_convT2x2_fwd_x6(P(x), ci, P(W6), P(b), P(y2), co, n, h, w, ci, co, None, 0, ST())
    L.unet_convT2x2_dgrad_x6(P(dz), co, P(W6d), P(dx), ci, n, h, w, ci, co, ST())
    L.unet_convT2x2_fwd(P(x), ci, P(wT), P(b), P(yn), co, n, h, w, ci, co, ST())
    L.unet_convT2x2_dgrad(P(dz), co, P(wT), P(dn), ci, n, h, w, ci, co, ST())
    # weight gradient: dw[a,b,co,ci] = sum_{n,i,j} dz[n,2i+a,2j+b,co] x[n,i,j,ci]
    nbw = L.unet_convT2x2_wgrad_x6_workspace(*shape); ws = torch.empty(nbw + 256, dtype=torch.uint8, device=DEV)
    nbn = L.unet_convT2x2_wgrad_workspace(*shape); wsn = torch.empty(nbn + 256, dtype=torch.uint8, device=DEV)
    dw = torch.full((2, 2, co, ci), float("nan"), device=DEV); dwn = torch.empty_like(dw)
    L.unet_convT2x2_wgrad_x6(P(x), ci, P(dz), co, P(dw), n, h, w, ci, co, P(ws), nbw, ST())
    L.unet_convT2x2_wgrad(P(x), ci, P(dz), co, P(dwn), n, h, w, ci, co, P(wsn), nbn, ST())
    rw = torch.einsum("nyaxbk,nyxc->abkc", dz.double().reshape(n, h, 2, w, 2, co), x.double())
    (wm_, wr_), (vm_, vr_) = rel(dw, rw), rel(dwn, rw)
    if shape in BASELINE_SHAPES:
        # the layers the route exists for (sums over 8e3 .. 5e5 pixels): fp32-grade by the same criterion as every other BF16x6 kernel --
        # rms error at most 1.25x, max error at most 1.5x the native fp32-MFMA kernel's
        assert wr_ <= 1.25 * vr_ and wm_ <= 1.5 * vm_ and wr_ < 2e-6, (wm_, wr_, vm_, vr_)
    else:
        # sums over 128 .. 384 pixels: the native kernel's error is a handful of fp32 roundings of the accumulator (measured 0.9e-7 .. 1.0e-7
        # rms, 1.3e-7 .. 1.9e-7 max: there is nothing below it), while the six-product form adds its dropped piece products -- at most 2^-21,
        # on average 2^-24.5 of each product and of the product's sign (tests/test_bf16x6_arithmetic.py), which no longer hides under an
        # accumulation error that has not had time to grow.  A ratio to a kernel sitting on the rounding floor says nothing, so the bound
        # here is ABSOLUTE: 2.5e-7 rms / 5e-7 max of the result's scale = two / four fp32 ulps (measured 1.7e-7 / 3.8e-7)
        assert wr_ < 2.5e-7 and wm_ < 5e-7, (wm_, wr_, vm_, vr_)
    if nbw > 16:
        with pytest.raises(pkg("_lib").UnetHipError, match="workspace too small"):
            L.unet_convT2x2_wgrad_x6(P(x), ci, P(dz), co, P(dw), n, h, w, ci, co, P(ws), nbw - 16, ST())
    rf, rd = ref_fwd(x, wT, b), ref_dgrad(dz, wT)
    assert torch.equal(y, y2)
    (fm, fr), (nm, nr) = rel(y, rf), rel(yn, rf)
    (gm, gr), (hm, hr) = rel(dx, rd), rel(dn, rd)
    assert fr <= 1.25 * nr and fm <= 1.5 * nm and fr < 1e-6, (fm, fr, nm, nr)
    assert gr <= 1.25 * hr and gm <= 1.5 * hm and gr < 2e-6, (gm, gr, hm, hr)
    # BatchNorm sums of the output: per channel sum and sum of squares over all output pixels
    sums = part.view(co // 64, rows, 64, 2).double().sum(1).reshape(co, 2)
    flat = rf.reshape(-1, co)
    assert torch.allclose(sums[:, 0], flat.sum(0), rtol=0, atol=2e-6 * float(flat.abs().sum(0).max()))
    assert torch.allclose(sums[:, 1], flat.pow(2).sum(0), rtol=2e-6, atol=0)
    # refusals: a statistics buffer that is too small, an unsupported pixel count
    E = pkg("_lib").UnetHipError
    with pytest.raises(E, match="workspace too small"):
        L.unet_convT2x2_fwd_x6(P(x), ci, P(W6), P(b), P(y), co, n, h, w, ci, co, P(part), 64, ST())
    assert L.unet_convT2x2_x6_supported(1, 5, 5, ci, co) == 0 and L.unet_convT2x2_x6_supported(n, h, w, 64, co) == 0


@pytest.mark.parametrize("shape", [(2, 8, 8, 128, 64), (1, 16, 16, 256, 128), (1, 12, 32, 384, 64), (8, 64, 64, 512, 256), (8, 256, 256, 128, 64)])
def test_data_gradient_leaves_the_producers_batchnorm_backward_sums(shape):
    # unet_convT2x2_dgrad_x6_sums (round 6): dx is the dy of the layer that produced x (dec_Nb -> up_(N-1), UNet/model.py:117-132), so the
    # kernel's epilogue also leaves that layer's BatchNorm-backward sums (sum dx, sum dx * r) per 128-pixel tile, from its fp32 accumulators:
    # dx bit-identical to the plain kernel's, sums = those of the stored dx against fp64, a saved activation with a leading dimension > Cin
    # (the concat buffer), the too-small-buffer and one-null refusals
    L = pkg("_lib").lib()
    n, h, w, ci, co = shape
    x, wT, b, dz = make(shape, 11 + h + ci)
    _, W6d = operands(L, wT)
    ldr = ci + 64
    r = torch.randn(n, h, w, ldr, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3 + ci))
    rows = L.unet_convT2x2_x6_bnbwd_rows(*shape)
    assert rows == n * h * w // 128
    part = torch.full(((ci // 64) * rows * 128,), float("nan"), device=DEV)
    dx = torch.full((n, h, w, ci), float("nan"), device=DEV); dref = torch.full_like(dx, float("nan"))
    L.unet_convT2x2_dgrad_x6_sums(P(dz), co, P(W6d), P(dx), ci, n, h, w, ci, co, P(r), ldr, P(part), part.numel() * 4, ST())
    L.unet_convT2x2_dgrad_x6(P(dz), co, P(W6d), P(dref), ci, n, h, w, ci, co, ST())
    assert torch.equal(dx, dref)
    sums = part.view(ci // 64, rows, 64, 2).double().sum(1).reshape(ci, 2)
    d64, r64 = dx.double().reshape(-1, ci), r[..., :ci].double().reshape(-1, ci)
    assert torch.allclose(sums[:, 0], d64.sum(0), rtol=0, atol=2e-6 * float(d64.abs().sum(0).max()))
    assert torch.allclose(sums[:, 1], (d64 * r64).sum(0), rtol=0, atol=2e-6 * float((d64 * r64).abs().sum(0).max()))
    E = pkg("_lib").UnetHipError
    with pytest.raises(E, match="workspace too small"):
        L.unet_convT2x2_dgrad_x6_sums(P(dz), co, P(W6d), P(dx), ci, n, h, w, ci, co, P(r), ldr, P(part), 64, ST())
    with pytest.raises(E, match="bad argument"):
        L.unet_convT2x2_dgrad_x6_sums(P(dz), co, P(W6d), P(dx), ci, n, h, w, ci, co, P(r), ldr, None, 0, ST())


def test_weight_operands_are_an_exact_three_piece_split():
    L = pkg("_lib").lib()
    ci, co = 128, 64
    g = torch.Generator(device=DEV).manual_seed(3)
    wT = torch.randn(2, 2, co, ci, device=DEV, generator=g) * torch.pow(10.0, torch.randint(-6, 6, (2, 2, co, ci), device=DEV, generator=g).float())
    W6, W6d = operands(L, wT)
    flat = wT.reshape(4 * co, ci)                                     # [n = tap * Cout + co][k = ci]
    for mode, u in ((0, W6), (1, W6d)):
        K, N = (ci, 4 * co) if mode == 0 else (4 * co, ci)
        t = u.view(torch.int16).view(K // 16, 3, N, 2, 8)             # [chunk][piece][column][slot][8 k]
        nn = torch.arange(N, device=DEV)
        swz = ((nn >> 3) & 1).view(1, 1, N, 1, 1).expand(K // 16, 3, N, 1, 8)
        lo = torch.gather(t, 3, swz).squeeze(3); hi = torch.gather(t, 3, 1 - swz).squeeze(3)     # k halves 0 / 1 after undoing the slot swizzle
        pieces = torch.cat([lo, hi], dim=-1)                          # [chunk][piece][column][16 k]
        val = (pieces.to(torch.int32) << 16).view(torch.float32).double().sum(1)                  # h + m + l
        B = val.permute(0, 2, 1).reshape(K, N)                        # [k][n]
        want = flat.t() if mode == 0 else flat
        assert torch.equal(B.float(), want) and torch.equal(B, want.double())


def test_channel_slices_and_strides():
    """the engine's use: the forward writes channels [C, 2C) of the concat buffer (ldo = 2C), the data gradient reads dz with its own stride"""
    L = pkg("_lib").lib()
    n, h, w, ci, co = 2, 16, 16, 128, 64
    x, wT, b, dz = make((n, h, w, ci, co), 5)
    W6, W6d = operands(L, wT)
    xs = torch.randn(n, h, w, ci + 8, device=DEV); xs[..., :ci] = x
    cat = torch.full((n, 2 * h, 2 * w, 2 * co), -3.0, device=DEV)
    upper = cat[..., co:]
    L.unet_convT2x2_fwd_x6(P(xs), ci + 8, P(W6), P(b), P(upper), 2 * co, n, h, w, ci, co, None, 0, ST())
    ref = torch.empty(n, 2 * h, 2 * w, co, device=DEV)
    L.unet_convT2x2_fwd_x6(P(x), ci, P(W6), P(b), P(ref), co, n, h, w, ci, co, None, 0, ST())
    assert torch.equal(cat[..., co:], ref) and bool((cat[..., :co] == -3.0).all())
    # weight gradient with strided operands (x inside a wider buffer, dz with padding channels)
    nbw = L.unet_convT2x2_wgrad_x6_workspace(n, h, w, ci, co); ws = torch.empty(nbw + 256, dtype=torch.uint8, device=DEV)
    dw1 = torch.empty(2, 2, co, ci, device=DEV); dw2 = torch.empty_like(dw1)
    dzp = torch.randn(n, 2 * h, 2 * w, co + 12, device=DEV); dzp[..., :co] = dz
    L.unet_convT2x2_wgrad_x6(P(x), ci, P(dz), co, P(dw1), n, h, w, ci, co, P(ws), nbw, ST())
    L.unet_convT2x2_wgrad_x6(P(xs), ci + 8, P(dzp), co + 12, P(dw2), n, h, w, ci, co, P(ws), nbw, ST())
    assert torch.equal(dw1, dw2)
    dzs = torch.randn(n, 2 * h, 2 * w, co + 12, device=DEV); dzs[..., :co] = dz
    dxs = torch.full((n, h, w, ci + 4), -5.0, device=DEV); dref = torch.empty(n, h, w, ci, device=DEV)
    L.unet_convT2x2_dgrad_x6(P(dzs), co + 12, P(W6d), P(dxs), ci + 4, n, h, w, ci, co, ST())
    L.unet_convT2x2_dgrad_x6(P(dz), co, P(W6d), P(dref), ci, n, h, w, ci, co, ST())
    assert torch.equal(dxs[..., :ci], dref) and bool((dxs[..., ci:] == -5.0).all())


def test_plan_routes_the_fp32_transposed_convs_to_bf16x6():
    plan, L = pkg("plan"), pkg("_lib").lib()
    pl = plan.build_plan(plan.EngineOptions(), 1, 2, 8, 512, 512, True, True, L)
    for name in ("up_4", "up_3", "up_2", "up_1"):
        assert pl.layer[name].fwd == "convt_x6" and pl.layer[name].dgrad == "convt_x6" and pl.layer[name].wgrad == "convt_x6" and pl.layer[name].fwd_stats, name
    nat = plan.build_plan(plan.EngineOptions(fp32_matrix="native"), 1, 2, 8, 512, 512, True, True, L)
    assert all(nat.layer[n].fwd == "convt_stream" and nat.layer[n].dgrad == "convt_igemm" and nat.layer[n].wgrad == "convt" for n in ("up_4", "up_3", "up_2", "up_1"))
    odd = plan.build_plan(plan.EngineOptions(), 1, 2, 1, 48, 80, True, True, L)          # 3 x 5 pixels at level 5: not a multiple of the 128-pixel tile
    assert odd.layer["up_4"].fwd != "convt_x6"
