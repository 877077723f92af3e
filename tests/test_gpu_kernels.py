"""GPU parity tests, per kernel family: each C-ABI entry point of include/unet_hip.h against the numpy oracle
(oracle/unet_numpy.py, fp64) on the same seeded inputs.  Tolerances are fp32 reassociation bounds, stated per test as
max|hip - oracle| <= tol * max|oracle|."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import unet_numpy as on

pytestmark = pytest.mark.gpu

DEV = "cuda"


def P(t):
    return ctypes.c_void_p(t.data_ptr())


def ST():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def to_nhwc(a):       # oracle NCHW -> device NHWC fp32
    return torch.as_tensor(np.ascontiguousarray(a.transpose(0, 2, 3, 1)).astype(np.float32)).to(DEV)


def from_nhwc(t):     # device NHWC -> NCHW float64
    return t.detach().cpu().numpy().transpose(0, 3, 1, 2).astype(np.float64)


def dev(a, dtype=np.float32):
    return torch.as_tensor(np.ascontiguousarray(a).astype(dtype)).to(DEV)


def relerr(a, b):
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


def ws_bytes(n):
    return torch.empty(int(n) + 256, dtype=torch.uint8, device=DEV)


CONV_SHAPES = [  # N, H, W, Cin, Cout
    (2, 8, 32, 64, 64),        # exact tiles, 64-wide N tile (8-row config)
    (1, 12, 40, 32, 128),      # ragged rows and columns, 128-wide N tile
    (2, 16, 16, 128, 256),     # W < tile width (deep layers of small inputs)
    (1, 5, 33, 64, 192),       # odd sizes, N = 192 -> 64-wide config
]


@pytest.mark.parametrize("shape", CONV_SHAPES)
def test_conv3x3_fwd_dgrad_wgrad_mfma(hip, shape):
    n, h, w, ci, co = shape
    rng = np.random.default_rng(hash(shape) & 0xffff)
    x = rng.standard_normal((n, ci, h, w))
    wt = rng.standard_normal((3, 3, ci, co)) / np.sqrt(9 * ci)
    b = rng.standard_normal(co)
    dz = rng.standard_normal((n, co, h, w))
    z_ref = np.maximum(on.conv_same_fwd(x, wt, b), 0)
    dx_ref, dw_ref, _ = on.conv_same_bwd(x, wt, dz)
    # input read from a channel slice of a wider buffer (ld > C), as the concat consumers do
    xbuf = torch.zeros(n, h, w, ci + 8, device=DEV); xbuf[..., 4:4 + ci] = to_nhwc(x)
    xv = xbuf[..., 4:4 + ci]
    wd, bd, dzd = dev(wt), dev(b), to_nhwc(dz)
    out = torch.full((n, h, w, co), 7.0, device=DEV)
    hip.unet_conv3x3_fwd_mfma(P(xv), ci + 8, P(wd), P(bd), P(out), co, n, h, w, ci, co, 1, ST())
    assert relerr(from_nhwc(out), z_ref) < 2e-5
    if ci % 64 == 0:          # dgrad's GEMM N is Cin: needs a multiple of 64 (true for every U-Net layer that needs dgrad)
        dx = torch.full((n, h, w, ci), 7.0, device=DEV)
        hip.unet_conv3x3_dgrad_mfma(P(dzd), co, P(wd), P(dx), ci, n, h, w, ci, co, ST())
        assert relerr(from_nhwc(dx), dx_ref) < 2e-5
    if ci % 64 == 0:
        nb = hip.unet_conv3x3_wgrad_mfma_workspace(n, h, w, ci, co)
        ws = ws_bytes(nb)
        dw = torch.full((3, 3, ci, co), 7.0, device=DEV)
        hip.unet_conv3x3_wgrad_mfma(P(xv), ci + 8, P(dzd), co, P(dw), n, h, w, ci, co, P(ws), nb, ST())
        assert relerr(dw.cpu().numpy().astype(np.float64), dw_ref) < 2e-5


def test_conv3x3_mfma_linearity_at_full_size(hip):
    # size-independent property at a BASELINE config-2 layer shape (B=8, 512x512, 64->64): conv(x1 + 2*x2) ==
    # conv(x1) + 2*conv(x2) (bias 0, no ReLU), and a delta kernel reproduces the shifted input exactly.
    n, h, w, c = 8, 512, 512, 64
    g = torch.Generator(device=DEV); g.manual_seed(0)
    x1 = torch.randn(n, h, w, c, device=DEV, generator=g); x2 = torch.randn(n, h, w, c, device=DEV, generator=g)
    wt = torch.randn(3, 3, c, c, device=DEV, generator=g) / 24.0
    o = [torch.empty(n, h, w, c, device=DEV) for _ in range(3)]
    for xin, out in ((x1, o[0]), (x2, o[1]), (x1 + 2 * x2, o[2])):
        hip.unet_conv3x3_fwd_mfma(P(xin), c, P(wt), None, P(out), c, n, h, w, c, c, 0, ST())
    err = (o[2] - (o[0] + 2 * o[1])).abs().max().item() / o[2].abs().max().item()
    assert err < 1e-5
    wd = torch.zeros(3, 3, c, c, device=DEV); wd[0, 2] = torch.eye(c, device=DEV)     # tap (a=0,b=2): reads (y-1, x+1)
    hip.unet_conv3x3_fwd_mfma(P(x1), c, P(wd), None, P(o[0]), c, n, h, w, c, c, 0, ST())
    assert torch.equal(o[0][:, 1:, :-1], x1[:, :-1, 1:])
    assert o[0][:, 0].abs().max().item() == 0 and o[0][:, :, -1].abs().max().item() == 0


@pytest.mark.parametrize("shape", [(8, 512, 512, 64, 64), (8, 512, 512, 128, 64), (8, 32, 32, 1024, 1024)])
def test_fused_winograd_trio_properties_at_full_size(hip, shape):
    # BASELINE config-2 layer shapes (the oracle is too slow here): (1) the fused Winograd forward agrees with the direct
    # MFMA kernel - two independent implementations; (2) the three kernels are mutually adjoint:
    #     <conv(x, w), dz> == <x, dgrad(dz, w)> == <w, wgrad(x, dz)>        (no bias, no ReLU; inner products in fp64)
    # 8192 / 8192 / 512 tile blocks on 256 persistent workgroups, image borders included.
    n, h, w, ci, co = shape
    g = torch.Generator(device=DEV); g.manual_seed(ci + co)
    x = torch.randn(n, h, w, ci, device=DEV, generator=g); dz = torch.randn(n, h, w, co, device=DEV, generator=g)
    wt = torch.randn(3, 3, ci, co, device=DEV, generator=g) / float(np.sqrt(9 * ci))
    Uc = torch.empty(16 * ci * co, device=DEV); Ucd = torch.empty(16 * ci * co, device=DEV)
    hip.unet_winograd_weight_transform(P(wt), P(Uc), ci, co, 2, ST()); hip.unet_winograd_weight_transform(P(wt), P(Ucd), ci, co, 3, ST())
    y = torch.empty(n, h, w, co, device=DEV); yd = torch.empty_like(y); dx = torch.empty_like(x); dw = torch.empty_like(wt)
    hip.unet_conv3x3_fwd_winograd_fused(P(x), ci, None, P(Uc), None, P(y), co, n, h, w, ci, co, 0, None, 0, ST())
    hip.unet_conv3x3_fwd_mfma(P(x), ci, P(wt), None, P(yd), co, n, h, w, ci, co, 0, ST())
    assert ((y - yd).abs().max() / yd.abs().max()).item() < 2e-5
    hip.unet_conv3x3_dgrad_winograd_fused(P(dz), co, P(Ucd), P(dx), ci, n, h, w, ci, co, None, 0, 0, 0, None, 0, ST())
    nb = hip.unet_conv3x3_wgrad_winograd_fused_workspace(n, h, w, ci, co, 0); ws = ws_bytes(nb)
    hip.unet_conv3x3_wgrad_winograd_fused(P(x), ci, P(dz), co, P(dw), n, h, w, ci, co, 0, P(ws), nb, ST())
    dot = lambda a, b: (a.double() * b.double()).sum().item()
    a0, a1, a2 = dot(y, dz), dot(x, dx), dot(wt, dw)
    scale = float(np.sqrt(dot(y, y) * dot(dz, dz)))
    assert abs(a0 - a1) < 1e-5 * scale and abs(a0 - a2) < 1e-5 * scale


@pytest.mark.parametrize("shape", [(2, 4, 32, 128, 64), (1, 6, 20, 64, 128), (2, 2, 2, 1024, 512)])
def test_convT2x2_fwd_dgrad_wgrad(hip, shape):
    n, h, w, ci, co = shape
    rng = np.random.default_rng(11)
    x = rng.standard_normal((n, ci, h, w))
    wt = rng.standard_normal((2, 2, co, ci)) / np.sqrt(ci)
    b = rng.standard_normal(co)
    dz = rng.standard_normal((n, co, 2 * h, 2 * w))
    z_ref = on.deconv2x2_fwd(x, wt, b)
    dx_ref, dw_ref, _ = on.deconv2x2_bwd(x, wt, dz)
    xd, wd, bd, dzd = to_nhwc(x), dev(wt), dev(b), to_nhwc(dz)
    # output written into the upper half of a concat buffer
    cat = torch.zeros(n, 2 * h, 2 * w, 2 * co, device=DEV)
    outv = cat[..., co:]
    hip.unet_convT2x2_fwd(P(xd), ci, P(wd), P(bd), P(outv), 2 * co, n, h, w, ci, co, ST())
    assert relerr(from_nhwc(outv), z_ref) < 2e-5
    assert cat[..., :co].abs().max().item() == 0
    dx = torch.empty(n, h, w, ci, device=DEV)
    hip.unet_convT2x2_dgrad(P(dzd), co, P(wd), P(dx), ci, n, h, w, ci, co, ST())
    assert relerr(from_nhwc(dx), dx_ref) < 2e-5
    nb = hip.unet_convT2x2_wgrad_workspace(n, h, w, ci, co)
    ws = ws_bytes(nb)
    dw = torch.empty(2, 2, co, ci, device=DEV)
    hip.unet_convT2x2_wgrad(P(xd), ci, P(dzd), co, P(dw), n, h, w, ci, co, P(ws), nb, ST())
    assert relerr(dw.cpu().numpy().astype(np.float64), dw_ref) < 2e-5


@pytest.mark.parametrize("shape", [(2, 8, 16, 128, 128), (1, 16, 16, 64, 64), (4, 8, 16, 32, 192), (8, 32, 32, 1024, 512), (8, 256, 256, 128, 64)])
def test_convT2x2_fwd_stream_matches_igemm_and_oracle(hip, shape):
    # persistent stream kernel for the transposed-conv forward: bit-level agreement is not expected (different summation
    # order), so compare with the first kernel at 2e-5 and, on the small shapes, with the fp64 oracle; the last two shapes are
    # BASELINE config-2 layers (up_4, up_1): 2-8 tiles per workgroup, 128 / 16 chunks per tile
    n, h, w, ci, co = shape
    assert hip.unet_convT2x2_fwd_stream_supported(n, h, w, ci, co) == 1
    g = torch.Generator(device=DEV); g.manual_seed(ci + co + h)
    x = torch.randn(n, h, w, ci + 4, device=DEV, generator=g)[..., :ci]                  # ld = ci + 4
    wt = torch.randn(2, 2, co, ci, device=DEV, generator=g) / float(np.sqrt(ci)); b = torch.randn(co, device=DEV, generator=g)
    cat = torch.zeros(n, 2 * h, 2 * w, 2 * co, device=DEV)
    out = cat[..., co:]
    hip.unet_convT2x2_fwd_stream(P(x), ci + 4, P(wt), P(b), P(out), 2 * co, n, h, w, ci, co, ST())
    ref = torch.empty(n, 2 * h, 2 * w, co, device=DEV)
    hip.unet_convT2x2_fwd(P(x), ci + 4, P(wt), P(b), P(ref), co, n, h, w, ci, co, ST())
    assert ((out - ref).abs().max() / ref.abs().max()).item() < 2e-5
    assert cat[..., :co].abs().max().item() == 0
    if n * h * w * ci * co < 5e7:
        z_ref = on.deconv2x2_fwd(from_nhwc(x.contiguous()), wt.cpu().numpy().astype(np.float64), b.cpu().numpy().astype(np.float64))
        assert relerr(from_nhwc(out), z_ref) < 2e-5


@pytest.mark.parametrize("cin,w", [(1, 21), (3, 21), (1, 24), (2, 24), (3, 24), (4, 24), (5, 24), (1, 70), (3, 96), (4, 33)])
def test_first_layer_direct_conv(hip, cin, w):
    # cin <= 4 with 64 output channels takes the matrix-core kernels (32-pixel row segments: widths below, at and across a segment; odd
    # widths), cin = 5 the generic ones
    n, h, co = 2, 18, 64
    rng = np.random.default_rng(cin)
    x = rng.standard_normal((n, cin, h, w)); wt = rng.standard_normal((3, 3, cin, co)); b = rng.standard_normal(co)
    dz = rng.standard_normal((n, co, h, w))
    z_ref = np.maximum(on.conv_same_fwd(x, wt, b), 0)
    _, dw_ref, _ = on.conv_same_bwd(x, wt, dz)
    xn = dev(x)                               # NCHW as the reader delivers it
    if cin == 1:
        xd = xn.view(n, h, w, 1)
    else:
        xd = torch.empty(n, h, w, cin, device=DEV)
        hip.unet_nchw_to_nhwc(P(xn), P(xd), n, cin, h, w, ST())
        assert torch.equal(xd, xn.permute(0, 2, 3, 1))
    wd, bd, dzd = dev(wt), dev(b), to_nhwc(dz)
    out = torch.empty(n, h, w, co, device=DEV)
    hip.unet_conv3x3_fwd_direct(P(xd), cin, P(wd), P(bd), P(out), co, n, h, w, cin, co, 1, ST())
    assert relerr(from_nhwc(out), z_ref) < 1e-5
    nb = hip.unet_conv3x3_wgrad_direct_workspace(n, h, w, cin, co)
    ws = ws_bytes(nb)
    dw = torch.empty(3, 3, cin, co, device=DEV)
    hip.unet_conv3x3_wgrad_direct(P(xd), cin, P(dzd), co, 0, P(dw), n, h, w, cin, co, P(ws), nb, ST())
    assert relerr(dw.cpu().numpy().astype(np.float64), dw_ref) < 1e-5
    # dz stored as bf16: the same (bf16-representable) values as a bf16 and as an fp32 tensor give the same gradient (the bf16 tensor takes
    # the matrix-core kernel for cin <= 3, the fp32 one the stencil kernels: same products, another summation order), both the oracle's
    dz16 = dzd.to(torch.bfloat16); dz32 = dz16.float()
    dwa, dwb = torch.empty_like(dw), torch.empty_like(dw)
    hip.unet_conv3x3_wgrad_direct(P(xd), cin, P(dz16), co, 1, P(dwa), n, h, w, cin, co, P(ws), nb, ST())
    hip.unet_conv3x3_wgrad_direct(P(xd), cin, P(dz32), co, 0, P(dwb), n, h, w, cin, co, P(ws), nb, ST())
    _, dw16_ref, _ = on.conv_same_bwd(x, wt, from_nhwc(dz32))
    assert relerr(dwa.cpu().numpy().astype(np.float64), dw16_ref) < 1e-5 and relerr(dwb.cpu().numpy().astype(np.float64), dw16_ref) < 1e-5
    dwa2 = torch.empty_like(dw)                                    # and bit-reproducible from call to call
    hip.unet_conv3x3_wgrad_direct(P(xd), cin, P(dz16), co, 1, P(dwa2), n, h, w, cin, co, P(ws), nb, ST())
    assert torch.equal(dwa, dwa2)
    rows = hip.unet_conv3x3_fwd_direct_stats_rows(n, h, w, cin, co)
    if rows > 0:
        # + BatchNorm sums, output stored as fp32 or bf16: the bf16 tensor is the fp32 one rounded (nearest even), the sums identical
        # (they are taken before the rounding)
        res = []
        for o16 in (0, 1):
            o = torch.zeros(n, h, w, co, device=DEV, dtype=torch.bfloat16 if o16 else torch.float32)
            part = torch.zeros((co // 64) * rows * 128, device=DEV)
            hip.unet_conv3x3_fwd_direct_stats(P(xd), cin, P(wd), P(bd), P(o), co, o16, n, h, w, cin, co, 1, P(part), part.numel() * 4, ST())
            res.append((o, part))
        assert torch.equal(res[0][0], out) and torch.equal(res[0][0].to(torch.bfloat16), res[1][0]) and torch.equal(res[0][1], res[1][1])
        sums = res[0][1].view(co // 64, rows, 64, 2).sum(1).reshape(co, 2).cpu().numpy().astype(np.float64)
        zr = z_ref.transpose(0, 2, 3, 1).reshape(-1, co)
        assert relerr(sums[:, 0], zr.sum(0)) < 1e-5 and relerr(sums[:, 1], (zr * zr).sum(0)) < 1e-5


@pytest.mark.parametrize("cin,k,n", [(1, 2, 8), (3, 4, 8)])
def test_first_layer_and_class_map_at_full_size_match_torch_convolutions(hip, cin, k, n):
    # The two ends of the network at BASELINE's tile size (8 x 512^2), bf16-stored 64-channel tensors as the mixed-precision step has them,
    # against torch's own fp64 convolutions: the matrix-core first layer (forward + BatchNorm sums + weight gradient, hand-counted waits:
    # a wait one count too lenient shows here as a handful of wrong or irreproducible elements) and the octet-per-lane class-map kernels.
    h = w = 512
    g = torch.Generator(device=DEV).manual_seed(100 + cin)
    x = torch.randn(n, h, w, cin, device=DEV, generator=g); wt = torch.randn(3, 3, cin, 64, device=DEV, generator=g) * 0.3; b = torch.randn(64, device=DEV, generator=g)
    ref = torch.relu(torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), wt.double().permute(3, 2, 0, 1), b.double(), padding=1)).permute(0, 2, 3, 1)
    rows = hip.unet_conv3x3_fwd_direct_stats_rows(n, h, w, cin, 64)
    outs = []
    for rep in range(2):
        o = torch.zeros(n, h, w, 64, device=DEV, dtype=torch.bfloat16); part = torch.zeros(rows * 128, device=DEV)
        hip.unet_conv3x3_fwd_direct_stats(P(x), cin, P(wt), P(b), P(o), 64, 1, n, h, w, cin, 64, 1, P(part), part.numel() * 4, ST())
        outs.append((o, part))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    o32 = torch.zeros(n, h, w, 64, device=DEV); part32 = torch.zeros(rows * 128, device=DEV)
    hip.unet_conv3x3_fwd_direct_stats(P(x), cin, P(wt), P(b), P(o32), 64, 0, n, h, w, cin, 64, 1, P(part32), part32.numel() * 4, ST())
    assert float((o32.double() - ref).abs().max()) < 2e-5 and torch.equal(o32.to(torch.bfloat16), outs[0][0]) and torch.equal(part32, outs[0][1])
    sums = outs[0][1].view(rows, 64, 2).double().sum(0)
    rs = torch.stack([ref.reshape(-1, 64).sum(0), (ref * ref).reshape(-1, 64).sum(0)], 1)
    assert float(((sums - rs).abs() / rs.abs()).max()) < 2e-6
    dz = torch.randn(n, h, w, 64, device=DEV, generator=g).to(torch.bfloat16)
    nb = hip.unet_conv3x3_wgrad_direct_workspace(n, h, w, cin, 64); ws = ws_bytes(nb)
    dws = []
    for rep in range(2):
        dw = torch.empty(3, 3, cin, 64, device=DEV)
        hip.unet_conv3x3_wgrad_direct(P(x), cin, P(dz), 64, 1, P(dw), n, h, w, cin, 64, P(ws), nb, ST())
        dws.append(dw)
    xp = torch.nn.functional.pad(x.double().permute(0, 3, 1, 2), (1, 1, 1, 1)); dzn = dz.double().permute(0, 3, 1, 2)
    dref = torch.stack([torch.stack([torch.einsum("ncyx,nkyx->ck", xp[:, :, a:a + h, bb:bb + w], dzn) for bb in range(3)]) for a in range(3)])
    assert torch.equal(dws[0], dws[1]) and float((dws[0].double() - dref).norm() / dref.norm()) < 2e-6
    # class map on the bf16 64-channel tensor
    pix = n * h * w
    y = torch.randn(n, h, w, 64, device=DEV, generator=g).to(torch.bfloat16); wk = torch.randn(64, k, device=DEV, generator=g) * 0.2; bk = torch.randn(k, device=DEV, generator=g)
    z = torch.empty(n, h, w, k, device=DEV)
    hip.unet_conv1x1_fwd(P(y), 64, 1, P(wk), P(bk), P(z), k, pix, 64, k, 1, ST())
    zref = torch.relu(y.double().reshape(-1, 64) @ wk.double() + bk.double()).reshape(n, h, w, k)
    assert float((z.double() - zref).abs().max()) < 2e-5
    dzk = torch.randn(n, h, w, k, device=DEV, generator=g)
    dx = torch.empty(n, h, w, 64, device=DEV, dtype=torch.bfloat16)
    hip.unet_conv1x1_dgrad(P(dzk), k, P(wk), P(dx), 64, 1, pix, 64, k, ST())
    dxref = (dzk.double().reshape(-1, k) @ wk.double().t()).reshape(n, h, w, 64)
    assert float((dx.double() - dxref).abs().max()) < 2.0 ** -8 * float(dxref.abs().max())          # bf16 storage: half an ulp of the largest element
    nb2 = hip.unet_conv1x1_wgrad_workspace(pix, 64, k); ws2 = ws_bytes(nb2)
    dwk = torch.empty(64, k, device=DEV)
    hip.unet_conv1x1_wgrad(P(y), 64, 1, P(dzk), k, P(dwk), pix, 64, k, P(ws2), nb2, ST())
    dwref = y.double().reshape(-1, 64).t() @ dzk.double().reshape(-1, k)
    assert float((dwk.double() - dwref).norm() / dwref.norm()) < 2e-6


@pytest.mark.parametrize("k", [2, 6, 11])
def test_conv1x1_class_map(hip, k):
    n, h, w, ci = 2, 9, 13, 64
    rng = np.random.default_rng(k)
    x = rng.standard_normal((n, ci, h, w)); wt = rng.standard_normal((1, 1, ci, k)); b = rng.standard_normal(k)
    dz = rng.standard_normal((n, k, h, w))
    z_ref = np.maximum(on.conv_same_fwd(x, wt, b), 0)
    dx_ref, dw_ref, _ = on.conv_same_bwd(x, wt, dz)
    xd, wd, bd, dzd = to_nhwc(x), dev(wt), dev(b), to_nhwc(dz)
    pix = n * h * w
    out = torch.empty(n, h, w, k, device=DEV)
    hip.unet_conv1x1_fwd(P(xd), ci, 0, P(wd), P(bd), P(out), k, pix, ci, k, 1, ST())
    assert relerr(from_nhwc(out), z_ref) < 1e-5
    dx = torch.empty(n, h, w, ci, device=DEV)
    hip.unet_conv1x1_dgrad(P(dzd), k, P(wd), P(dx), ci, 0, pix, ci, k, ST())
    assert relerr(from_nhwc(dx), dx_ref) < 1e-5
    nb = hip.unet_conv1x1_wgrad_workspace(pix, ci, k)
    ws = ws_bytes(nb)
    dw = torch.empty(1, 1, ci, k, device=DEV)
    hip.unet_conv1x1_wgrad(P(xd), ci, 0, P(dzd), k, P(dw), pix, ci, k, P(ws), nb, ST())
    assert relerr(dw.cpu().numpy().astype(np.float64), dw_ref) < 1e-5
    # the Cin-channel side stored as bf16 (bf16 activation storage): same values handed over as bf16 and as fp32 tensors give
    # bit-identical class maps and weight gradients; a bf16 input gradient is the fp32 one rounded
    x16 = xd.to(torch.bfloat16); x32 = x16.float()
    o16, o32 = torch.empty_like(out), torch.empty_like(out)
    hip.unet_conv1x1_fwd(P(x16), ci, 1, P(wd), P(bd), P(o16), k, pix, ci, k, 1, ST())
    hip.unet_conv1x1_fwd(P(x32), ci, 0, P(wd), P(bd), P(o32), k, pix, ci, k, 1, ST())
    assert torch.equal(o16, o32)
    dx16 = torch.empty(n, h, w, ci, device=DEV, dtype=torch.bfloat16)
    hip.unet_conv1x1_dgrad(P(dzd), k, P(wd), P(dx16), ci, 1, pix, ci, k, ST())
    assert torch.equal(dx16, dx.to(torch.bfloat16))
    dw16, dw32 = torch.empty_like(dw), torch.empty_like(dw)
    hip.unet_conv1x1_wgrad(P(x16), ci, 1, P(dzd), k, P(dw16), pix, ci, k, P(ws), nb, ST())
    hip.unet_conv1x1_wgrad(P(x32), ci, 0, P(dzd), k, P(dw32), pix, ci, k, P(ws), nb, ST())
    assert torch.equal(dw16, dw32)


@pytest.mark.parametrize("c,relu", [(64, 1), (256, 0), (2, 1), (6, 1), (4, 1)])
def test_batchnorm_train_eval_backward(hip, c, relu):
    n, h, w = 3, 10, 14
    rng = np.random.default_rng(c)
    r = rng.standard_normal((n, c, h, w)) + 0.3
    if relu:
        r = np.maximum(r, 0)
    gamma = rng.uniform(0.5, 1.5, c); beta = rng.standard_normal(c)
    mm0 = rng.standard_normal(c); mv0 = rng.uniform(0.5, 2, c)
    dy = rng.standard_normal((n, c, h, w))
    y_ref, cache = on.bn_train_fwd(r, gamma, beta, 1e-3)
    dr_ref, dg_ref, db_ref = on.bn_train_bwd(dy, gamma, cache)
    dz_ref = dr_ref * (r > 0) if relu else dr_ref
    pix = n * h * w
    rd = to_nhwc(r)
    cp = (c + 3) // 4 * 4
    stat = torch.zeros(4, cp, device=DEV)
    par = torch.zeros(2, cp, device=DEV); par[0, :c] = dev(gamma); par[1, :c] = dev(beta)
    mov = torch.zeros(2, cp, device=DEV); mov[0, :c] = dev(mm0); mov[1, :c] = dev(mv0)
    nb = hip.unet_bn_workspace(pix, c)
    ws = ws_bytes(nb)
    hip.unet_bn_train_stats(P(rd), c, pix, c, P(par[0]), P(par[1]), 1e-3, 0.99, 1, P(mov[0]), P(mov[1]),
                            P(stat[0]), P(stat[1]), P(stat[2]), P(stat[3]), P(ws), nb, ST())
    mu, var = cache[2], cache[3]
    assert np.abs(stat[0, :c].cpu().numpy() - mu).max() < 1e-6
    assert relerr(stat[1, :c].cpu().numpy(), 1 / np.sqrt(var + 1e-3)) < 2e-6
    assert relerr(mov[0, :c].cpu().numpy(), 0.99 * mm0 + 0.01 * mu) < 2e-6
    assert relerr(mov[1, :c].cpu().numpy(), 0.99 * mv0 + 0.01 * var * pix / (pix - 1)) < 2e-6
    # apply into a channel slice of a wider buffer
    ybuf = torch.zeros(n, h, w, c + 4, device=DEV)
    yv = ybuf[..., 4:]
    hip.unet_bn_apply(P(rd), c, P(stat[2]), P(stat[3]), P(yv), c + 4, pix, c, ST())
    assert relerr(from_nhwc(yv), y_ref) < 5e-6
    # eval-mode coefficients
    hip.unet_bn_eval_coeffs(P(par[0]), P(par[1]), P(mov[0]), P(mov[1]), 1e-3, c, P(stat[2]), P(stat[3]), ST())
    mmn, mvn = mov[0, :c].cpu().numpy().astype(np.float64), mov[1, :c].cpu().numpy().astype(np.float64)
    hip.unet_bn_apply(P(rd), c, P(stat[2]), P(stat[3]), P(yv), c + 4, pix, c, ST())
    assert relerr(from_nhwc(yv), on.bn_eval_fwd(r, gamma, beta, mmn, mvn, 1e-3)) < 5e-6
    # backward
    dyd = to_nhwc(dy)
    dz = torch.empty(n, h, w, c, device=DEV)
    gr = torch.zeros(3, cp, device=DEV)
    hip.unet_bn_bwd(P(dyd), c, P(rd), c, P(par[0]), P(stat[0]), P(stat[1]), pix, c, relu, P(dz), c, P(gr[0]), P(gr[1]),
                    P(gr[2]), P(ws), nb, ST())
    assert relerr(gr[0, :c].cpu().numpy(), dg_ref) < 1e-5
    assert relerr(gr[1, :c].cpu().numpy(), db_ref) < 1e-5
    assert relerr(from_nhwc(dz), dz_ref) < 1e-5
    assert np.abs(gr[2, :c].cpu().numpy() - dz_ref.sum(axis=(0, 2, 3))).max() < 1e-4 * np.abs(dz_ref).sum(axis=(0, 2, 3)).max()


def test_bn_apply_maxpool_fused_equals_separate(hip):
    # one pass == unet_bn_apply followed by unet_maxpool2x2_fwd, bit for bit (values, pooled, first-max indices incl. ties)
    n, h, w, c = 2, 12, 20, 64
    g = torch.Generator(device=DEV); g.manual_seed(1)
    r = torch.relu(torch.randn(n, h, w, c, device=DEV, generator=g))          # post-ReLU: plenty of exact ties at 0
    sc = torch.rand(c, device=DEV, generator=g) + 0.5; sh = torch.zeros(c, device=DEV)
    cat1 = torch.zeros(n, h, w, 2 * c, device=DEV); cat2 = torch.zeros_like(cat1)
    p1 = torch.empty(n, h // 2, w // 2, c, device=DEV); p2 = torch.empty_like(p1)
    i1 = torch.empty(n, h // 2, w // 2, c, dtype=torch.uint8, device=DEV); i2 = torch.empty_like(i1)
    y1, y2 = cat1[..., :c], cat2[..., :c]
    hip.unet_bn_apply_maxpool(P(r), c, P(sc), P(sh), P(y1), 2 * c, P(p1), c, P(i1), n, h, w, c, ST())
    hip.unet_bn_apply(P(r), c, P(sc), P(sh), P(y2), 2 * c, n * h * w, c, ST())
    hip.unet_maxpool2x2_fwd(P(y2), 2 * c, P(p2), c, P(i2), n, h, w, c, 0, ST())
    assert torch.equal(cat1, cat2) and torch.equal(p1, p2) and torch.equal(i1, i2)
    assert (i1 == 0).float().mean().item() > 0.27                              # 0.234 strict wins + 1/16 all-zero windows (ties -> first position)


def test_bn_bwd_pooled_equals_pool_backward_then_bn_bwd(hip):
    # dy = skip gradient + un-pooled gradient formed inside the BatchNorm-backward kernels == the separate pool-backward pass
    # (accumulating into the skip gradient) followed by unet_bn_bwd
    n, h, w, c = 2, 12, 20, 64
    g = torch.Generator(device=DEV); g.manual_seed(4)
    r = torch.relu(torch.randn(n, h, w, c, device=DEV, generator=g))
    y = r * 1.3
    pooled = torch.empty(n, h // 2, w // 2, c, device=DEV); idx = torch.empty(n, h // 2, w // 2, c, dtype=torch.uint8, device=DEV)
    hip.unet_maxpool2x2_fwd(P(y), c, P(pooled), c, P(idx), n, h, w, c, 0, ST())
    dcat = torch.randn(n, h, w, 2 * c, device=DEV, generator=g); dskip = dcat[..., :c]
    pdy = torch.randn(n, h // 2, w // 2, c, device=DEV, generator=g)
    gm = torch.rand(c, device=DEV, generator=g) + 0.5
    mean = r.mean((0, 1, 2)).contiguous(); invstd = (1.0 / torch.sqrt(r.var((0, 1, 2), unbiased=False) + 1e-3)).contiguous()
    npx = n * h * w
    nb = hip.unet_bn_workspace(npx, c); ws = ws_bytes(nb)
    out = []
    for fused in (True, False):
        dz = torch.empty(n, h, w, c, device=DEV); dg, db, dbias = [torch.empty(c, device=DEV) for _ in range(3)]
        if fused:
            hip.unet_bn_bwd_pooled(P(dskip), 2 * c, P(pdy), c, P(idx), n, h, w, P(r), c, P(gm), P(mean), P(invstd), c, 1,
                                   P(dz), c, P(dg), P(db), P(dbias), P(ws), nb, ST())
        else:
            d2 = dcat.clone(); ds2 = d2[..., :c]
            hip.unet_maxpool2x2_bwd(P(pdy), c, P(idx), P(ds2), 2 * c, n, h, w, c, 1, 0, ST())
            hip.unet_bn_bwd(P(ds2), 2 * c, P(r), c, P(gm), P(mean), P(invstd), npx, c, 1, P(dz), c, P(dg), P(db), P(dbias), P(ws), nb, ST())
        out.append((dz, dg, db, dbias))
    # (the fused form walks one 2x2 window per lane, the plain form one pixel per lane: the per-channel sums are added in a different
    # order, so equality holds to summation-order rounding, not bit for bit)
    for a_, b_ in zip(out[0], out[1]):
        assert (a_ - b_).abs().max().item() <= 2e-6 * b_.abs().max().item() + 1e-7


def test_maxpool_fwd_bwd_with_ties(hip):
    n, h, w, c = 2, 8, 12, 64
    rng = np.random.default_rng(0)
    x = np.round(rng.standard_normal((n, c, h, w)) * 1.5)           # many exact ties
    x[0, :, 0:2, 0:2] = 0.25                                       # an all-equal window
    dy = rng.standard_normal((n, c, h // 2, w // 2))
    y_ref, idx_ref = on.maxpool2x2_fwd(x)
    dx_ref = on.maxpool2x2_bwd(dy, idx_ref)
    base = rng.standard_normal((n, c, h, w))
    cat = torch.zeros(n, h, w, 2 * c, device=DEV); cat[..., :c] = to_nhwc(x)
    xv = cat[..., :c]
    y = torch.empty(n, h // 2, w // 2, c, device=DEV)
    idx = torch.empty(n, h // 2, w // 2, c, dtype=torch.uint8, device=DEV)
    hip.unet_maxpool2x2_fwd(P(xv), 2 * c, P(y), c, P(idx), n, h, w, c, 0, ST())
    assert np.array_equal(from_nhwc(y), y_ref)
    assert np.array_equal(idx.cpu().numpy().transpose(0, 3, 1, 2), idx_ref)
    dcat = torch.zeros(n, h, w, 2 * c, device=DEV); dcat[..., :c] = to_nhwc(base)
    dv = dcat[..., :c]
    hip.unet_maxpool2x2_bwd(P(to_nhwc(dy)), c, P(idx), P(dv), 2 * c, n, h, w, c, 1, 0, ST())
    assert relerr(from_nhwc(dv), dx_ref + base) < 1e-6
    assert dcat[..., c:].abs().max().item() == 0
    # the same tensors stored as bf16 (level 4 of the bf16 mode): pooling picks existing values -> equal to the fp32 kernel on the
    # bf16 values, indices included; the backward accumulate rounds its one sum per element to bf16
    cat16 = cat.to(torch.bfloat16); y16 = torch.empty(n, h // 2, w // 2, c, device=DEV, dtype=torch.bfloat16); idx16 = torch.empty_like(idx)
    hip.unet_maxpool2x2_fwd(P(cat16[..., :c]), 2 * c, P(y16), c, P(idx16), n, h, w, c, 1, ST())
    y32 = torch.empty_like(y); idx32 = torch.empty_like(idx); cat32 = cat16.float()
    hip.unet_maxpool2x2_fwd(P(cat32[..., :c]), 2 * c, P(y32), c, P(idx32), n, h, w, c, 0, ST())
    assert torch.equal(y16.float(), y32) and torch.equal(idx16, idx32)
    dy16 = to_nhwc(dy).to(torch.bfloat16); dcat16 = dcat.clone().zero_(); dcat16[..., :c] = to_nhwc(base); dcat16 = dcat16.to(torch.bfloat16)
    ref16 = dcat16.float().clone()
    hip.unet_maxpool2x2_bwd(P(dy16.float()), c, P(idx), P(ref16[..., :c]), 2 * c, n, h, w, c, 1, 0, ST())
    hip.unet_maxpool2x2_bwd(P(dy16), c, P(idx), P(dcat16[..., :c]), 2 * c, n, h, w, c, 1, 1, ST())
    assert torch.equal(dcat16, ref16.to(torch.bfloat16))


def test_dropout_mask_and_rng(hip):
    pix, c = 4096, 512
    x = torch.randn(pix, c, device=DEV)
    mask = (torch.rand(pix, c, device=DEV) < 0.5).to(torch.uint8)
    out = torch.empty_like(x)
    hip.unet_dropout(P(x), c, P(out), c, pix, c, P(mask), 0, 0.5, 0, ST())
    assert torch.equal(out, x * mask.float() * 2.0)
    o1, o2, o3 = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    hip.unet_dropout(P(x), c, P(o1), c, pix, c, None, 123, 0.5, 0, ST())
    hip.unet_dropout(P(x), c, P(o2), c, pix, c, None, 123, 0.5, 0, ST())
    hip.unet_dropout(P(x), c, P(o3), c, pix, c, None, 124, 0.5, 0, ST())
    assert torch.equal(o1, o2)                                    # same (seed, index) -> same mask (backward regenerates it)
    x16 = x.to(torch.bfloat16); o16 = torch.empty_like(x16)       # bf16 tensors: the same mask, the kept values doubled exactly
    hip.unet_dropout(P(x16), c, P(o16), c, pix, c, None, 123, 0.5, 1, ST())
    assert torch.equal(o16 != 0, (o1 != 0) & (x16 != 0)) or torch.equal((o16 != 0) | (x16 == 0), (o1 != 0) | (x16 == 0))
    assert torch.equal(o16[o16 != 0].float(), (x16.float() * 2.0)[o16 != 0])
    keep = (o1 != 0).float().mean().item()
    assert abs(keep - 0.5) < 0.005
    assert (o1 != o3).float().mean().item() > 0.3
    kept = o1 != 0
    assert torch.equal(o1[kept], (x * 2.0)[kept])


@pytest.mark.parametrize("k,ls,clip", [(2, 0.0, 0.0), (4, 0.0, 0.0), (6, 0.1, 0.0), (2, 0.0, 1e-7), (4, 0.1, 1e-7), (3, 0.0, 1e-3)])
def test_softmax_ce_accuracy_argmax(hip, k, ls, clip):
    # clip = 0: cross-entropy from the softmax's logits; clip > 0: Keras' clipped-probability path (Contract.ce_from_softmax_logits
    # False) -- logits scaled so that some probabilities really fall outside [eps, 1-eps] and take the zero-gradient branch
    n, h, w = 2, 16, 24
    rng = np.random.default_rng(k)
    z = rng.standard_normal((n, h, w, k)) * (3 if clip == 0.0 else 14)
    cls = rng.integers(0, k, (n, h, w))
    lab = (cls[..., None] == np.arange(k)).astype(np.int32)
    G = 4
    contract = on.Contract(ce_from_softmax_logits=(clip == 0.0), ce_clip_eps=clip if clip else 1e-7)
    if clip:
        # fp32 cannot resolve the clip edges exactly (1 - 1e-7 is 0.99999988 in fp32, as it is for TF's own fp32 constants), so a
        # probability within rounding of an edge may legitimately land on the other side: move such pixels to uniform logits
        p0 = on.softmax_lastaxis(z)
        q0 = 1 - p0
        near = (np.abs(p0 - clip) < 1e-3 * clip) | ((q0 > 0.4 * clip) & (q0 < 3 * clip) if clip < 1e-5 else (np.abs(q0 - clip) < 1e-3 * clip))
        z[near.any(-1)] = 0.0
    loss_ref, p_ref, y = on.ce_loss_fwd(z, lab, G, ls, contract)
    dl_ref = on.ce_loss_bwd(p_ref, y, G, contract)
    if clip:
        assert ((p_ref < clip) | (p_ref > 1 - clip)).any()
    zd, labd = dev(z), dev(lab, np.int32)
    pix = n * h * w
    prob = torch.empty(n, h, w, k, device=DEV); dl = torch.empty(n, h, w, k, device=DEV)
    res = torch.zeros(2, device=DEV)
    nb = hip.unet_softmax_ce_workspace(pix)
    ws = ws_bytes(nb)
    s = 1.0 / (G * h * w)
    hip.unet_softmax_ce(P(zd), k, P(labd), P(prob), P(dl), k, pix, k, ls, s, s, clip, P(res[0:1]), P(res[1:2]), P(ws), nb, ST())
    assert relerr(prob.cpu().numpy(), p_ref) < 2e-6
    assert relerr(dl.cpu().numpy(), dl_ref) < 5e-6
    assert abs(res[0].item() - loss_ref) < (2e-6 if clip == 0.0 else 2e-5) * abs(loss_ref)   # (log(1 - 1e-7) carries fp32's 6e-8 ulp of 1)
    assert res[1].item() == float((np.argmax(p_ref, -1) == cls).sum())
    am = torch.empty(n, h, w, dtype=torch.int32, device=DEV)
    hip.unet_argmax(P(prob), k, P(am), pix, k, ST())
    assert np.array_equal(am.cpu().numpy(), np.argmax(prob.cpu().numpy(), -1))
    # exact tie -> first index (np.argmax semantics, reference UNet/inference.py:107)
    tie = torch.zeros(1, 1, 4, k, device=DEV)
    hip.unet_argmax(P(tie), k, P(am), 4, k, ST())
    assert am.view(-1)[:4].abs().max().item() == 0


def test_keras_adam_flat(hip):
    n = 4 * 1000
    rng = np.random.default_rng(0)
    th = rng.standard_normal(n); g = rng.standard_normal(n) * 10.0 ** rng.uniform(-7, 0, n)
    m = np.zeros(n); v = np.zeros(n)
    thd, gd, md, vd = dev(th), dev(g), dev(m), dev(v)
    c = on.Contract()
    ref_th, ref_m, ref_v = th.astype(np.float32).astype(np.float64), m, v
    g32 = g.astype(np.float32).astype(np.float64)
    for t in (1, 2, 3):
        alpha = 3e-4 * np.sqrt(1 - c.adam_beta2 ** t) / (1 - c.adam_beta1 ** t)
        hip.unet_adam_keras(P(thd), P(gd), P(md), P(vd), n, alpha, 0.9, 0.999, 1e-7, ST())
        ref_th, ref_m, ref_v = on.adam_keras_step(ref_th, g32, ref_m, ref_v, t, 3e-4, c)
    assert np.abs(thd.cpu().numpy() - ref_th).max() < 2e-6
    assert relerr(vd.cpu().numpy(), ref_v) < 1e-5


def test_winograd_weight_transform_batch_matches_single(hip):
    # one launch for several layers == the per-layer transforms: mode 2 bit for bit; mode 3 (rotated filter) is obtained by
    # permuting the forward transform's points, which sums the taps in the other order -> equal to rounding
    shapes = [(64, 64), (128, 64), (8, 192), (72, 64)]
    g = torch.Generator(device=DEV); g.manual_seed(3)
    ws = [torch.randn(3, 3, ci, co, device=DEV, generator=g) for ci, co in shapes]
    uf = [torch.empty(16 * ci * co, device=DEV) for ci, co in shapes]; ud = [torch.empty(16 * ci * co, device=DEV) for ci, co in shapes]
    rows, blk = [], 0
    for (ci, co), w_, a, b in zip(shapes, ws, uf, ud):
        rows.append([w_.data_ptr(), a.data_ptr(), b.data_ptr(), ci | (co << 32), blk, 0]); blk += (ci * co + 2047) // 2048
    jobs = torch.tensor(rows, dtype=torch.int64, device=DEV)
    hip.unet_winograd_weight_transform_batch(P(jobs), len(shapes), blk, ST())
    for (ci, co), w_, a, b in zip(shapes, ws, uf, ud):
        ra = torch.empty_like(a); rb = torch.empty_like(b)
        hip.unet_winograd_weight_transform(P(w_), P(ra), ci, co, 2, ST()); hip.unet_winograd_weight_transform(P(w_), P(rb), ci, co, 3, ST())
        assert torch.equal(a, ra) and torch.allclose(b, rb, rtol=0, atol=1e-6)


@pytest.mark.parametrize("shape", [(2, 16, 16, 64, 64), (1, 16, 32, 128, 128), (2, 32, 48, 64, 128), (1, 6, 10, 72, 64),
                                   (1, 20, 36, 256, 128), (1, 16, 16, 32, 64), (5, 104, 136, 64, 64)])
def test_conv3x3_winograd_fully_fused_fwd_dgrad(hip, shape):
    # raw patch -> LDS, in-kernel transform, 16-point MFMA, output transform in the epilogue; ragged tile grids included.
    # 72 channels = odd chunk count (one-tile-per-workgroup kernel); 32 channels = the shortest chunk stream the persistent
    # kernel takes; the last shape has 630 tile blocks, so persistent workgroups walk 2-3 tiles each across image borders
    n, h, w, ci, co = shape
    rng = np.random.default_rng(ci + 7 * co + h)
    x = rng.standard_normal((n, ci, h, w))
    wt = rng.standard_normal((3, 3, ci, co)) / np.sqrt(9 * ci)
    b = rng.standard_normal(co)
    z_ref = np.maximum(on.conv_same_fwd(x, wt, b), 0)
    xbuf = torch.zeros(n, h, w, ci + 8, device=DEV); xbuf[..., 4:4 + ci] = to_nhwc(x)
    xv = xbuf[..., 4:4 + ci]
    wd, bd = dev(wt), dev(b)
    Uc = torch.empty(16 * ci * co, device=DEV)
    hip.unet_winograd_weight_transform(P(wd), P(Uc), ci, co, 2, ST())
    cat = torch.zeros(n, h, w, 2 * co, device=DEV)
    outv = cat[..., co:]
    hip.unet_conv3x3_fwd_winograd_fused(P(xv), ci + 8, None, P(Uc), P(bd), P(outv), 2 * co, n, h, w, ci, co, 1, None, 0, ST())
    assert relerr(from_nhwc(outv), z_ref) < 3e-5
    assert cat[..., :co].abs().max().item() == 0
    if ci % 64 == 0:
        dz = rng.standard_normal((n, co, h, w))
        dx_ref, _, _ = on.conv_same_bwd(x, wt, dz)
        Ucd = torch.empty(16 * ci * co, device=DEV)
        hip.unet_winograd_weight_transform(P(wd), P(Ucd), ci, co, 3, ST())
        dx = torch.full((n, h, w, ci), 7.0, device=DEV)
        hip.unet_conv3x3_dgrad_winograd_fused(P(to_nhwc(dz)), co, P(Ucd), P(dx), ci, n, h, w, ci, co, None, 0, 0, 0, None, 0, ST())
        assert relerr(from_nhwc(dx), dx_ref) < 3e-5


@pytest.mark.parametrize("shape", [(2, 16, 16, 64, 64), (1, 16, 32, 128, 128), (5, 104, 136, 64, 64), (2, 32, 48, 64, 192)])
def test_fused_conv_bn_stats_match_bn_train_stats(hip, shape):
    # BatchNorm sums written by the conv epilogue -> finalize == the separate statistics pass over the stored activation
    # (mean / invstd / scale / shift / moving statistics), ragged tile grids and multi-tile workgroups included
    n, h, w, ci, co = shape
    rows = hip.unet_conv3x3_fwd_winograd_fused_stats_rows(n, h, w, ci, co)
    if co // 64 == 3:
        assert rows == 0 or (n * ((h // 2 + 7) // 8) * ((w // 2 + 7) // 8) * 3) % 3 == 0
    if rows == 0:
        pytest.skip("persistent grid not a multiple of the n-tile count for this shape")
    g = torch.Generator(device=DEV); g.manual_seed(co + h)
    x = torch.randn(n, h, w, ci, device=DEV, generator=g); wt = torch.randn(3, 3, ci, co, device=DEV, generator=g) / float(np.sqrt(9 * ci))
    b = torch.randn(co, device=DEV, generator=g); gm = torch.rand(co, device=DEV, generator=g) + 0.5; bt = torch.randn(co, device=DEV, generator=g)
    Uc = torch.empty(16 * ci * co, device=DEV)
    hip.unet_winograd_weight_transform(P(wt), P(Uc), ci, co, 2, ST())
    r = torch.empty(n, h, w, co, device=DEV)
    part = torch.zeros((co // 64) * rows * 128, device=DEV)
    hip.unet_conv3x3_fwd_winograd_fused(P(x), ci, None, P(Uc), P(b), P(r), co, n, h, w, ci, co, 1, P(part), part.numel() * 4, ST())
    r2 = torch.empty_like(r)
    hip.unet_conv3x3_fwd_winograd_fused(P(x), ci, None, P(Uc), P(b), P(r2), co, n, h, w, ci, co, 1, None, 0, ST())
    assert torch.equal(r, r2)
    npx = n * h * w
    outs = [[torch.zeros(co, device=DEV) for _ in range(4)] for _ in range(2)]
    mm = [torch.zeros(co, device=DEV) for _ in range(2)]; mv = [torch.ones(co, device=DEV) for _ in range(2)]
    hip.unet_bn_train_finalize_partials(P(part), rows, npx, co, P(gm), P(bt), 1e-3, 0.99, 1, P(mm[0]), P(mv[0]),
                                        P(outs[0][0]), P(outs[0][1]), P(outs[0][2]), P(outs[0][3]), ST())
    nb = hip.unet_bn_workspace(npx, co); ws = ws_bytes(nb)
    hip.unet_bn_train_stats(P(r), co, npx, co, P(gm), P(bt), 1e-3, 0.99, 1, P(mm[1]), P(mv[1]),
                            P(outs[1][0]), P(outs[1][1]), P(outs[1][2]), P(outs[1][3]), P(ws), nb, ST())
    for a_, b_ in list(zip(outs[0], outs[1])) + [(mm[0], mm[1]), (mv[0], mv[1])]:
        assert torch.allclose(a_, b_, rtol=2e-5, atol=2e-6), (a_ - b_).abs().max().item()


@pytest.mark.parametrize("shape", [(2, 8, 16, 128, 128), (1, 16, 16, 64, 64), (8, 64, 64, 512, 256), (8, 256, 256, 128, 64)])
def test_convT_stream_bn_stats_match_bn_train_stats(hip, shape):
    # BatchNorm sums from the transposed-conv stream kernel's epilogue == the separate statistics pass
    n, h, w, ci, co = shape
    rows = hip.unet_convT2x2_fwd_stream_stats_rows(n, h, w, ci, co)
    assert rows > 0
    g = torch.Generator(device=DEV); g.manual_seed(co + h)
    x = torch.randn(n, h, w, ci, device=DEV, generator=g); wt = torch.randn(2, 2, co, ci, device=DEV, generator=g) / float(np.sqrt(ci))
    b = torch.randn(co, device=DEV, generator=g); gm = torch.rand(co, device=DEV, generator=g) + 0.5; bt = torch.randn(co, device=DEV, generator=g)
    r = torch.empty(n, 2 * h, 2 * w, co, device=DEV); r2 = torch.empty_like(r)
    part = torch.zeros((co // 64) * rows * 128, device=DEV)
    hip.unet_convT2x2_fwd_stream_stats(P(x), ci, P(wt), P(b), P(r), co, n, h, w, ci, co, P(part), part.numel() * 4, ST())
    hip.unet_convT2x2_fwd_stream(P(x), ci, P(wt), P(b), P(r2), co, n, h, w, ci, co, ST())
    assert torch.equal(r, r2)
    npx = n * 4 * h * w
    outs = [[torch.zeros(co, device=DEV) for _ in range(4)] for _ in range(2)]
    mm = [torch.zeros(co, device=DEV) for _ in range(2)]; mv = [torch.ones(co, device=DEV) for _ in range(2)]
    hip.unet_bn_train_finalize_partials(P(part), rows, npx, co, P(gm), P(bt), 1e-3, 0.99, 1, P(mm[0]), P(mv[0]),
                                        P(outs[0][0]), P(outs[0][1]), P(outs[0][2]), P(outs[0][3]), ST())
    nb = hip.unet_bn_workspace(npx, co); ws = ws_bytes(nb)
    hip.unet_bn_train_stats(P(r), co, npx, co, P(gm), P(bt), 1e-3, 0.99, 1, P(mm[1]), P(mv[1]),
                            P(outs[1][0]), P(outs[1][1]), P(outs[1][2]), P(outs[1][3]), P(ws), nb, ST())
    for a_, b_ in list(zip(outs[0], outs[1])) + [(mm[0], mm[1]), (mv[0], mv[1])]:
        assert torch.allclose(a_, b_, rtol=2e-5, atol=2e-6), (a_ - b_).abs().max().item()


@pytest.mark.parametrize("shape,crange", [((2, 16, 16, 64, 64), (0, 64)), ((1, 16, 32, 128, 64), (64, 128)), ((5, 104, 136, 64, 64), (0, 64)),
                                          ((2, 32, 32, 256, 128), (128, 256))])
def test_dgrad_bn_backward_sums_match_reduction(hip, shape, crange):
    # data gradient + (sum dy, sum dy*r) of the producer layer's BatchNorm from the epilogue -> bn_bwd_from_partials ==
    # the plain data gradient followed by the full unet_bn_bwd (dz, dgamma, dbeta, dbias), for a channel sub-range too
    n, h, w, ci, co = shape
    c0, c1 = crange
    cp = c1 - c0
    rows = hip.unet_conv3x3_fwd_winograd_fused_stats_rows(n, h, w, co, ci)
    assert rows > 0
    g = torch.Generator(device=DEV); g.manual_seed(ci + co + h)
    dzin = torch.randn(n, h, w, co, device=DEV, generator=g); wt = torch.randn(3, 3, ci, co, device=DEV, generator=g) / float(np.sqrt(9 * co))
    r_prev = torch.relu(torch.randn(n, h, w, cp, device=DEV, generator=g))
    gm = torch.rand(cp, device=DEV, generator=g) + 0.5
    mean = r_prev.mean((0, 1, 2)).contiguous(); invstd = (1.0 / torch.sqrt(r_prev.var((0, 1, 2), unbiased=False) + 1e-3)).contiguous()
    Ucd = torch.empty(16 * ci * co, device=DEV)
    hip.unet_winograd_weight_transform(P(wt), P(Ucd), ci, co, 3, ST())
    dx = torch.empty(n, h, w, ci, device=DEV); dx2 = torch.empty_like(dx)
    part = torch.zeros((ci // 64) * rows * 128, device=DEV)
    hip.unet_conv3x3_dgrad_winograd_fused(P(dzin), co, P(Ucd), P(dx), ci, n, h, w, ci, co, P(r_prev), cp, c0, c1,
                                                  P(part), part.numel() * 4, ST())
    hip.unet_conv3x3_dgrad_winograd_fused(P(dzin), co, P(Ucd), P(dx2), ci, n, h, w, ci, co, None, 0, 0, 0, None, 0, ST())
    assert torch.equal(dx, dx2)
    npx = n * h * w
    dy = dx[..., c0:c1]
    nb = hip.unet_bn_workspace(npx, cp); ws = ws_bytes(nb)
    res = []
    for fused in (True, False):
        dz = torch.empty(n, h, w, cp, device=DEV); dg, db, dbias = [torch.empty(cp, device=DEV) for _ in range(3)]
        if fused:
            import ctypes
            hip.unet_bn_bwd_from_partials(P(dy), ci, P(r_prev), cp, P(gm), P(mean), P(invstd), npx, cp, 1, P(dz), cp, P(dg), P(db), P(dbias),
                                          ctypes.c_void_p(part.data_ptr() + (c0 // 64) * rows * 128 * 4), rows, P(ws), nb, ST())
        else:
            hip.unet_bn_bwd(P(dy), ci, P(r_prev), cp, P(gm), P(mean), P(invstd), npx, cp, 1, P(dz), cp, P(dg), P(db), P(dbias), P(ws), nb, ST())
        res.append((dz, dg, db, dbias))
    for a_, b_ in zip(res[0], res[1]):
        scale = b_.abs().max().item() + 1e-30
        assert (a_ - b_).abs().max().item() < 2e-5 * scale + 1e-6, (a_ - b_).abs().max().item() / scale


@pytest.mark.parametrize("shape", [(2, 16, 16, 64, 64), (1, 16, 32, 128, 64), (3, 8, 24, 64, 192), (1, 6, 10, 64, 64), (2, 32, 48, 128, 128),
                                   # (round 5: buffer loads with range-check zeros and scalar chunk bases) one tile row = top AND bottom border in every chunk;
                                   # one 2-pixel column; a chunk row of 16 + 2 columns; more workgroups than chunks; one split over 256 channel blocks
                                   (1, 2, 2, 64, 64), (1, 2, 34, 64, 64), (2, 4, 18, 64, 128), (1, 4, 4, 128, 128), (1, 8, 8, 1024, 1024)])
def test_conv3x3_winograd_fused_wgrad(hip, shape):
    # raw rows through LDS, per-lane Winograd transforms in registers, G^T dU G in the epilogue; ragged tile rows included
    n, h, w, ci, co = shape
    assert hip.unet_winograd_wgrad_fused_supported(n, h, w, ci, co) == 1
    rng = np.random.default_rng(ci * 5 + co + h)
    x = rng.standard_normal((n, ci, h, w)); dz = rng.standard_normal((n, co, h, w))
    _, dw_ref, _ = on.conv_same_bwd(x, np.zeros((3, 3, ci, co)), dz)
    xbuf = torch.zeros(n, h, w, ci + 4, device=DEV); xbuf[..., 4:] = to_nhwc(x)
    xv = xbuf[..., 4:]
    dzd = to_nhwc(dz)
    nb = hip.unet_conv3x3_wgrad_winograd_fused_workspace(n, h, w, ci, co, 0)
    ws = ws_bytes(nb)
    dw = torch.full((3, 3, ci, co), 7.0, device=DEV)
    hip.unet_conv3x3_wgrad_winograd_fused(P(xv), ci + 4, P(dzd), co, P(dw), n, h, w, ci, co, 0, P(ws), nb, ST())
    assert relerr(dw.cpu().numpy().astype(np.float64), dw_ref) < 3e-5


def bf16_round(a):
    """round-to-nearest-even to bf16, returned as float64 (what v_cvt_pk_bf16_f32 does to finite values)"""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return u.view(np.float32).astype(np.float64)


@pytest.mark.parametrize("shape", [(1, 16, 32, 64, 64), (2, 20, 40, 64, 128), (1, 32, 32, 128, 64), (1, 48, 64, 192, 256), (2, 7, 9, 64, 64)])
def test_conv3x3_bf16_fwd_dgrad_match_oracle_on_rounded_operands(hip, shape):
    # bf16 matrix-core path: with both operands rounded to bf16 the products are exact, so the fp64 oracle on the ROUNDED
    # operands must agree to fp32-accumulation accuracy (2e-5 of the output range); against the unrounded oracle the
    # difference is the rounding itself (2^-9 per operand, checked loosely).  Ragged tiles, padded leading dimensions, both
    # channel-tile widths.
    n, h, w, ci, co = shape
    assert hip.unet_conv3x3_bf16_supported(n, h, w, ci, co) == 1
    rng = np.random.default_rng(ci + co + h)
    x = rng.standard_normal((n, ci, h, w)).astype(np.float32)
    wt = (rng.standard_normal((3, 3, ci, co)) / np.sqrt(9 * ci)).astype(np.float32)
    b = rng.standard_normal(co).astype(np.float32)
    dz = rng.standard_normal((n, co, h, w)).astype(np.float32)
    z_ref = on.conv_same_fwd(bf16_round(x), bf16_round(wt), b.astype(np.float64))
    z_full = on.conv_same_fwd(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64))
    dx_ref, _, _ = on.conv_same_bwd(bf16_round(x), bf16_round(wt), bf16_round(dz))
    xd = torch.zeros(n, h, w, ci + 8, device=DEV); xd[..., :ci] = to_nhwc(x); xv = xd[..., :ci]
    dzd = to_nhwc(dz)
    wd, bd = dev(wt), dev(b)
    nb = hip.unet_conv3x3_bf16_packed_bytes(ci, co)
    wp, wpd = ws_bytes(nb), ws_bytes(nb)
    hip.unet_conv3x3_bf16_pack_weights(P(wd), P(wp), ci, co, 0, ST())
    hip.unet_conv3x3_bf16_pack_weights(P(wd), P(wpd), ci, co, 1, ST())
    cat = torch.full((n, h, w, co + 4), float("nan"), device=DEV)
    hip.unet_conv3x3_fwd_bf16(P(xv), ci + 8, 0, None, None, P(wp), P(bd), P(cat), co + 4, 0, n, h, w, ci, co, 0, None, 0, ST())
    z = from_nhwc(cat[..., :co])
    assert relerr(z, z_ref) < 2e-5
    assert relerr(z, z_full) < 2e-2
    assert torch.isnan(cat[..., co:]).all()
    hip.unet_conv3x3_fwd_bf16(P(xv), ci + 8, 0, None, None, P(wp), P(bd), P(cat), co + 4, 0, n, h, w, ci, co, 1, None, 0, ST())
    assert relerr(from_nhwc(cat[..., :co]), np.maximum(z_ref, 0)) < 2e-5
    dx = torch.empty(n, h, w, ci, device=DEV)
    hip.unet_conv3x3_dgrad_bf16(P(dzd), co, 0, P(wpd), P(dx), ci, 0, n, h, w, ci, co, None, 0, 0, 0, 0, None, 0, ST())
    assert relerr(from_nhwc(dx), dx_ref) < 2e-5


@pytest.mark.parametrize("shape", [(1, 4, 32, 64, 64), (2, 6, 40, 64, 128), (1, 7, 33, 128, 64), (1, 64, 32, 64, 64), (2, 4, 32, 1024, 1024),
                                   (3, 10, 70, 192, 64)])
def test_conv3x3_bf16_wgrad_matches_oracle_on_rounded_operands(hip, shape):
    # weight gradient on the bf16 matrix cores (transposing LDS reads): fp64 oracle on the bf16-rounded operands to
    # fp32-accumulation accuracy.  Cases: one strip; ragged width; odd height and width; row chunks split over workgroups
    # (chunk boundaries); one workgroup walking several strips (256 channel pairs); three strips with a ragged last one.
    n, h, w, ci, co = shape
    assert hip.unet_conv3x3_wgrad_bf16_supported(n, h, w, ci, co) == 1
    rng = np.random.default_rng(ci + co + h)
    x = rng.standard_normal((n, ci, h, w)).astype(np.float32)
    dz = rng.standard_normal((n, co, h, w)).astype(np.float32)
    _, dw_ref, _ = on.conv_same_bwd(bf16_round(x), np.zeros((3, 3, ci, co)), bf16_round(dz))
    xd = torch.zeros(n, h, w, ci + 4, device=DEV); xd[..., :ci] = to_nhwc(x); xv = xd[..., :ci]
    dzd = torch.zeros(n, h, w, co + 8, device=DEV); dzd[..., :co] = to_nhwc(dz); dzv = dzd[..., :co]
    nb = hip.unet_conv3x3_wgrad_bf16_workspace(n, h, w, ci, co)
    ws = ws_bytes(nb)
    dw = torch.full((3, 3, ci, co), float("nan"), device=DEV)
    hip.unet_conv3x3_wgrad_bf16(P(xv), ci + 4, 0, P(dzv), co + 8, 0, P(dw), n, h, w, ci, co, P(ws), nb, ST())
    assert relerr(dw.cpu().numpy().astype(np.float64), dw_ref) < 2e-5
    dw2 = torch.empty_like(dw)
    hip.unet_conv3x3_wgrad_bf16(P(xv), ci + 4, 0, P(dzv), co + 8, 0, P(dw2), n, h, w, ci, co, P(ws), nb, ST())
    assert torch.equal(dw, dw2)


@pytest.mark.parametrize("shape", [(2, 20, 40, 64, 128), (1, 32, 64, 128, 64), (2, 16, 32, 64, 256)])
def test_conv3x3_bf16_fused_batchnorm_sums(hip, shape):
    # the statistics variants of the bf16 kernels: same tensors as the plain kernels, bit for bit, plus per-tile partial sums whose
    # totals are the BatchNorm forward sums (sum y, sum y^2) / backward sums (sum dx, sum dx * r) of what the kernel wrote
    n, h, w, ci, co = shape
    g = torch.Generator(device=DEV); g.manual_seed(ci + co + h)
    x = torch.randn(n, h, w, ci, device=DEV, generator=g)
    wt = torch.randn(3, 3, ci, co, device=DEV, generator=g) / (3 * ci ** 0.5)
    b = torch.randn(co, device=DEV, generator=g)
    dz = torch.randn(n, h, w, co, device=DEV, generator=g)
    nb = hip.unet_conv3x3_bf16_packed_bytes(ci, co)
    wp, wpd = ws_bytes(nb), ws_bytes(nb)
    hip.unet_conv3x3_bf16_pack_weights(P(wt), P(wp), ci, co, 0, ST())
    hip.unet_conv3x3_bf16_pack_weights(P(wt), P(wpd), ci, co, 1, ST())
    y0 = torch.empty(n, h, w, co, device=DEV); y1 = torch.empty_like(y0)
    hip.unet_conv3x3_fwd_bf16(P(x), ci, 0, None, None, P(wp), P(b), P(y0), co, 0, n, h, w, ci, co, 1, None, 0, ST())
    rows = hip.unet_conv3x3_bf16_stats_rows(n, h, w, ci, co)
    assert rows == n * ((h + 15) // 16) * ((w + 31) // 32)
    part = torch.full(((co // 64) * rows * 128,), float("nan"), device=DEV)
    hip.unet_conv3x3_fwd_bf16(P(x), ci, 0, None, None, P(wp), P(b), P(y1), co, 0, n, h, w, ci, co, 1, P(part), part.numel() * 4, ST())
    assert torch.equal(y0, y1)
    pv = part.view(co // 64, rows, 64, 2).double().sum(1)                        # [C/64][64][2]
    s1 = y0.double().sum((0, 1, 2)).view(co // 64, 64); s2 = (y0.double() ** 2).sum((0, 1, 2)).view(co // 64, 64)
    assert (pv[..., 0] - s1).abs().max().item() < 1e-4 * s1.abs().max().item() + 1e-3
    assert (pv[..., 1] - s2).abs().max().item() < 1e-4 * s2.abs().max().item()
    # data gradient: dx channels [c0, c1) are the dy of a producer whose saved activation r has c1 - c0 channels
    c0, c1 = (ci // 2, ci) if ci >= 128 else (0, ci)
    r_prev = torch.randn(n, h, w, c1 - c0, device=DEV, generator=g)
    dx0 = torch.empty(n, h, w, ci, device=DEV); dx1 = torch.empty_like(dx0)
    hip.unet_conv3x3_dgrad_bf16(P(dz), co, 0, P(wpd), P(dx0), ci, 0, n, h, w, ci, co, None, 0, 0, 0, 0, None, 0, ST())
    rows2 = hip.unet_conv3x3_bf16_stats_rows(n, h, w, co, ci)
    part2 = torch.full(((ci // 64) * rows2 * 128,), float("nan"), device=DEV)
    hip.unet_conv3x3_dgrad_bf16(P(dz), co, 0, P(wpd), P(dx1), ci, 0, n, h, w, ci, co, P(r_prev), c1 - c0, 0, c0, c1,
                                P(part2), part2.numel() * 4, ST())
    assert torch.equal(dx0, dx1)
    pv2 = part2.view(ci // 64, rows2, 64, 2).double().sum(1).view(ci, 2)[c0:c1]
    t1 = dx0[..., c0:c1].double().sum((0, 1, 2)); t2 = (dx0[..., c0:c1].double() * r_prev.double()).sum((0, 1, 2))
    assert (pv2[:, 0] - t1).abs().max().item() < 1e-4 * t1.abs().max().item() + 1e-3
    assert (pv2[:, 1] - t2).abs().max().item() < 1e-4 * t2.abs().max().item() + 1e-3


def test_bf16_stored_operands_bit_identical(hip):
    # an operand stored as bf16 must give exactly what the fp32 tensor gives (the kernels round fp32 inputs to the same bf16 values)
    n, h, w, ci, co = 2, 20, 40, 128, 64
    g = torch.Generator(device=DEV); g.manual_seed(3)
    x = torch.randn(n, h, w, ci, device=DEV, generator=g); dz = torch.randn(n, h, w, co, device=DEV, generator=g)
    wt = torch.randn(3, 3, ci, co, device=DEV, generator=g) / (3 * ci ** 0.5); b = torch.randn(co, device=DEV, generator=g)
    x16, dz16 = x.to(torch.bfloat16), dz.to(torch.bfloat16)                      # torch rounds to nearest even too
    nb = hip.unet_conv3x3_bf16_packed_bytes(ci, co)
    wp, wpd = ws_bytes(nb), ws_bytes(nb)
    hip.unet_conv3x3_bf16_pack_weights(P(wt), P(wp), ci, co, 0, ST())
    hip.unet_conv3x3_bf16_pack_weights(P(wt), P(wpd), ci, co, 1, ST())
    y0 = torch.empty(n, h, w, co, device=DEV); y1 = torch.empty_like(y0)
    hip.unet_conv3x3_fwd_bf16(P(x), ci, 0, None, None, P(wp), P(b), P(y0), co, 0, n, h, w, ci, co, 1, None, 0, ST())
    hip.unet_conv3x3_fwd_bf16(P(x16), ci, 1, None, None, P(wp), P(b), P(y1), co, 0, n, h, w, ci, co, 1, None, 0, ST())
    assert torch.equal(y0, y1)
    d0 = torch.empty(n, h, w, ci, device=DEV); d1 = torch.empty_like(d0)
    hip.unet_conv3x3_dgrad_bf16(P(dz), co, 0, P(wpd), P(d0), ci, 0, n, h, w, ci, co, None, 0, 0, 0, 0, None, 0, ST())
    hip.unet_conv3x3_dgrad_bf16(P(dz16), co, 1, P(wpd), P(d1), ci, 0, n, h, w, ci, co, None, 0, 0, 0, 0, None, 0, ST())
    assert torch.equal(d0, d1)
    nbw = hip.unet_conv3x3_wgrad_bf16_workspace(n, h, w, ci, co)
    ws = ws_bytes(nbw)
    outs = []
    for xa, xf, za, zf in ((x, 0, dz, 0), (x16, 1, dz16, 1), (x, 0, dz16, 1), (x16, 1, dz, 0)):
        dw = torch.empty(3, 3, ci, co, device=DEV)
        hip.unet_conv3x3_wgrad_bf16(P(xa), ci, xf, P(za), co, zf, P(dw), n, h, w, ci, co, P(ws), nbw, ST())
        outs.append(dw)
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    # producers: BatchNorm apply / backward storing bf16 == rounding their fp32 result
    r = torch.randn(n, h, w, co, device=DEV, generator=g)
    sc = torch.rand(co, device=DEV, generator=g) + 0.5; sh = torch.randn(co, device=DEV, generator=g)
    yf = torch.empty(n, h, w, co, device=DEV); yh = torch.empty(n, h, w, co, device=DEV, dtype=torch.bfloat16)
    hip.unet_bn_apply(P(r), co, P(sc), P(sh), P(yf), co, n * h * w, co, ST())
    hip.unet_bn_apply_any(P(r), co, 0, P(sc), P(sh), P(yh), co, 1, None, 0, None, n, h, w, co, ST())
    assert torch.equal(yf.to(torch.bfloat16), yh)
    gm = torch.rand(co, device=DEV, generator=g) + 0.5
    mean = r.mean((0, 1, 2)); invstd = 1.0 / torch.sqrt(r.var((0, 1, 2), unbiased=False) + 1e-3)
    dy = torch.randn(n, h, w, co, device=DEV, generator=g)
    nbb = hip.unet_bn_workspace(n * h * w, co)
    wsb = ws_bytes(nbb)
    zf_ = torch.empty(n, h, w, co, device=DEV); zh = torch.empty(n, h, w, co, device=DEV, dtype=torch.bfloat16)
    gr = [torch.empty(co, device=DEV) for _ in range(6)]
    hip.unet_bn_bwd(P(dy), co, P(r), co, P(gm), P(mean), P(invstd), n * h * w, co, 1, P(zf_), co, P(gr[0]), P(gr[1]), P(gr[2]), P(wsb), nbb, ST())
    hip.unet_bn_bwd_any(P(dy), co, None, 0, None, n, h, w, P(r), co, P(gm), P(mean), P(invstd), co, 1, P(zh), co, 1,
                        P(gr[3]), P(gr[4]), P(gr[5]), None, 0, P(wsb), nbb, ST(), 0, 0, 0, None)
    assert torch.equal(zf_.to(torch.bfloat16), zh)
    assert all(torch.equal(gr[i], gr[i + 3]) for i in range(3))


@pytest.mark.parametrize("shape", [(1, 16, 32, 64, 64), (2, 10, 20, 128, 64), (1, 8, 40, 64, 128), (2, 4, 8, 256, 128)])
def test_convT2x2_bf16_fwd_dgrad_match_oracle_on_rounded_operands(hip, shape):
    # bf16 transposed conv (forward, data gradient): fp64 oracle on the bf16-rounded operands to fp32-accumulation accuracy; ragged
    # tiles, padded leading dimensions, both column-tile widths, the fused BatchNorm sums, a bf16-stored input
    n, h, w, ci, co = shape
    assert hip.unet_convT2x2_bf16_supported(n, h, w, ci, co) == 1
    rng = np.random.default_rng(ci + co + h)
    x = rng.standard_normal((n, ci, h, w)).astype(np.float32)
    wt = (rng.standard_normal((2, 2, co, ci)) / np.sqrt(ci)).astype(np.float32)
    b = rng.standard_normal(co).astype(np.float32)
    dz = rng.standard_normal((n, co, 2 * h, 2 * w)).astype(np.float32)
    z_ref = on.deconv2x2_fwd(bf16_round(x), bf16_round(wt), b.astype(np.float64))
    dx_ref, _, _ = on.deconv2x2_bwd(bf16_round(x), bf16_round(wt), bf16_round(dz))
    xd = torch.zeros(n, h, w, ci + 8, device=DEV); xd[..., :ci] = to_nhwc(x); xv = xd[..., :ci]
    dzd, wd, bd = to_nhwc(dz), dev(wt), dev(b)
    nb = hip.unet_convT2x2_bf16_packed_bytes(ci, co)
    wp, wpd = ws_bytes(nb), ws_bytes(nb)
    hip.unet_convT2x2_bf16_pack_weights(P(wd), P(wp), ci, co, 0, ST())
    hip.unet_convT2x2_bf16_pack_weights(P(wd), P(wpd), ci, co, 1, ST())
    cat = torch.full((n, 2 * h, 2 * w, 2 * co), float("nan"), device=DEV)
    outv = cat[..., co:]                                               # the upper half of a concat buffer
    rows = hip.unet_convT2x2_bf16_stats_rows(n, h, w, ci, co, 0)
    part = torch.full(((co // 64) * rows * 128,), float("nan"), device=DEV)
    hip.unet_convT2x2_fwd_bf16(P(xv), ci + 8, 0, P(wp), P(bd), P(outv), 2 * co, 0, n, h, w, ci, co, P(part), part.numel() * 4, ST())
    z = from_nhwc(outv)
    assert relerr(z, z_ref) < 2e-5
    assert torch.isnan(cat[..., :co]).all()
    pv = part.view(co // 64, rows, 64, 2).double().sum(1).view(co, 2)
    s1 = outv.double().sum((0, 1, 2)); s2 = (outv.double() ** 2).sum((0, 1, 2))
    assert (pv[:, 0] - s1).abs().max().item() < 1e-4 * s1.abs().max().item() + 1e-3
    assert (pv[:, 1] - s2).abs().max().item() < 1e-4 * s2.abs().max().item()
    out2 = torch.empty(n, 2 * h, 2 * w, co, device=DEV)
    hip.unet_convT2x2_fwd_bf16(P(xv.contiguous().to(torch.bfloat16)), ci, 1, P(wp), P(bd), P(out2), co, 0, n, h, w, ci, co, None, 0, ST())
    assert torch.equal(out2, outv.contiguous())
    # bf16-stored output (into the upper half of a bf16 concat buffer) == the fp32 output rounded; the fused sums are those of the fp32 values
    cat16 = torch.full((n, 2 * h, 2 * w, 2 * co), float("nan"), device=DEV, dtype=torch.bfloat16)
    part16 = torch.full_like(part, float("nan"))
    hip.unet_convT2x2_fwd_bf16(P(xv), ci + 8, 0, P(wp), P(bd), P(cat16[..., co:]), 2 * co, 1, n, h, w, ci, co, P(part16), part16.numel() * 4, ST())
    assert torch.equal(cat16[..., co:], outv.to(torch.bfloat16)) and torch.isnan(cat16[..., :co].float()).all()
    assert torch.equal(part16, part)
    # data gradient (+ the producer's BatchNorm-backward sums)
    dx = torch.empty(n, h, w, ci, device=DEV); dx2 = torch.empty_like(dx)
    hip.unet_convT2x2_dgrad_bf16(P(dzd), co, 0, P(wpd), P(dx), ci, 0, n, h, w, ci, co, None, 0, 0, None, 0, ST())
    assert relerr(from_nhwc(dx), dx_ref) < 2e-5
    g = torch.Generator(device=DEV); g.manual_seed(1)
    r_prev = torch.randn(n, h, w, ci, device=DEV, generator=g)
    rows2 = hip.unet_convT2x2_bf16_stats_rows(n, h, w, ci, co, 1)
    part2 = torch.full(((ci // 64) * rows2 * 128,), float("nan"), device=DEV)
    hip.unet_convT2x2_dgrad_bf16(P(dzd.to(torch.bfloat16)), co, 1, P(wpd), P(dx2), ci, 0, n, h, w, ci, co, P(r_prev), ci, 0, P(part2), part2.numel() * 4, ST())
    assert torch.equal(dx, dx2)
    pv2 = part2.view(ci // 64, rows2, 64, 2).double().sum(1).view(ci, 2)
    t1 = dx.double().sum((0, 1, 2)); t2 = (dx.double() * r_prev.double()).sum((0, 1, 2))
    assert (pv2[:, 0] - t1).abs().max().item() < 1e-4 * t1.abs().max().item() + 1e-3
    assert (pv2[:, 1] - t2).abs().max().item() < 1e-4 * t2.abs().max().item() + 1e-3
    # bf16-stored dx and bf16-stored saved activation: dx == the fp32 result rounded, sums == those against the bf16 values of r_prev
    dx16 = torch.empty(n, h, w, ci, device=DEV, dtype=torch.bfloat16); r16 = r_prev.to(torch.bfloat16)
    part3 = torch.full_like(part2, float("nan")); part4 = torch.full_like(part2, float("nan")); dx3 = torch.empty_like(dx)
    hip.unet_convT2x2_dgrad_bf16(P(dzd), co, 0, P(wpd), P(dx16), ci, 1, n, h, w, ci, co, P(r16), ci, 1, P(part3), part3.numel() * 4, ST())
    hip.unet_convT2x2_dgrad_bf16(P(dzd), co, 0, P(wpd), P(dx3), ci, 0, n, h, w, ci, co, P(r16.float()), ci, 0, P(part4), part4.numel() * 4, ST())
    assert torch.equal(dx16, dx.to(torch.bfloat16)) and torch.equal(part3, part4)


@pytest.mark.parametrize("shape", [(1, 4, 32, 128, 64), (2, 6, 40, 128, 128), (1, 8, 16, 256, 128), (3, 5, 70, 128, 64), (2, 4, 8, 1024, 512)])
def test_convT2x2_bf16_wgrad_matches_oracle_on_rounded_operands(hip, shape):
    # bf16 transposed-conv weight gradient (stride-2 sub-lattice gathers through the transposing LDS read): fp64 oracle on the
    # bf16-rounded operands; ragged / narrow widths, padded leading dimensions, both channel-tile widths, several units per
    # workgroup (last shape: 32 channel tiles), bf16-stored operands bit-identical, reruns bit-identical
    n, h, w, ci, co = shape
    assert hip.unet_convT2x2_wgrad_bf16_supported(n, h, w, ci, co) == 1
    rng = np.random.default_rng(ci + co + h)
    x = rng.standard_normal((n, ci, h, w)).astype(np.float32)
    dz = rng.standard_normal((n, co, 2 * h, 2 * w)).astype(np.float32)
    _, dw_ref, _ = on.deconv2x2_bwd(bf16_round(x), np.zeros((2, 2, co, ci)), bf16_round(dz))
    xd = torch.zeros(n, h, w, ci + 8, device=DEV); xd[..., :ci] = to_nhwc(x); xv = xd[..., :ci]
    dzd = torch.zeros(n, 2 * h, 2 * w, co + 8, device=DEV); dzd[..., :co] = to_nhwc(dz); dzv = dzd[..., :co]
    nb = hip.unet_convT2x2_wgrad_bf16_workspace(n, h, w, ci, co)
    ws = ws_bytes(nb)
    outs = []
    for xa, lx, xf, za, lz, zf in ((xv, ci + 8, 0, dzv, co + 8, 0), (xv, ci + 8, 0, dzv, co + 8, 0),
                                   (xv.contiguous().to(torch.bfloat16), ci, 1, dzv.contiguous().to(torch.bfloat16), co, 1)):
        dw = torch.full((2, 2, co, ci), float("nan"), device=DEV)
        hip.unet_convT2x2_wgrad_bf16(P(xa), lx, xf, P(za), lz, zf, P(dw), n, h, w, ci, co, P(ws), nb, ST())
        outs.append(dw)
    assert relerr(outs[0].cpu().numpy().astype(np.float64), dw_ref) < 2e-5
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("shape", [(8, 32, 32, 1024, 512), (8, 64, 64, 512, 256), (8, 128, 128, 256, 128), (8, 256, 256, 128, 64), (3, 100, 164, 128, 64)])
def test_convT2x2_bf16_kernels_at_full_size_dma_staging_equals_register_staging(hip, shape):
    # The four up-sampling layers of BASELINE config 4 at their real sizes (the oracle is too slow there): a bf16-stored operand takes the
    # LDS-DMA staged kernels, the same values stored as fp32 the register-staged ones -- forward (+ BatchNorm sums), data gradient (+ the
    # producer's backward sums) and weight gradient must be BIT-IDENTICAL between the two, with bf16-stored outputs as the step uses them.
    # (Multi-tile persistent walks, thousands of partial rows, the 128- and 64-column kernels, one ragged shape.)
    n, h, w, ci, co = shape
    g = torch.Generator(device=DEV).manual_seed(ci + h)
    bf = torch.bfloat16
    x16 = torch.randn(n, h, w, ci, device=DEV, generator=g).to(bf); x32 = x16.float()
    dz16 = (torch.randn(n, 2 * h, 2 * w, co, device=DEV, generator=g) * 0.1).to(bf); dz32 = dz16.float()
    r16 = torch.relu(torch.randn(n, h, w, ci, device=DEV, generator=g)).to(bf)
    wt = torch.randn(2, 2, co, ci, device=DEV, generator=g) / float(np.sqrt(ci)); b = torch.randn(co, device=DEV, generator=g)
    nb = hip.unet_convT2x2_bf16_packed_bytes(ci, co); wp, wpd = ws_bytes(nb), ws_bytes(nb)
    hip.unet_convT2x2_bf16_pack_weights(P(wt), P(wp), ci, co, 0, ST()); hip.unet_convT2x2_bf16_pack_weights(P(wt), P(wpd), ci, co, 1, ST())
    rows = hip.unet_convT2x2_bf16_stats_rows(n, h, w, ci, co, 0); rows2 = hip.unet_convT2x2_bf16_stats_rows(n, h, w, ci, co, 1)
    outs = []
    for xin, xf, dzin, zf in ((x16, 1, dz16, 1), (x32, 0, dz32, 0)):
        z = torch.empty(n, 2 * h, 2 * w, co, device=DEV, dtype=bf); part = torch.zeros((co // 64) * rows * 128, device=DEV)
        hip.unet_convT2x2_fwd_bf16(P(xin), ci, xf, P(wp), P(b), P(z), co, 1, n, h, w, ci, co, P(part), part.numel() * 4, ST())
        dx = torch.empty(n, h, w, ci, device=DEV, dtype=bf); part2 = torch.zeros((ci // 64) * rows2 * 128, device=DEV)
        hip.unet_convT2x2_dgrad_bf16(P(dzin), co, zf, P(wpd), P(dx), ci, 1, n, h, w, ci, co, P(r16), ci, 1, P(part2), part2.numel() * 4, ST())
        dxp = torch.empty(n, h, w, ci, device=DEV, dtype=bf)
        hip.unet_convT2x2_dgrad_bf16(P(dzin), co, zf, P(wpd), P(dxp), ci, 1, n, h, w, ci, co, None, 0, 0, None, 0, ST())
        nbw = hip.unet_convT2x2_wgrad_bf16_workspace(n, h, w, ci, co); wsw = ws_bytes(nbw); dw = torch.zeros(2, 2, co, ci, device=DEV)
        hip.unet_convT2x2_wgrad_bf16(P(xin), ci, xf, P(dzin), co, zf, P(dw), n, h, w, ci, co, P(wsw), nbw, ST())
        torch.cuda.synchronize()
        outs.append((z, part, dx, part2, dxp, dw))
    for a, b_ in zip(*outs):
        assert torch.equal(a, b_)
    z, part, dx, part2, dxp, dw = outs[0]
    assert torch.equal(dx, dxp) and torch.isfinite(dw).all() and dw.abs().max().item() > 0
    # ... and against an independent implementation at this size: torch's own transposed convolution, its input gradient (a stride-2
    # convolution) and its weight gradient, on the same bf16-valued operands
    wq = wt.to(bf).float().permute(3, 2, 0, 1).contiguous()                              # [ci][co][a][b] = W[a][b][co][ci]
    F = torch.nn.functional
    zref = F.conv_transpose2d(x32.permute(0, 3, 1, 2), wq, b, stride=2).permute(0, 2, 3, 1)
    assert bool(((z.float() - zref).abs() <= 2.0 ** -7 * zref.abs() + 2e-5 * zref.abs().max()).all())
    pv = part.view(co // 64, rows, 64, 2).double().sum(1).view(co, 2)
    s1, s2 = zref.double().sum((0, 1, 2)), (zref.double() ** 2).sum((0, 1, 2))
    assert (pv[:, 0] - s1).abs().max().item() < 1e-4 * zref.double().abs().sum((0, 1, 2)).max().item() and (pv[:, 1] - s2).abs().max().item() < 1e-4 * s2.abs().max().item()
    dxref = F.conv2d(dz32.permute(0, 3, 1, 2), wq, stride=2).permute(0, 2, 3, 1)
    assert bool(((dx.float() - dxref).abs() <= 2.0 ** -7 * dxref.abs() + 2e-5 * dxref.abs().max()).all())
    pv2 = part2.view(ci // 64, rows2, 64, 2).double().sum(1).view(ci, 2)
    t1, t2 = dxref.double().sum((0, 1, 2)), (dxref.double() * r16.double()).sum((0, 1, 2))
    tol = 1e-4 * dxref.double().abs().sum((0, 1, 2)).max().item()
    assert (pv2[:, 0] - t1).abs().max().item() < tol and (pv2[:, 1] - t2).abs().max().item() < tol
    dwref = torch.nn.grad.conv2d_weight(dz32.permute(0, 3, 1, 2).contiguous(), (ci, co, 2, 2), x32.permute(0, 3, 1, 2).contiguous(), stride=2).permute(2, 3, 1, 0)
    assert (dw - dwref).abs().max().item() < 1e-4 * dwref.abs().max().item()


def test_bf16_pack_weights_batch_matches_single_packs(hip):
    # one launch for every layer's operands == the per-layer pack kernels, bit for bit (3x3 and transposed-conv jobs mixed)
    g = torch.Generator(device=DEV); g.manual_seed(9)
    layers = [(0, 64, 128), (1, 256, 128), (0, 128, 64), (1, 128, 64)]           # (kind, Cin, Cout)
    rows, blk, keep = [], 0, []
    for kind, ci, co in layers:
        taps = 4 if kind else 9
        w = torch.randn((2, 2, co, ci) if kind else (3, 3, ci, co), device=DEV, generator=g)
        nb = taps * ci * co * 2
        a, b = ws_bytes(nb), ws_bytes(nb); a0, b0 = ws_bytes(nb), ws_bytes(nb)
        pack = hip.unet_convT2x2_bf16_pack_weights if kind else hip.unet_conv3x3_bf16_pack_weights
        pack(P(w), P(a0), ci, co, 0, ST()); pack(P(w), P(b0), ci, co, 1, ST())
        rows.append([w.data_ptr(), a.data_ptr(), b.data_ptr(), ci | (co << 32), kind, blk])
        blk += (taps * ci * co // 8 + 255) // 256
        keep.append((w, a, b, a0, b0, nb))
    jobs = torch.tensor(rows, dtype=torch.int64, device=DEV)
    hip.unet_bf16_pack_weights_batch(P(jobs), len(rows), blk, ST())
    for w, a, b, a0, b0, nb in keep:
        assert torch.equal(a[:nb], a0[:nb]) and torch.equal(b[:nb], b0[:nb])


@pytest.mark.parametrize("shape", [(2, 20, 40, 128, 64), (1, 16, 32, 64, 128)])
def test_conv3x3_bf16_output_storage_is_the_rounded_fp32_output(hip, shape):
    # activation-storage mode: the forward / data-gradient kernels can store their OUTPUT as bf16 and read the producer's saved
    # activation (for the fused BatchNorm-backward sums) as bf16.  The stored tensor must be exactly the rounded fp32 output, and
    # the sums -- taken before the rounding -- must be bit-equal to those of the fp32-output call.
    n, h, w, ci, co = shape
    g = torch.Generator(device=DEV); g.manual_seed(co + h)
    x = torch.randn(n, h, w, ci, device=DEV, generator=g); dz = torch.randn(n, h, w, co, device=DEV, generator=g)
    wt = torch.randn(3, 3, ci, co, device=DEV, generator=g) / (3 * ci ** 0.5); b = torch.randn(co, device=DEV, generator=g)
    nb = hip.unet_conv3x3_bf16_packed_bytes(ci, co)
    wp, wpd = ws_bytes(nb), ws_bytes(nb)
    hip.unet_conv3x3_bf16_pack_weights(P(wt), P(wp), ci, co, 0, ST())
    hip.unet_conv3x3_bf16_pack_weights(P(wt), P(wpd), ci, co, 1, ST())
    rows = hip.unet_conv3x3_bf16_stats_rows(n, h, w, ci, co)
    p0 = torch.empty((co // 64) * rows * 128, device=DEV); p1 = torch.empty_like(p0)
    y32 = torch.empty(n, h, w, co, device=DEV); y16 = torch.full((n, h, w, co + 8), float("nan"), device=DEV, dtype=torch.bfloat16)
    hip.unet_conv3x3_fwd_bf16(P(x), ci, 0, None, None, P(wp), P(b), P(y32), co, 0, n, h, w, ci, co, 1, P(p0), p0.numel() * 4, ST())
    hip.unet_conv3x3_fwd_bf16(P(x), ci, 0, None, None, P(wp), P(b), P(y16), co + 8, 1, n, h, w, ci, co, 1, P(p1), p1.numel() * 4, ST())
    assert torch.equal(y32.to(torch.bfloat16), y16[..., :co]) and torch.isnan(y16[..., co:].float()).all()
    assert torch.equal(p0, p1)
    r32 = torch.randn(n, h, w, ci, device=DEV, generator=g).to(torch.bfloat16).float()      # bf16-representable values
    r16 = r32.to(torch.bfloat16)
    rows2 = hip.unet_conv3x3_bf16_stats_rows(n, h, w, co, ci)
    q0 = torch.empty((ci // 64) * rows2 * 128, device=DEV); q1 = torch.empty_like(q0)
    d32 = torch.empty(n, h, w, ci, device=DEV); d16 = torch.empty(n, h, w, ci, device=DEV, dtype=torch.bfloat16)
    hip.unet_conv3x3_dgrad_bf16(P(dz), co, 0, P(wpd), P(d32), ci, 0, n, h, w, ci, co, P(r32), ci, 0, 0, ci, P(q0), q0.numel() * 4, ST())
    hip.unet_conv3x3_dgrad_bf16(P(dz), co, 0, P(wpd), P(d16), ci, 1, n, h, w, ci, co, P(r16), ci, 1, 0, ci, P(q1), q1.numel() * 4, ST())
    assert torch.equal(d32.to(torch.bfloat16), d16)
    assert torch.equal(q0, q1)


@pytest.mark.parametrize("shape", [(2, 6, 10, 64, 64), (1, 8, 12, 128, 192), (2, 4, 6, 1024, 1024), (1, 6, 8, 256, 264)])
@pytest.mark.parametrize("mix", ["all16", "r16", "dy16", "out16"])
def test_batchnorm_kernels_on_bf16_stored_tensors(hip, shape, mix):
    # The BatchNorm passes with any of their tensors stored as bf16 (8 channels = 16 bytes per lane): results must equal the fp32
    # kernels run on the same (bf16-representable) values -- elementwise outputs bit for bit after the same rounding, the per-channel
    # sums to fp64-summation-order noise.  Shapes include padded leading dimensions (ld > C).
    n, h, w, c, ld = shape
    g = torch.Generator(device=DEV).manual_seed(c + h)
    r16f = mix in ("all16", "r16"); dy16f = mix in ("all16", "dy16"); o16f = mix in ("all16", "out16")
    def mk(scale=1.0):
        t = (torch.randn(n, h, w, ld, device=DEV, generator=g) * scale).to(torch.bfloat16)
        return t
    r16 = mk().clamp_min(0) ; dy16 = mk(0.1); pdy16 = (torch.randn(n, h // 2, w // 2, ld, device=DEV, generator=g) * 0.1).to(torch.bfloat16)
    r32, dy32, pdy32 = r16.float(), dy16.float(), pdy16.float()
    sc = torch.rand(c, device=DEV, generator=g) + 0.5; sh = torch.randn(c, device=DEV, generator=g)
    rr, dd, pp = (r16 if r16f else r32), (dy16 if dy16f else dy32), (pdy16 if dy16f else pdy32)
    odt = torch.bfloat16 if o16f else torch.float32
    # ---- apply (+ pool)
    y_ref = torch.empty(n, h, w, ld, device=DEV); y = torch.zeros(n, h, w, ld, device=DEV, dtype=odt)
    hip.unet_bn_apply(P(r32), ld, P(sc), P(sh), P(y_ref), ld, n * h * w, c, ST())
    hip.unet_bn_apply_any(P(rr), ld, int(r16f), P(sc), P(sh), P(y), ld, int(o16f), None, 0, None, n, h, w, c, ST())
    assert torch.equal(y_ref[..., :c].to(odt), y[..., :c])
    pl_ref = torch.empty(n, h // 2, w // 2, ld, device=DEV); ix_ref = torch.empty(n, h // 2, w // 2, c, device=DEV, dtype=torch.uint8)
    pl = torch.zeros(n, h // 2, w // 2, ld, device=DEV, dtype=odt); ix = torch.empty_like(ix_ref)
    hip.unet_bn_apply_maxpool(P(r32), ld, P(sc), P(sh), P(y_ref), ld, P(pl_ref), ld, P(ix_ref), n, h, w, c, ST())
    hip.unet_bn_apply_any(P(rr), ld, int(r16f), P(sc), P(sh), P(y), ld, int(o16f), P(pl), ld, P(ix), n, h, w, c, ST())
    assert torch.equal(y_ref[..., :c].to(odt), y[..., :c]) and torch.equal(pl_ref[..., :c].to(odt), pl[..., :c]) and torch.equal(ix_ref, ix)
    # ---- backward: plain, pooled, from partial sums
    gm = torch.rand(c, device=DEV, generator=g) + 0.5
    rv = r32[..., :c]
    mean = rv.mean((0, 1, 2)).contiguous(); invstd = (1.0 / torch.sqrt(rv.var((0, 1, 2), unbiased=False) + 1e-3)).contiguous()
    nbb = hip.unet_bn_workspace(n * h * w, c); wsb = ws_bytes(nbb)
    for pooled in (False, True):
        z_ref = torch.empty(n, h, w, ld, device=DEV); z = torch.zeros(n, h, w, ld, device=DEV, dtype=odt)
        gr = [torch.empty(c, device=DEV) for _ in range(6)]
        if pooled:
            hip.unet_bn_bwd_pooled(P(dy32), ld, P(pdy32), ld, P(ix_ref), n, h, w, P(r32), ld, P(gm), P(mean), P(invstd), c, 1, P(z_ref), ld,
                                   P(gr[0]), P(gr[1]), P(gr[2]), P(wsb), nbb, ST())
        else:
            hip.unet_bn_bwd(P(dy32), ld, P(r32), ld, P(gm), P(mean), P(invstd), n * h * w, c, 1, P(z_ref), ld, P(gr[0]), P(gr[1]), P(gr[2]), P(wsb), nbb, ST())
        hip.unet_bn_bwd_any(P(dd), ld, P(pp) if pooled else None, ld if pooled else 0, P(ix_ref) if pooled else None, n, h, w, P(rr), ld,
                            P(gm), P(mean), P(invstd), c, 1, P(z), ld, int(o16f), P(gr[3]), P(gr[4]), P(gr[5]), None, 0, P(wsb), nbb, ST(),
                            int(r16f), int(dy16f), int(dy16f), None)
        # the deferred form of the bias gradient: same call with host_bias_rows, finished by unet_bn_bwd_bias -> the same bits
        import ctypes as _ct
        rows_ = _ct.c_int(0); gb = torch.empty_like(gr[5]); z2 = torch.empty_like(z)
        hip.unet_bn_bwd_any(P(dd), ld, P(pp) if pooled else None, ld if pooled else 0, P(ix_ref) if pooled else None, n, h, w, P(rr), ld,
                            P(gm), P(mean), P(invstd), c, 1, P(z2), ld, int(o16f), P(gr[3]), P(gr[4]), None, None, 0, P(wsb), nbb, ST(),
                            int(r16f), int(dy16f), int(dy16f), _ct.byref(rows_))
        assert rows_.value > 0
        hip.unet_bn_bwd_bias(P(wsb), rows_.value, c, P(gb), ST())
        assert torch.equal(gb, gr[5])
        for i in range(3):
            assert (gr[i] - gr[i + 3]).abs().max().item() <= 2e-6 * gr[i].abs().max().item() + 1e-7, (pooled, i)
        zr = z_ref[..., :c]
        if o16f:     # one bf16 ulp where the sums' last bits move a value across a rounding boundary
            assert ((zr.to(torch.bfloat16).float() - z[..., :c].float()).abs() <= 2.0 ** -7 * zr.abs() + 1e-9).all()
            assert (zr.to(torch.bfloat16) != z[..., :c]).float().mean().item() < 1e-3
        else:
            assert (zr - z[..., :c]).abs().max().item() <= 2e-6 * zr.abs().max().item()


@pytest.mark.parametrize("shape", [(2, 16, 32, 64, 64), (1, 20, 40, 128, 128), (2, 10, 12, 64, 192), (1, 34, 66, 128, 64)])
@pytest.mark.parametrize("r_bf16", [0, 1])
def test_conv3x3_bf16_batchnorm_apply_on_load_is_bit_identical_to_two_passes(hip, shape, r_bf16):
    # BatchNorm-apply on load: conv(in_scale, in_shift; r) must equal conv(y) with y = unet_bn_apply_any(r) materialised as bf16,
    # BIT FOR BIT (same fma, same rounding instruction, exact zeros at the padding -- the shift must not leak into the halo), for an
    # fp32 or a bf16 conv output r, with and without the fused BatchNorm sums, on ragged tiles and padded leading dimensions
    n, h, w, ci, co = shape
    g = torch.Generator(device=DEV).manual_seed(ci + w)
    ld = ci + 8
    r32 = torch.relu(torch.randn(n, h, w, ld, device=DEV, generator=g))
    r = r32.to(torch.bfloat16) if r_bf16 else r32
    sc = torch.randn(ci, device=DEV, generator=g); sh = torch.randn(ci, device=DEV, generator=g) * 2 + 1.0      # shift far from 0
    wt = torch.randn(3, 3, ci, co, device=DEV, generator=g) * 0.05; b = torch.randn(co, device=DEV, generator=g)
    wp = torch.empty(hip.unet_conv3x3_bf16_packed_bytes(ci, co), dtype=torch.uint8, device=DEV)
    hip.unet_conv3x3_bf16_pack_weights(P(wt), P(wp), ci, co, 0, ST())
    y16 = torch.zeros(n, h, w, ld, device=DEV, dtype=torch.bfloat16)
    hip.unet_bn_apply_any(P(r), ld, r_bf16, P(sc), P(sh), P(y16), ld, 1, None, 0, None, n, h, w, ci, ST())
    rows = hip.unet_conv3x3_bf16_stats_rows(n, h, w, ci, co)
    for stats in (False, True):
        outs, parts = [], []
        for norm in (False, True):
            out = torch.zeros(n, h, w, co, device=DEV)
            part = torch.zeros((co // 64) * rows * 128, device=DEV)
            if norm:
                hip.unet_conv3x3_fwd_bf16(P(r), ld, r_bf16, P(sc), P(sh), P(wp), P(b), P(out), co, 0, n, h, w, ci, co, 1,
                                          P(part) if stats else None, part.numel() * 4 if stats else 0, ST())
            else:
                hip.unet_conv3x3_fwd_bf16(P(y16), ld, 1, None, None, P(wp), P(b), P(out), co, 0, n, h, w, ci, co, 1,
                                          P(part) if stats else None, part.numel() * 4 if stats else 0, ST())
            outs.append(out); parts.append(part)
        assert torch.equal(outs[0], outs[1]) and torch.equal(parts[0], parts[1])
    # and against the fp64 oracle on the operands the contract names: bf16(scale * r + shift), zero padded
    yref = (sc.double() * r.double()[..., :ci] + sh.double()).float().to(torch.bfloat16).double().cpu().numpy()
    wref = wt.to(torch.bfloat16).double().cpu().numpy()
    ref = on.relu_fwd(on.conv_same_fwd(yref.transpose(0, 3, 1, 2), wref, b.double().cpu().numpy())).transpose(0, 2, 3, 1)
    assert relerr(outs[1].cpu().numpy(), ref) < 2e-5


@pytest.mark.parametrize("shape", [(2, 16, 16, 64, 64), (1, 18, 34, 64, 128), (2, 32, 48, 128, 64), (1, 8, 8, 256, 256)])
def test_winograd_batchnorm_apply_on_load_matches_two_passes(hip, shape):
    # fp32 fused Winograd forward with BatchNorm-apply on load (scaled weight transform + folded bias + per-channel padding value) ==
    # the same kernel on the MATERIALISED BatchNorm output, to fp32 rounding: interior and every border / corner pixel (the shift must
    # cancel exactly where the zero padding of the BatchNorm output contributes nothing), shift chosen large so a leak would be gross;
    # and against the fp64 oracle.  Second half: the weight gradient on the raw conv output, corrected by unet_conv3x3_wgrad_fold_fix,
    # == the weight gradient on the materialised tensor.
    n, h, w, ci, co = shape
    g = torch.Generator(device=DEV).manual_seed(h * w + ci)
    ldx = ci + 4
    r = torch.relu(torch.randn(n, h, w, ldx, device=DEV, generator=g))
    sc = (torch.rand(ci, device=DEV, generator=g) + 0.5) * (torch.randint(0, 2, (ci,), device=DEV, generator=g).float() * 2 - 1)    # both signs
    sh = torch.randn(ci, device=DEV, generator=g) * 3 + 2.0
    wt = torch.randn(3, 3, ci, co, device=DEV, generator=g) * 0.05; b = torch.randn(co, device=DEV, generator=g)
    y = torch.zeros(n, h, w, ldx, device=DEV)
    hip.unet_bn_apply(P(r), ldx, P(sc), P(sh), P(y), ldx, n * h * w, ci, ST())
    Uc = torch.empty(16 * ci * co, device=DEV)
    hip.unet_winograd_weight_transform(P(wt), P(Uc), ci, co, 2, ST())
    out_ref = torch.empty(n, h, w, co, device=DEV)
    hip.unet_conv3x3_fwd_winograd_fused(P(y), ldx, None, P(Uc), P(b), P(out_ref), co, n, h, w, ci, co, 1, None, 0, ST())
    Uf = torch.empty(16 * ci * co, device=DEV); bf = torch.empty(co, device=DEV); pad = torch.empty(ci + 8, device=DEV)
    nbf = hip.unet_winograd_weight_fold_workspace(ci, co); wsf = ws_bytes(nbf)
    hip.unet_winograd_weight_fold(P(wt), P(b), P(sc), P(sh), P(Uf), P(bf), P(pad), ci, co, P(wsf), nbf, ST())
    out = torch.empty(n, h, w, co, device=DEV)
    hip.unet_conv3x3_fwd_winograd_fused(P(r), ldx, P(pad), P(Uf), P(bf), P(out), co, n, h, w, ci, co, 1, None, 0, ST())
    assert torch.allclose(pad[:ci], -sh / sc, rtol=1e-6) and (pad[ci:] == 0).all()
    yref = (sc.double() * r.double()[..., :ci] + sh.double()).cpu().numpy()
    ref = on.relu_fwd(on.conv_same_fwd(yref.transpose(0, 3, 1, 2), wt.double().cpu().numpy(), b.double().cpu().numpy())).transpose(0, 2, 3, 1)
    scale_ = np.abs(ref).max()
    for name_, o in (("two passes", out_ref), ("on load", out)):
        err = np.abs(o.cpu().numpy() - ref)
        assert err.max() < 3e-5 * scale_, (name_, err.max() / scale_)
        border = np.ones((h, w), bool); border[1:-1, 1:-1] = False
        assert err[:, border].max() < 3e-5 * scale_, (name_, "border")
    # ---- weight gradient
    dz = torch.randn(n, h, w, co, device=DEV, generator=g)
    if hip.unet_winograd_wgrad_fused_supported(n, h, w, ci, co) == 1:
        nbw = hip.unet_conv3x3_wgrad_winograd_fused_workspace(n, h, w, ci, co, 0); wsw = ws_bytes(nbw)
        dw_ref = torch.empty(3, 3, ci, co, device=DEV); dw = torch.empty(3, 3, ci, co, device=DEV)
        hip.unet_conv3x3_wgrad_winograd_fused(P(y), ldx, P(dz), co, P(dw_ref), n, h, w, ci, co, 0, P(wsw), nbw, ST())
        hip.unet_conv3x3_wgrad_winograd_fused(P(r), ldx, P(dz), co, P(dw), n, h, w, ci, co, 0, P(wsw), nbw, ST())
        total = dz.sum((0, 1, 2)).contiguous()
        nb8 = hip.unet_conv3x3_wgrad_fold_fix_workspace(co); ws8 = ws_bytes(nb8)
        hip.unet_conv3x3_wgrad_fold_fix(P(dw), P(sc), P(sh), P(dz), co, P(total), n, h, w, ci, co, P(ws8), nb8, ST())
        _, dw64, _ = on.conv_same_bwd(yref.transpose(0, 3, 1, 2), wt.double().cpu().numpy(), dz.double().cpu().numpy().transpose(0, 3, 1, 2))
        sw = np.abs(dw64).max()
        assert np.abs(dw_ref.cpu().numpy() - dw64).max() < 3e-5 * sw
        assert np.abs(dw.cpu().numpy() - dw64).max() < 3e-5 * sw


@pytest.mark.parametrize("shape", [(4, 256, 256, 64), (8, 512, 512, 64), (4, 256, 256, 128), (3, 200, 328, 64)])
@pytest.mark.parametrize("pooled", [False, True])
def test_bn_bwd_packed_bf16_kernels_equal_the_generic_kernels_at_full_size(hip, shape, pooled):
    # The mixed-precision step's BatchNorm backward on all-bf16 tensors runs packed-register kernels (bn_bwd_apply16 / pool16); the same
    # values with dy stored as fp32 take the generic kernels.  dz must be bit-identical and every sum equal to rounding -- at the sizes of
    # the training step: multi-iteration block loops, 1024-block grids (a first version of the packed kernel was right at the small
    # test shapes and stored 0.05 % of the elements wrong at these; only the 500-step training test on the reference's tiles noticed).
    n, h, w, c = shape
    g = torch.Generator(device=DEV).manual_seed(c + h)
    bf = torch.bfloat16
    r = torch.relu(torch.randn(n, h, w, c, device=DEV, generator=g)).to(bf); dy = torch.randn(n, h, w, c, device=DEV, generator=g).to(bf)
    pdy = torch.randn(n, h // 2, w // 2, c, device=DEV, generator=g).to(bf)
    idx = torch.randint(0, 4, (n, h // 2, w // 2, c), device=DEV, generator=g, dtype=torch.uint8)
    gm = torch.rand(c, device=DEV, generator=g) + 0.5; mean = torch.rand(c, device=DEV, generator=g); inv = torch.rand(c, device=DEV, generator=g) + 0.5
    nb = hip.unet_bn_workspace(n * h * w, c); ws = ws_bytes(nb)
    outs = []
    for dy16 in (1, 0):
        d = dy if dy16 else dy.float()
        pp = pdy if dy16 else pdy.float()
        dz = torch.zeros(n, h, w, c, device=DEV, dtype=bf)
        gr = [torch.empty(c, device=DEV) for _ in range(3)]
        hip.unet_bn_bwd_any(P(d), c, P(pp) if pooled else None, c if pooled else 0, P(idx) if pooled else None, n, h, w, P(r), c, P(gm), P(mean), P(inv),
                            c, 1, P(dz), c, 1, P(gr[0]), P(gr[1]), P(gr[2]), None, 0, P(ws), nb, ST(), 1, dy16, dy16, None)
        outs.append((dz, gr))
    (za, ga), (zb, gb) = outs
    assert torch.equal(za, zb)
    for i in range(3):
        assert (ga[i] - gb[i]).abs().max().item() <= 2e-6 * gb[i].abs().max().item() + 1e-6, i
    # and against the formula itself (torch, from the device's own dgamma / dbeta): at most the odd rounding that falls the other way
    dyt = dy.float()
    if pooled:
        up = torch.zeros(n, h, w, c, device=DEV)
        for pos in range(4):
            up[:, pos >> 1::2, pos & 1::2, :] = torch.where(idx == pos, pdy.float(), torch.zeros_like(pdy.float()))
        dyt = dyt + up
    a = gm * inv; c1 = ga[1] / (n * h * w); c2 = ga[0] / (n * h * w)
    ref = a * dyt + (-(a * (c2 * inv))) * r.float() + a * (c2 * inv * mean - c1)
    ref = torch.where(r.float() > 0, ref, torch.zeros_like(ref))
    assert int(((za.float() - ref).abs() > 2.0 ** -7 * ref.abs() + 1e-3).sum().item()) == 0


@pytest.mark.parametrize("shape", [(2, 16, 16, 64, 64), (1, 18, 34, 64, 128), (1, 8, 8, 256, 256)])
def test_winograd_batchnorm_apply_on_load_with_vanishing_scales(hip, shape):
    # Dead / pruned channels: BatchNorm scale s = gamma / sqrt(var + eps) of magnitude 0, 1e-38, 1e-30, 1e-6, 1e-4, 1e-2 (both signs) on
    # half of the input channels, shift = O(1): the padding value -t / s is then up to 1e30 in magnitude.  The folded forward must still
    # match the fp64 oracle evaluated with the TRUE scales (0 included) at the usual bound, borders and corners included, and so must the
    # fold-corrected weight gradient (csrc/winograd.hip, "Conditioning").
    n, h, w, ci, co = shape
    g = torch.Generator(device=DEV).manual_seed(h * w + ci + 1)
    ldx = ci + 4
    r = torch.relu(torch.randn(n, h, w, ldx, device=DEV, generator=g))
    sc = (torch.rand(ci, device=DEV, generator=g) + 0.5) * (torch.randint(0, 2, (ci,), device=DEV, generator=g).float() * 2 - 1)
    tiny = [0.0, -0.0, 1e-38, -1e-38, 1e-30, -1e-30, 1e-6, -1e-6, 1e-4, -1e-4, 1e-2, -1e-2]
    for j in range(0, ci, 2):
        sc[j] = tiny[(j // 2) % len(tiny)]
    sh = torch.randn(ci, device=DEV, generator=g) * 3 + 2.0
    wt = torch.randn(3, 3, ci, co, device=DEV, generator=g) * 0.05; b = torch.randn(co, device=DEV, generator=g)
    Uf = torch.empty(16 * ci * co, device=DEV); bf = torch.empty(co, device=DEV); pad = torch.empty(ci + 8, device=DEV)
    nbf = hip.unet_winograd_weight_fold_workspace(ci, co); wsf = ws_bytes(nbf)
    hip.unet_winograd_weight_fold(P(wt), P(b), P(sc), P(sh), P(Uf), P(bf), P(pad), ci, co, P(wsf), nbf, ST())
    out = torch.empty(n, h, w, co, device=DEV)
    hip.unet_conv3x3_fwd_winograd_fused(P(r), ldx, P(pad), P(Uf), P(bf), P(out), co, n, h, w, ci, co, 1, None, 0, ST())
    assert torch.isfinite(pad).all() and pad.abs().max().item() <= 1.0001e30 * max(1.0, sh.abs().max().item())
    assert torch.isfinite(Uf).all() and torch.isfinite(out).all()
    yref = (sc.double() * r.double()[..., :ci] + sh.double()).cpu().numpy()
    ref = on.relu_fwd(on.conv_same_fwd(yref.transpose(0, 3, 1, 2), wt.double().cpu().numpy(), b.double().cpu().numpy())).transpose(0, 2, 3, 1)
    scale_ = np.abs(ref).max()
    err = np.abs(out.cpu().numpy() - ref)
    border = np.ones((h, w), bool); border[1:-1, 1:-1] = False
    assert err.max() < 3e-5 * scale_ and err[:, border].max() < 3e-5 * scale_, (err.max() / scale_, err[:, border].max() / scale_)
    dz = torch.randn(n, h, w, co, device=DEV, generator=g)
    if hip.unet_winograd_wgrad_fused_supported(n, h, w, ci, co) == 1:
        nbw = hip.unet_conv3x3_wgrad_winograd_fused_workspace(n, h, w, ci, co, 0); wsw = ws_bytes(nbw)
        dw = torch.empty(3, 3, ci, co, device=DEV)
        hip.unet_conv3x3_wgrad_winograd_fused(P(r), ldx, P(dz), co, P(dw), n, h, w, ci, co, 0, P(wsw), nbw, ST())
        total = dz.sum((0, 1, 2)).contiguous()
        nb8 = hip.unet_conv3x3_wgrad_fold_fix_workspace(co); ws8 = ws_bytes(nb8)
        hip.unet_conv3x3_wgrad_fold_fix(P(dw), P(sc), P(sh), P(dz), co, P(total), n, h, w, ci, co, P(ws8), nb8, ST())
        _, dw64, _ = on.conv_same_bwd(yref.transpose(0, 3, 1, 2), wt.double().cpu().numpy(), dz.double().cpu().numpy().transpose(0, 3, 1, 2))
        assert np.abs(dw.cpu().numpy() - dw64).max() < 3e-5 * np.abs(dw64).max()


@pytest.mark.parametrize("shape", [(8, 512, 512, 64, 64), (8, 256, 256, 128, 128), (3, 200, 328, 64, 128), (2, 520, 300, 64, 64), (8, 512, 512, 128, 64)])
def test_conv3x3_bf16_persistent_kernels_equal_the_per_tile_kernels_at_full_size(hip, shape):
    # The persistent kernels (bf16-stored input, output and saved activation: LDS-DMA patch staging, a workgroup walks tiles t, t + grid,
    # ... and prefetches the next tile's first chunk under the epilogue) against the per-tile kernels, which the small-shape tests above
    # hold to the fp64 oracle: the same bf16 values handed over as an fp32 tensor take the per-tile path, and every output element and
    # every row of fused BatchNorm sums must be BIT-IDENTICAL.  Shapes with several tiles per workgroup (4096 / 1024 tiles on 512 / 256
    # slots), ragged right / bottom tiles (200 x 328, 520 x 300) and both tile widths -- the multi-tile walk never runs in the small tests.
    n, h, w, ci, co = shape
    g = torch.Generator(device=DEV).manual_seed(ci + h)
    bf = torch.bfloat16
    x16 = torch.randn(n, h, w, ci, device=DEV, generator=g).to(bf); x32 = x16.float()
    wt = torch.randn(3, 3, ci, co, device=DEV, generator=g) * 0.05; b = torch.randn(co, device=DEV, generator=g)
    wp = torch.empty(hip.unet_conv3x3_bf16_packed_bytes(ci, co), dtype=torch.uint8, device=DEV); wpd = torch.empty_like(wp)
    hip.unet_conv3x3_bf16_pack_weights(P(wt), P(wp), ci, co, 0, ST()); hip.unet_conv3x3_bf16_pack_weights(P(wt), P(wpd), ci, co, 1, ST())
    rows = hip.unet_conv3x3_bf16_stats_rows(n, h, w, ci, co)
    for stats in (True, False):
        outs, parts = [], []
        for xin, f16 in ((x16, 1), (x32, 0)):
            out = torch.zeros(n, h, w, co, device=DEV, dtype=bf)
            part = torch.zeros((co // 64) * rows * 128, device=DEV)
            hip.unet_conv3x3_fwd_bf16(P(xin), ci, f16, None, None, P(wp), P(b), P(out), co, 1, n, h, w, ci, co, 1,
                                      P(part) if stats else None, part.numel() * 4 if stats else 0, ST())
            outs.append(out); parts.append(part)
        torch.cuda.synchronize()
        assert torch.equal(outs[0], outs[1]) and torch.equal(parts[0], parts[1]), stats
        assert outs[0].float().abs().max().item() > 0.1
        if stats:
            # ... and against an INDEPENDENT implementation at this size (both kernel forms share one epilogue): torch's own fp32 convolution
            # of the same bf16-valued operands.  Output: equal but for roundings that fall the other way; fused sums: those of the fp32 values.
            ref = torch.relu(torch.nn.functional.conv2d(x32.permute(0, 3, 1, 2), wt.to(bf).float().permute(3, 2, 0, 1).contiguous(), b, padding=1)).permute(0, 2, 3, 1)
            got = outs[0].float()
            d = (got - ref).abs()
            assert bool((d <= 2.0 ** -7 * ref.abs() + 2e-5 * ref.abs().max()).all())
            assert (got != ref.to(bf).float()).float().mean().item() < 0.02
            pv = parts[0].view(co // 64, rows, 64, 2).double().sum(1).view(co, 2)
            s1, s2 = ref.double().sum((0, 1, 2)), (ref.double() ** 2).sum((0, 1, 2))
            assert (pv[:, 0] - s1).abs().max().item() < 1e-4 * s1.abs().max().item() and (pv[:, 1] - s2).abs().max().item() < 1e-4 * s2.abs().max().item()
            del ref, got, d
    # data gradient (+ the producer's BatchNorm-backward sums): dz as bf16 (persistent) vs fp32 (per tile), dx and r_prev bf16 in both
    dz16 = (torch.randn(n, h, w, co, device=DEV, generator=g) * 0.1).to(bf); dz32 = dz16.float()
    rp = torch.randn(n, h, w, ci, device=DEV, generator=g).to(bf)
    rows2 = hip.unet_conv3x3_bf16_stats_rows(n, h, w, co, ci)
    for stats in (True, False):
        outs, parts = [], []
        for dzin, f16 in ((dz16, 1), (dz32, 0)):
            dx = torch.zeros(n, h, w, ci, device=DEV, dtype=bf)
            part = torch.zeros((ci // 64) * rows2 * 128, device=DEV)
            hip.unet_conv3x3_dgrad_bf16(P(dzin), co, f16, P(wpd), P(dx), ci, 1, n, h, w, ci, co, P(rp) if stats else None, ci if stats else 0, 1,
                                        0, ci if stats else 0, P(part) if stats else None, part.numel() * 4 if stats else 0, ST())
            outs.append(dx); parts.append(part)
        torch.cuda.synchronize()
        assert torch.equal(outs[0], outs[1]) and torch.equal(parts[0], parts[1]), stats
        if stats:
            # independent check at this size: the input gradient of torch's convolution for the same operands; the producer's backward sums
            # sum(dx), sum(dx * r) are those of the UNROUNDED data gradient against the stored saved activation
            ref = torch.nn.functional.conv_transpose2d(dz32.permute(0, 3, 1, 2), wt.to(bf).float().permute(3, 2, 0, 1).contiguous(), padding=1).permute(0, 2, 3, 1)
            got = outs[0].float()
            d = (got - ref).abs()
            assert bool((d <= 2.0 ** -7 * ref.abs() + 2e-5 * ref.abs().max()).all())
            pv = parts[0].view(ci // 64, rows2, 64, 2).double().sum(1).view(ci, 2)
            t1, t2 = ref.double().sum((0, 1, 2)), (ref.double() * rp.double()).sum((0, 1, 2))
            tol1 = 1e-4 * ref.double().abs().sum((0, 1, 2)).max().item()
            assert (pv[:, 0] - t1).abs().max().item() < tol1 and (pv[:, 1] - t2).abs().max().item() < tol1
            del ref, got, d


@pytest.mark.parametrize("shape", [(2, 20, 40, 128, 64, 0), (1, 7, 33, 64, 128, 8), (3, 9, 20, 64, 64, 24), (8, 512, 512, 64, 64, 0), (8, 64, 64, 512, 512, 0),
                                   (2, 32, 32, 1024, 1024, 0), (5, 104, 136, 128, 64, 0), (8, 256, 256, 128, 128, 64)])
def test_conv3x3_bf16_wgrad_dma_staging_equals_register_staging(hip, shape):
    # Both operands stored as bf16 take the LDS-DMA staging (rows copied global -> LDS, zero page outside the image, 8 / 6-row rings);
    # the same values as fp32 tensors take the register staging the oracle tests above cover: dW must be BIT-IDENTICAL.  Odd heights
    # (a last step with one row), widths below / not a multiple of the 32-pixel strip, padded leading dimensions, row chunks and
    # split partial sums (full-size layers), 256 channel pairs without splits.
    n, h, w, ci, co, pad = shape
    g = torch.Generator(device=DEV).manual_seed(h * w + ci)
    x16 = torch.randn(n, h, w, ci + pad, device=DEV, generator=g).to(torch.bfloat16)
    z16 = (torch.randn(n, h, w, co + pad, device=DEV, generator=g) * 0.1).to(torch.bfloat16)
    x32, z32 = x16.float(), z16.float()
    nbw = hip.unet_conv3x3_wgrad_bf16_workspace(n, h, w, ci, co); ws = ws_bytes(nbw)
    dwa = torch.zeros(3, 3, ci, co, device=DEV); dwb = torch.zeros_like(dwa)
    hip.unet_conv3x3_wgrad_bf16(P(x16), ci + pad, 1, P(z16), co + pad, 1, P(dwa), n, h, w, ci, co, P(ws), nbw, ST())
    hip.unet_conv3x3_wgrad_bf16(P(x32), ci + pad, 0, P(z32), co + pad, 0, P(dwb), n, h, w, ci, co, P(ws), nbw, ST())
    torch.cuda.synchronize()
    assert torch.equal(dwa, dwb)
    assert dwa.abs().max().item() > 0 and torch.isfinite(dwa).all()
    if pad == 0 and n * h * w >= 8 * 32 * 32:
        # full-size layers: against an independent implementation (torch's own weight gradient of the same bf16-valued operands)
        ref = torch.nn.grad.conv2d_weight(x32.permute(0, 3, 1, 2).contiguous(), (co, ci, 3, 3), z32.permute(0, 3, 1, 2).contiguous(), padding=1).permute(2, 3, 1, 0)
        assert (dwa - ref).abs().max().item() < 1e-4 * ref.abs().max().item(), ((dwa - ref).abs().max().item(), ref.abs().max().item())


@pytest.mark.parametrize("case", [  # N, H, W, C, rows, r bf16, y bf16, pooled
    (8, 512, 512, 64, 4096, 1, 1, 0), (8, 512, 512, 64, 4096, 1, 1, 1), (8, 32, 32, 1024, 16, 1, 1, 0), (2, 64, 64, 512, 37, 0, 0, 0),
    (2, 64, 64, 128, 8, 0, 0, 1), (1, 16, 16, 192, 5, 0, 1, 0), (2, 8, 8, 1024, 3, 1, 0, 0)])
def test_bn_finalize_merged_into_the_apply_launch_is_bit_identical_to_two_launches(hip, case):
    # unet_bn_finalize_apply_any (round 6): the statistics finalize runs in the first workgroups of the apply grid, every workgroup waits on a
    # device counter.  Everything the two-launch form writes must come out bit for bit: mean, invstd, scale, shift, the moving statistics, y,
    # the pooled tensor and its winners -- on THREE consecutive calls sharing one counter word (the target only grows), with full-resolution row
    # counts (4096 partial rows), more channels than ... and fewer channels than workgroups, both storages, C = 192 (a layout the lane form of
    # the apply kernel does not cover: the entry point falls back to two launches and still advances the counter).
    n, h, w, c, rows, r16, y16, pooled = case
    g = torch.Generator(device=DEV).manual_seed(c + rows)
    bf = torch.bfloat16
    Pn = n * h * w
    counter = torch.zeros(1, dtype=torch.int32, device=DEV)
    target = 0
    mm1 = torch.randn(c, device=DEV, generator=g); mv1 = torch.rand(c, device=DEV, generator=g) + 0.5
    mm2, mv2 = mm1.clone(), mv1.clone()
    for call in range(3):
        part = torch.randn((c // 64) * rows * 128, device=DEV, generator=g).abs() * (Pn / rows)
        part.view(c // 64, rows, 64, 2)[..., 1] += part.view(c // 64, rows, 64, 2)[..., 0] ** 2 / (Pn / rows)      # sum x^2 >= (sum x)^2 / n
        gm = torch.rand(c, device=DEV, generator=g) + 0.5; bt = torch.randn(c, device=DEV, generator=g)
        r = torch.randn(n, h, w, c, device=DEV, generator=g)
        if r16:
            r = r.to(bf)
        outs = []
        for merged in (0, 1):
            mean, inv, sc, sh = (torch.zeros(c, device=DEV) for _ in range(4))
            y = torch.zeros(n, h, w, c, device=DEV, dtype=bf if y16 else torch.float32)
            pl = torch.zeros(n, h // 2, w // 2, c, device=DEV, dtype=y.dtype) if pooled else None
            ix = torch.zeros(n, h // 2, w // 2, c, device=DEV, dtype=torch.uint8) if pooled else None
            mm, mv = (mm2, mv2) if merged else (mm1, mv1)
            if merged:
                target = (target + c) & 0xFFFFFFFF
                hip.unet_bn_finalize_apply_any(P(part), rows, P(gm), P(bt), 1e-3, 0.99, 1, P(mm), P(mv), P(mean), P(inv), P(counter), target,
                                               P(r), c, r16, P(sc), P(sh), P(y), c, y16, P(pl) if pooled else None, c, P(ix) if pooled else None,
                                               n, h, w, c, ST())
            else:
                hip.unet_bn_train_finalize_partials(P(part), rows, Pn, c, P(gm), P(bt), 1e-3, 0.99, 1, P(mm), P(mv), P(mean), P(inv), P(sc), P(sh), ST())
                hip.unet_bn_apply_any(P(r), c, r16, P(sc), P(sh), P(y), c, y16, P(pl) if pooled else None, c, P(ix) if pooled else None, n, h, w, c, ST())
            outs.append((mean, inv, sc, sh, mm.clone(), mv.clone(), y, pl, ix))
        torch.cuda.synchronize()
        for a, b in zip(*outs):
            if a is not None:
                assert torch.equal(a, b), (call, case)
        assert int(counter.item()) & 0xFFFFFFFF == target
        assert outs[0][6].float().abs().max().item() > 0.1
