"""BF16x6 Winograd route (csrc/winograd_x6.hip; reference layer UNet._conv_layer, UNet/model.py:28-35): the weight operand is an EXACT
three-piece split of the fp32 transform, the fused BatchNorm statistics / BatchNorm-backward sums / BatchNorm-apply on load behave as in
the native kernels, and the forward / data-gradient pair is adjoint at BASELINE's layer shapes.  Accuracy against fp64 and against the
native route: tests/test_gpu_fp32_errors.py."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import unet_numpy as on
import fp32_error_cases as fc

pytestmark = pytest.mark.gpu
DEV = "cuda"
P, ST = fc.P, fc.ST


def ws_bytes(n):
    return torch.empty(int(n) + 256, dtype=torch.uint8, device=DEV)


def unpack_u6(u, K, N):
    """U6 bytes (MFMA A-operand order: [N/64][K/16][point row 4][point 4][piece 3][channel block 2][lane: k half 2, row 32][8 k], one 1-KB
    fragment per (.., channel block), rows in the order the kernel's epilogue wants) -> float64 [K/16][point row 4][point 4][piece 3][N][16]"""
    bits = u.view(torch.int16).to(torch.int32) << 16
    t = bits.view(torch.float32).double().reshape(N // 64, K // 16, 4, 4, 3, 2, 2, 32, 8)        # [tn][c][r][j][piece][cb][lh][row][e]
    # fragment row 8 g + 4 a + i holds channel 16 a + 4 g + i of the block (an accumulator lane then owns 16 consecutive channels)
    ch = torch.arange(32, device=t.device)
    row_of = 8 * ((ch >> 2) & 3) + 4 * (ch >> 4) + (ch & 3)
    t = t.index_select(7, row_of)                                                               # [..][channel of the block][e]
    return t.permute(1, 2, 3, 4, 0, 5, 7, 6, 8).reshape(K // 16, 4, 4, 3, N, 16)


@pytest.mark.parametrize("ci,co", [(64, 64), (128, 64), (64, 192), (256, 128), (256, 256), (512, 320)])
def test_x6_weight_operand_is_an_exact_split_of_the_fp32_transform(hip, ci, co):
    g = torch.Generator(device=DEV).manual_seed(ci + co)
    w = torch.randn(3, 3, ci, co, device=DEV, generator=g) * torch.pow(10.0, torch.randint(-3, 3, (3, 3, ci, co), device=DEV, generator=g).float())
    # forward: h + m + l == the native kernel's fp32 transform, bit for bit (same fp32 arithmetic, then an exact split)
    Uc = fc.native_weights(hip, w, 2).reshape(16, ci // 8, co, 8)                        # [xi][k/8][n][k%8]
    u6 = unpack_u6(fc.x6_weights(hip, w, 0), ci, co)
    tot = u6.sum(3)                                                                      # [k/16][r][j][n][16]
    want = Uc.double().reshape(4, 4, ci // 16, 2, co, 8).permute(2, 0, 1, 4, 3, 5).reshape(ci // 16, 4, 4, co, 16)
    assert torch.equal(tot, want)
    # the pieces are bf16 values of decreasing size, each the round-to-nearest-even bf16 of what the larger ones left (round 6; rounds 4-5 truncated):
    # |m| <= half a bf16 ulp of h = 2^-8 |h|, |l| <= 2^-8 |m| (or 2^-16 |h| where m = 0), and m takes either sign
    h, m, l = u6[:, :, :, 0], u6[:, :, :, 1], u6[:, :, :, 2]
    assert bool((m.abs() <= h.abs() * 2.0 ** -8).all()) and bool((l.abs() <= m.abs() * 2.0 ** -8 + (m == 0) * h.abs() * 2.0 ** -16).all())
    assert 0.3 < ((h * m) < 0).double().mean().item() < 0.7
    tw = want.float()
    assert torch.equal(h.float(), tw.to(torch.bfloat16).float())                       # h = rn_bf16(value), torch rounds the same way
    assert torch.equal(m.float(), (tw - h.float()).to(torch.bfloat16).float())
    # data gradient (k = Cout, n = Cin, rotated filter = the forward transform with points 0 and 3 swapped both ways): against the native
    # batch kernel, which forms it the same way
    uf = torch.empty(16 * ci * co, device=DEV); ud = torch.empty(16 * ci * co, device=DEV)
    jobs = torch.tensor([[w.data_ptr(), uf.data_ptr(), ud.data_ptr(), ci | (co << 32), 0, 0]], dtype=torch.int64, device=DEV)
    hip.unet_winograd_weight_transform_batch(P(jobs), 1, (ci * co + 2047) // 2048, ST())
    u6d = unpack_u6(fc.x6_weights(hip, w, 1), co, ci).sum(3)
    wantd = ud.double().reshape(4, 4, co // 16, 2, ci, 8).permute(2, 0, 1, 4, 3, 5).reshape(co // 16, 4, 4, ci, 16)
    assert torch.equal(u6d, wantd)
    # one launch for every layer and direction == the single transforms
    a0 = torch.empty_like(fc.x6_weights(hip, w, 0)); a1 = torch.empty_like(a0)
    jobs6 = torch.tensor([[w.data_ptr(), a0.data_ptr(), ci | (co << 32), 0, 0, 0], [w.data_ptr(), a1.data_ptr(), ci | (co << 32), (ci * co + 2047) // 2048, 1, 0]],
                         dtype=torch.int64, device=DEV)
    hip.unet_winograd_weight_transform_x6_batch(P(jobs6), 2, 2 * ((ci * co + 2047) // 2048), ST())
    assert torch.equal(a0, fc.x6_weights(hip, w, 0)) and torch.equal(a1, fc.x6_weights(hip, w, 1))


@pytest.mark.parametrize("shape", [(2, 16, 16, 64, 64), (1, 16, 32, 128, 128), (5, 104, 136, 64, 64), (2, 32, 48, 64, 192),
                                   (2, 16, 16, 256, 64), (3, 40, 24, 512, 128), (5, 104, 136, 256, 64)])
def test_x6_fused_bn_stats_match_bn_train_stats(hip, shape):
    # BatchNorm sums from the BF16x6 conv epilogue -> finalize == the separate statistics pass over the stored activation; the stored
    # activation itself is identical with and without the statistics
    n, h, w, ci, co = shape
    rows = hip.unet_conv3x3_fwd_winograd_fused_stats_rows(n, h, w, ci, co)
    if rows == 0:
        pytest.skip("persistent grid not a multiple of the n-tile count for this shape")
    x, _, wt, b = fc.inputs(shape, 11)
    g = torch.Generator(device=DEV).manual_seed(5)
    gm = torch.rand(co, device=DEV, generator=g) + 0.5; bt = torch.randn(co, device=DEV, generator=g)
    U6 = fc.x6_weights(hip, wt, 0)
    r = torch.empty(n, h, w, co, device=DEV); r2 = torch.empty_like(r)
    part = torch.zeros((co // 64) * rows * 128, device=DEV)
    hip.unet_conv3x3_fwd_winograd_x6(P(x), ci, None, P(U6), P(b), P(r), co, n, h, w, ci, co, 1, P(part), part.numel() * 4, ST())
    hip.unet_conv3x3_fwd_winograd_x6(P(x), ci, None, P(U6), P(b), P(r2), co, n, h, w, ci, co, 1, None, 0, ST())
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


@pytest.mark.parametrize("shape,crange", [((2, 16, 16, 64, 64), (0, 64)), ((1, 16, 32, 128, 64), (64, 128)), ((5, 104, 136, 64, 64), (0, 64)),
                                          ((2, 32, 32, 256, 128), (128, 256)),
                                          ((2, 16, 16, 64, 256), (0, 64)), ((1, 16, 32, 128, 512), (64, 128)), ((3, 40, 56, 64, 256), (0, 64))])
def test_x6_dgrad_bn_backward_sums_match_reduction(hip, shape, crange):
    # BF16x6 data gradient + (sum dy, sum dy * r) of the producer's BatchNorm from the epilogue -> bn_bwd_from_partials == the plain
    # data gradient followed by the full unet_bn_bwd, for a channel sub-range too
    n, h, w, ci, co = shape
    c0, c1 = crange
    cp = c1 - c0
    rows = hip.unet_conv3x3_fwd_winograd_fused_stats_rows(n, h, w, co, ci)
    assert rows > 0
    g = torch.Generator(device=DEV).manual_seed(ci + co + h)
    dzin = torch.randn(n, h, w, co, device=DEV, generator=g); wt = torch.randn(3, 3, ci, co, device=DEV, generator=g) / float(np.sqrt(9 * co))
    r_prev = torch.relu(torch.randn(n, h, w, cp, device=DEV, generator=g))
    gm = torch.rand(cp, device=DEV, generator=g) + 0.5
    mean = r_prev.mean((0, 1, 2)).contiguous(); invstd = (1.0 / torch.sqrt(r_prev.var((0, 1, 2), unbiased=False) + 1e-3)).contiguous()
    U6d = fc.x6_weights(hip, wt, 1)
    dx = torch.empty(n, h, w, ci, device=DEV); dx2 = torch.empty_like(dx)
    part = torch.zeros((ci // 64) * rows * 128, device=DEV)
    hip.unet_conv3x3_dgrad_winograd_x6(P(dzin), co, P(U6d), P(dx), ci, n, h, w, ci, co, P(r_prev), cp, c0, c1, P(part), part.numel() * 4, ST())
    hip.unet_conv3x3_dgrad_winograd_x6(P(dzin), co, P(U6d), P(dx2), ci, n, h, w, ci, co, None, 0, 0, 0, None, 0, ST())
    assert torch.equal(dx, dx2)
    npx = n * h * w
    dy = dx[..., c0:c1]
    nb = hip.unet_bn_workspace(npx, cp); ws = ws_bytes(nb)
    res = []
    for fused in (True, False):
        dz = torch.empty(n, h, w, cp, device=DEV); dg, db, dbias = [torch.empty(cp, device=DEV) for _ in range(3)]
        if fused:
            hip.unet_bn_bwd_from_partials(P(dy), ci, P(r_prev), cp, P(gm), P(mean), P(invstd), npx, cp, 1, P(dz), cp, P(dg), P(db), P(dbias),
                                          ctypes.c_void_p(part.data_ptr() + (c0 // 64) * rows * 128 * 4), rows, P(ws), nb, ST())
        else:
            hip.unet_bn_bwd(P(dy), ci, P(r_prev), cp, P(gm), P(mean), P(invstd), npx, cp, 1, P(dz), cp, P(dg), P(db), P(dbias), P(ws), nb, ST())
        res.append((dz, dg, db, dbias))
    for a_, b_ in zip(res[0], res[1]):
        scale = b_.abs().max().item() + 1e-30
        assert (a_ - b_).abs().max().item() < 2e-5 * scale + 1e-6, (a_ - b_).abs().max().item() / scale


@pytest.mark.parametrize("shape", [(2, 16, 16, 64, 64), (1, 18, 34, 64, 128), (1, 8, 8, 256, 256), (3, 40, 56, 128, 64), (2, 18, 34, 512, 64)])
@pytest.mark.parametrize("vanishing", [False, True])
def test_x6_batchnorm_apply_on_load_matches_the_oracle(hip, shape, vanishing):
    # BatchNorm-apply on load through the BF16x6 kernel (unet_winograd_weight_fold_x6: scaled three-piece weights, folded bias, per-channel
    # padding value) against the fp64 oracle on the TRUE BatchNorm output, interior and every border pixel; with vanishing scales
    # (|s| down to 0: the padding value reaches 1e30) on half of the channels as well.  The padding values pass through the SPLIT like
    # any activation, so this also pins the split on huge magnitudes.
    n, h, w, ci, co = shape
    g = torch.Generator(device=DEV).manual_seed(h * w + ci + int(vanishing))
    ldx = ci + 4
    r = torch.relu(torch.randn(n, h, w, ldx, device=DEV, generator=g))
    sc = (torch.rand(ci, device=DEV, generator=g) + 0.5) * (torch.randint(0, 2, (ci,), device=DEV, generator=g).float() * 2 - 1)
    if vanishing:
        tiny = [0.0, -0.0, 1e-38, -1e-38, 1e-30, -1e-30, 1e-6, -1e-6, 1e-4, -1e-4, 1e-2, -1e-2]
        for j in range(0, ci, 2):
            sc[j] = tiny[(j // 2) % len(tiny)]
    sh = torch.randn(ci, device=DEV, generator=g) * 3 + 2.0
    wt = torch.randn(3, 3, ci, co, device=DEV, generator=g) * 0.05; b = torch.randn(co, device=DEV, generator=g)
    U6 = torch.empty(hip.unet_winograd_x6_weight_bytes(ci, co), dtype=torch.uint8, device=DEV)
    bf = torch.empty(co, device=DEV); pad = torch.empty(ci + 8, device=DEV)
    hip.unet_winograd_weight_fold_x6(P(wt), P(b), P(sc), P(sh), P(U6), P(bf), P(pad), ci, co, ST())
    out = torch.empty(n, h, w, co, device=DEV)
    hip.unet_conv3x3_fwd_winograd_x6(P(r), ldx, P(pad), P(U6), P(bf), P(out), co, n, h, w, ci, co, 1, None, 0, ST())
    assert torch.isfinite(pad).all() and (pad[ci:] == 0).all() and torch.isfinite(out).all()
    if not vanishing:
        assert torch.allclose(pad[:ci], -sh / sc, rtol=1e-6)
        # same folded bias and padding values as the native fold
        Uf = torch.empty(16 * ci * co, device=DEV); bf2 = torch.empty(co, device=DEV); pad2 = torch.empty(ci + 8, device=DEV)
        nbf = hip.unet_winograd_weight_fold_workspace(ci, co); wsf = ws_bytes(nbf)
        hip.unet_winograd_weight_fold(P(wt), P(b), P(sc), P(sh), P(Uf), P(bf2), P(pad2), ci, co, P(wsf), nbf, ST())
        assert torch.equal(pad, pad2) and torch.allclose(bf, bf2, rtol=1e-6, atol=1e-6)
    yref = (sc.double() * r.double()[..., :ci] + sh.double()).cpu().numpy()
    ref = on.relu_fwd(on.conv_same_fwd(yref.transpose(0, 3, 1, 2), wt.double().cpu().numpy(), b.double().cpu().numpy())).transpose(0, 2, 3, 1)
    scale_ = np.abs(ref).max()
    err = np.abs(out.cpu().numpy() - ref)
    border = np.ones((h, w), bool); border[1:-1, 1:-1] = False
    assert err.max() < 3e-5 * scale_ and err[:, border].max() < 3e-5 * scale_, (err.max() / scale_, err[:, border].max() / scale_)


@pytest.mark.parametrize("shape", [(8, 512, 512, 64, 64), (8, 512, 512, 128, 64), (8, 32, 32, 1024, 1024), (8, 128, 128, 256, 256)])
def test_x6_forward_and_data_gradient_at_full_size(hip, shape):
    # BASELINE config-2 layer shapes (the oracle is too slow here): the BF16x6 forward agrees with the native fused Winograd kernel to
    # fp32 rounding, and forward / data gradient are adjoint:  <conv(x, w), dz> == <x, dgrad(dz, w)>  (inner products in fp64);
    # 8192 / 8192 / 512 tile blocks on 256 persistent workgroups, image borders included; bit-reproducible from launch to launch
    n, h, w, ci, co = shape
    x, dz, wt, _ = fc.inputs(shape, ci + co)
    y6 = torch.empty(n, h, w, co, device=DEV); y6b = torch.empty_like(y6); yn = torch.empty_like(y6); dx = torch.empty_like(x)
    U6, U6d, Uc = fc.x6_weights(hip, wt, 0), fc.x6_weights(hip, wt, 1), fc.native_weights(hip, wt, 2)
    hip.unet_conv3x3_fwd_winograd_x6(P(x), ci, None, P(U6), None, P(y6), co, n, h, w, ci, co, 0, None, 0, ST())
    hip.unet_conv3x3_fwd_winograd_x6(P(x), ci, None, P(U6), None, P(y6b), co, n, h, w, ci, co, 0, None, 0, ST())
    hip.unet_conv3x3_fwd_winograd_fused(P(x), ci, None, P(Uc), None, P(yn), co, n, h, w, ci, co, 0, None, 0, ST())
    assert torch.equal(y6, y6b)
    assert ((y6 - yn).abs().max() / yn.abs().max()).item() < 3e-6
    hip.unet_conv3x3_dgrad_winograd_x6(P(dz), co, P(U6d), P(dx), ci, n, h, w, ci, co, None, 0, 0, 0, None, 0, ST())
    dot = lambda a, b: (a.double() * b.double()).sum().item()
    a0, a1 = dot(y6, dz), dot(x, dx)
    assert abs(a0 - a1) < 2e-6 * float(np.sqrt(dot(y6, y6) * dot(dz, dz)))
