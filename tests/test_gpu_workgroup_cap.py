"""The `_wg` entry points (include/unet_hip.h: max_workgroups of every persistent kernel) against the uncapped launches of the same
kernels: outputs that do not depend on the grid are bit-identical, statistics partials (whose row count follows the grid) finalize to the
same sums, split-K weight gradients agree to fp32 summation-order noise.  The reference has no counterpart (its all-reduce overlap is
TensorFlow's, UNet/train.py:57-61); the uncapped kernels are pinned against the oracle elsewhere (tests/test_gpu_kernels.py)."""
import ctypes

import pytest
import torch

from conftest import pkg
import fp32_error_cases as F

pytestmark = pytest.mark.gpu
P, ST, DEV = F.P, F.ST, F.DEV
CAPS = (0, 224, 96, 32)
E = pkg("_lib").UnetHipError


def _sums(part, blocks, rows):
    return part.view(blocks, rows, 64, 2).double().sum(1)


@pytest.mark.parametrize("shape", [(2, 64, 96, 64, 128), (1, 128, 128, 128, 64)])
def test_winograd_forward_and_data_gradient(shape):
    L = pkg("_lib").lib()
    n, h, w, ci, co = shape
    x, dz, wt, b = F.inputs(shape, 3)
    r_prev = torch.randn(n, h, w, ci, device=DEV)
    ref = {}
    for x6 in (False, True):
        U = F.x6_weights(L, wt, 0) if x6 else F.native_weights(L, wt, 2)
        Ud = F.x6_weights(L, wt, 1) if x6 else F.native_weights(L, wt, 3)
        fwd = L.unet_conv3x3_fwd_winograd_x6_wg if x6 else L.unet_conv3x3_fwd_winograd_fused_wg
        bwd = L.unet_conv3x3_dgrad_winograd_x6_wg if x6 else L.unet_conv3x3_dgrad_winograd_fused_wg
        for cap in CAPS:
            rows = L.unet_conv3x3_fwd_winograd_fused_stats_rows_wg(n, h, w, ci, co, cap)
            rows_d = L.unet_conv3x3_fwd_winograd_fused_stats_rows_wg(n, h, w, co, ci, cap)
            assert rows > 0 and rows_d > 0
            if cap == 0:
                assert rows == L.unet_conv3x3_fwd_winograd_fused_stats_rows(n, h, w, ci, co)
            part = torch.full(((co // 64) * rows * 128,), float("nan"), device=DEV); out = torch.empty(n, h, w, co, device=DEV)
            fwd(P(x), ci, None, P(U), P(b), P(out), co, n, h, w, ci, co, 1, P(part), part.numel() * 4, cap, ST())
            partd = torch.full(((ci // 64) * rows_d * 128,), float("nan"), device=DEV); dx = torch.empty(n, h, w, ci, device=DEV)
            bwd(P(dz), co, P(Ud), P(dx), ci, n, h, w, ci, co, P(r_prev), ci, 0, ci, P(partd), partd.numel() * 4, cap, ST())
            got = (out, _sums(part, co // 64, rows), dx, _sums(partd, ci // 64, rows_d))
            if cap == 0:
                ref[x6] = got
                continue
            assert torch.equal(got[0], ref[x6][0]) and torch.equal(got[2], ref[x6][2]), (x6, cap)
            for a, r in ((got[1], ref[x6][1]), (got[3], ref[x6][3])):
                assert torch.allclose(a, r, rtol=1e-5, atol=1e-5 * float(r.abs().max())), (x6, cap)
            # a partial buffer sized for a LARGER grid than the launch's is refused, not overrun in the other direction
            with pytest.raises(E, match="workspace too small"):
                fwd(P(x), ci, None, P(U), P(b), P(out), co, n, h, w, ci, co, 1, P(part), 16, cap, ST())


def test_weight_gradients_follow_the_cap():
    L = pkg("_lib").lib()
    n, h, w, ci, co = 2, 32, 64, 128, 128
    x, dz, wt, b = F.inputs((n, h, w, ci, co), 5)
    x16, dz16 = x.bfloat16(), dz.bfloat16()
    for name, ws_fn, run in (
        ("winograd", lambda c: L.unet_conv3x3_wgrad_winograd_fused_workspace(n, h, w, ci, co, c),
         lambda c, dw, ws, nb: L.unet_conv3x3_wgrad_winograd_fused(P(x), ci, P(dz), co, P(dw), n, h, w, ci, co, c, P(ws), nb, ST())),
        ("mfma", lambda c: L.unet_conv3x3_wgrad_mfma_workspace_wg(n, h, w, ci, co, c),
         lambda c, dw, ws, nb: L.unet_conv3x3_wgrad_mfma_wg(P(x), ci, P(dz), co, P(dw), n, h, w, ci, co, c, P(ws), nb, ST())),
        ("bf16", lambda c: L.unet_conv3x3_wgrad_bf16_workspace_wg(n, h, w, ci, co, c),
         lambda c, dw, ws, nb: L.unet_conv3x3_wgrad_bf16_wg(P(x16), ci, 1, P(dz16), co, 1, P(dw), n, h, w, ci, co, c, P(ws), nb, ST())),
    ):
        ref = None
        for cap in CAPS:
            nb = ws_fn(cap); ws = torch.empty(nb + 256, dtype=torch.uint8, device=DEV)
            dw = torch.full((3, 3, ci, co), float("nan"), device=DEV)
            run(cap, dw, ws, nb)
            if cap == 0:
                ref = dw
                full = nb
            else:
                assert nb <= full, (name, cap)
                assert torch.allclose(dw, ref, rtol=2e-5, atol=2e-5 * float(ref.abs().max())), (name, cap, float((dw - ref).abs().max()))
                if nb > 16:
                    with pytest.raises(E, match="workspace too small"):
                        run(cap, dw, ws, nb - 16)


def test_transposed_conv_kernels_follow_the_cap():
    L = pkg("_lib").lib()
    n, h, w, ci, co = 2, 32, 32, 256, 128
    g = torch.Generator(device=DEV).manual_seed(11)
    x = torch.randn(n, h, w, ci, device=DEV, generator=g); wT = torch.randn(2, 2, co, ci, device=DEV, generator=g) / 16
    b = torch.randn(co, device=DEV, generator=g); dz = torch.randn(n, 2 * h, 2 * w, co, device=DEV, generator=g)
    ref = None
    for cap in CAPS:
        rows = L.unet_convT2x2_fwd_stream_stats_rows_wg(n, h, w, ci, co, cap)
        assert rows > 0
        part = torch.full(((co // 64) * rows * 128,), float("nan"), device=DEV); out = torch.empty(n, 2 * h, 2 * w, co, device=DEV)
        L.unet_convT2x2_fwd_stream_wg(P(x), ci, P(wT), P(b), P(out), co, n, h, w, ci, co, P(part), part.numel() * 4, cap, ST())
        out2 = torch.empty_like(out)
        L.unet_convT2x2_fwd_stream_wg(P(x), ci, P(wT), P(b), P(out2), co, n, h, w, ci, co, None, 0, cap, ST())
        nb = L.unet_convT2x2_wgrad_workspace_wg(n, h, w, ci, co, cap); ws = torch.empty(nb + 256, dtype=torch.uint8, device=DEV)
        dw = torch.full((2, 2, co, ci), float("nan"), device=DEV)
        L.unet_convT2x2_wgrad_wg(P(x), ci, P(dz), co, P(dw), n, h, w, ci, co, cap, P(ws), nb, ST())
        nb16 = L.unet_convT2x2_wgrad_bf16_workspace_wg(n, h, w, ci, co, cap); ws16 = torch.empty(nb16 + 256, dtype=torch.uint8, device=DEV)
        dw16 = torch.full((2, 2, co, ci), float("nan"), device=DEV)
        L.unet_convT2x2_wgrad_bf16_wg(P(x.bfloat16()), ci, 1, P(dz.bfloat16()), co, 1, P(dw16), n, h, w, ci, co, cap, P(ws16), nb16, ST())
        nbx = L.unet_convT2x2_wgrad_x6_workspace_wg(n, h, w, ci, co, cap); wsx = torch.empty(nbx + 256, dtype=torch.uint8, device=DEV)
        dwx = torch.full((2, 2, co, ci), float("nan"), device=DEV)
        L.unet_convT2x2_wgrad_x6_wg(P(x), ci, P(dz), co, P(dwx), n, h, w, ci, co, cap, P(wsx), nbx, ST())
        got = (out, _sums(part, co // 64, rows), dw, dw16, dwx)
        assert torch.equal(out, out2)
        if cap == 0:
            ref = got
            continue
        assert torch.equal(got[0], ref[0])
        for a, r in zip(got[1:], ref[1:]):
            assert torch.allclose(a, r, rtol=2e-5, atol=2e-5 * float(r.abs().max())), cap


def test_bf16_convolutions_are_bit_identical_under_the_cap():
    L = pkg("_lib").lib()
    n, h, w, ci, co = 2, 64, 96, 128, 128
    x, dz, wt, b = F.inputs((n, h, w, ci, co), 7)
    x16, dz16 = x.bfloat16(), dz.bfloat16()
    wp = torch.empty(L.unet_conv3x3_bf16_packed_bytes(ci, co), dtype=torch.uint8, device=DEV); wpd = torch.empty_like(wp)
    L.unet_conv3x3_bf16_pack_weights(P(wt), P(wp), ci, co, 0, ST()); L.unet_conv3x3_bf16_pack_weights(P(wt), P(wpd), ci, co, 1, ST())
    rows = L.unet_conv3x3_bf16_stats_rows(n, h, w, ci, co)
    ref = None
    for cap in CAPS:
        part = torch.full(((co // 64) * rows * 128,), float("nan"), device=DEV)
        out = torch.empty(n, h, w, co, dtype=torch.bfloat16, device=DEV); dx = torch.empty(n, h, w, ci, dtype=torch.bfloat16, device=DEV)
        L.unet_conv3x3_fwd_bf16_wg(P(x16), ci, 1, None, None, P(wp), P(b), P(out), co, 1, n, h, w, ci, co, 1, P(part), part.numel() * 4, cap, ST())
        L.unet_conv3x3_dgrad_bf16_wg(P(dz16), co, 1, P(wpd), P(dx), ci, 1, n, h, w, ci, co, None, 0, 0, 0, 0, None, 0, cap, ST())
        got = (out, part, dx)
        if cap == 0:
            ref = got
        else:
            assert all(torch.equal(a, r) for a, r in zip(got, ref)), cap        # per-tile statistics rows: identical too
