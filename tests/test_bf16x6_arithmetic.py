"""The arithmetic claim behind the BF16x6 kernels (csrc/winograd_x6.hip, csrc/convt_x6.hip), restated in numpy and checked on the CPU:
an fp32 value is EXACTLY the sum of three bf16 pieces h = rn(v), m = rn(v - h), l = v - h - m (round-to-nearest-even, the kernels'
v_cvt_pk_bf16_f32; rounds 4-5 took the pieces by truncation, whose dropped terms were a bias of the product's sign up to 2^-21); every piece
product is exact in fp32; the six kept products hh, hm, mh, hl, lh, mm differ from a * b by the three dropped terms: ZERO-MEAN, at most
2^-24.4, rms 2^-27.4 of |a b| (an fp32 multiply's own rounding is up to 2^-24); a dot product accumulated from them in fp32 (six roundings of
the accumulator per 16 terms) is closer to the fp64 result than the term-by-term fp32 dot product, whose every product is rounded.
(The reference computes these layers in fp32: UNet/model.py:28-48; this pins what "fp32-grade" means for the route that replaces the fp32
matrix instruction.)  Not covered, as in the kernels: |v| within half a bf16 ulp of FLT_MAX (h rounds to Inf), Inf / NaN inputs."""
import numpy as np


def rn_bf16(v):
    u = v.view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32).view(np.float32)


def trunc_bf16(v):
    return (v.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)


def split3(v, piece=rn_bf16):
    h = piece(v)
    a = (v - h).astype(np.float32)          # exact: at most 16 significant bits, a multiple of v's fp32 ulp
    m = piece(a)
    b = (a - m).astype(np.float32)          # exact: at most 8 significant bits
    l = piece(b)
    return h, m, l, (b - l).astype(np.float32)


def samples(rng, n):
    v = rng.standard_normal(n).astype(np.float32) * np.float32(10.0) ** rng.integers(-20, 20, n).astype(np.float32)
    edge = np.array([0x3F800000, 0x3F7FFFFF, 0x3F808000, 0x3F807FFF, 0x3F80FFFF, 0x00800000, 0x7F7F7FFF, 0x3F800001, 0x3F818000, 0x3F828000],
                    dtype=np.uint32).view(np.float32)          # (0x7F7F7FFF: the largest value whose high piece stays finite)
    return np.concatenate([v, edge, -edge])


def test_three_rounded_pieces_are_an_exact_split():
    v = samples(np.random.default_rng(1), 200000)
    h, m, l, rest = split3(v)
    assert np.all(rest == 0)                                                     # 8 + 8 + 8 significant bits cover the 24-bit significand
    assert np.array_equal((h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64)).astype(np.float32), v)
    for p in (h, m, l):
        assert np.all((p.view(np.uint32) & np.uint32(0xFFFF)) == 0)              # each piece is a bf16 value
    nz = v != 0
    assert np.all(np.abs(m[nz]) <= np.abs(h[nz]) * 2.0 ** -8) and np.all(np.abs(l[nz]) <= np.abs(h[nz]) * 2.0 ** -16)
    assert 0.4 < np.mean(np.sign(m[nz]) != np.sign(v[nz])) < 0.6                 # the remainders take either sign: nothing is biased
    # (the truncation split of rounds 4-5 is exact too, with every piece of the value's sign)
    ht, mt, lt, rt = split3(v, trunc_bf16)
    assert np.all(rt == 0) and np.all((mt == 0) | (np.sign(mt) == np.sign(v)))


def test_six_piece_products_are_exact_and_the_dropped_terms_are_zero_mean():
    rng = np.random.default_rng(2)
    a, b = samples(rng, 100000), samples(rng, 100000)
    keep = np.isfinite(a.astype(np.float64) * b.astype(np.float64)) & (np.abs(a.astype(np.float64) * b.astype(np.float64)) > 1e-30) \
        & (np.abs(a.astype(np.float64) * b.astype(np.float64)) < 1e30)
    a, b = a[keep], b[keep]
    exact = a.astype(np.float64) * b.astype(np.float64)

    def dropped(piece):
        ah, am, al, _ = split3(a, piece); bh, bm, bl, _ = split3(b, piece)
        total = np.zeros(a.shape, np.float64)
        for x, y in [(ah, bh), (ah, bm), (am, bh), (ah, bl), (al, bh), (am, bm)]:
            p64 = x.astype(np.float64) * y.astype(np.float64)
            assert np.array_equal((x * y).astype(np.float64), p64)               # 8 x 8 significant bits: exact in an fp32 accumulator
            total += p64
        return (exact - total) / exact                                           # the dropped am bl + al bm + al bl, relative to a b

    rel = dropped(rn_bf16)
    # |m| <= 2^-8 |a|-ish, |l| <= 2^-16: the dropped terms are below 2^-24.4 |a b| -- under an fp32 multiply's own rounding (up to 2^-24) --
    # and they take either sign: mean ~ 0 (against an rms of 2^-27.4), half of them positive
    assert np.max(np.abs(rel)) < 2.0 ** -24.3 and 2.0 ** -28 < np.sqrt(np.mean(rel ** 2)) < 2.0 ** -27, (np.max(np.abs(rel)), np.sqrt(np.mean(rel ** 2)))
    assert abs(np.mean(rel)) < 0.02 * np.sqrt(np.mean(rel ** 2)) and 0.48 < np.mean(rel > 0) < 0.52, (np.mean(rel), np.mean(rel > 0))
    # the truncation split it replaces: every dropped term of the product's sign, up to 2^-21, mean 2^-24.5 -- ~8x the size and a bias
    relt = dropped(trunc_bf16)
    assert np.all(relt >= 0) and np.max(relt) < 2.0 ** -21 and 2.0 ** -25.5 < np.mean(relt) < 2.0 ** -24
    assert np.mean(np.abs(relt)) > 8 * np.mean(np.abs(rel))
    assert np.max(np.abs((a * b).astype(np.float64) - exact) / np.abs(exact)) <= 2.0 ** -24


def test_dot_product_from_pieces_is_fp32_grade():
    rng = np.random.default_rng(3)
    K, R = 4096, 64
    a = rng.standard_normal((R, K)).astype(np.float32); b = rng.standard_normal((R, K)).astype(np.float32)
    ref = (a.astype(np.float64) * b.astype(np.float64)).sum(1)
    ah, am, al, _ = split3(a.ravel()); bh, bm, bl, _ = split3(b.ravel())
    order = [(al, bh), (ah, bl), (am, bm), (am, bh), (ah, bm), (ah, bh)]         # the kernels' order: smallest products first
    acc6 = np.zeros(R, np.float32); acc1 = np.zeros(R, np.float32); acc3 = np.zeros(R, np.float32)
    for k0 in range(0, K, 16):                                                    # one MFMA K-step = 16 terms, accumulated in fp32
        for x, y in order:
            acc6 = (acc6 + (x.reshape(R, K)[:, k0:k0 + 16] * y.reshape(R, K)[:, k0:k0 + 16]).sum(1, dtype=np.float32)).astype(np.float32)
        for x, y in order[3:]:                                                    # a three-product emulation (hh + hm + mh) for contrast
            acc3 = (acc3 + (x.reshape(R, K)[:, k0:k0 + 16] * y.reshape(R, K)[:, k0:k0 + 16]).sum(1, dtype=np.float32)).astype(np.float32)
        for k in range(k0, k0 + 16):                                              # the textbook fp32 dot product: one rounding per term
            acc1 = (acc1 + a[:, k] * b[:, k]).astype(np.float32)
    scale = np.abs(a.astype(np.float64) * b.astype(np.float64)).sum(1)
    e6 = np.sqrt(np.mean(((acc6 - ref) / scale) ** 2)); e1 = np.sqrt(np.mean(((acc1 - ref) / scale) ** 2)); e3 = np.sqrt(np.mean(((acc3 - ref) / scale) ** 2))
    assert e6 < 0.6 * e1 and e6 < 2e-8, (e6, e1)                                  # (measured 0.47: the piece products are exact, the fp32 products are not)
    assert e3 > 3 * e6, (e3, e6)                                                  # ... which is what the per-kernel error bounds of the GPU tests reject
