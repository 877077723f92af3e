"""The arithmetic claim behind the BF16x6 kernels (csrc/winograd_x6.hip, csrc/convt_x6.hip), restated in numpy and checked on the CPU:
an fp32 value is EXACTLY the sum of three bf16 pieces taken by truncation; every piece product is exact in fp32; the six kept products
hh, hm, mh, hl, lh, mm fall short of a * b by the three dropped terms: at most 2^-21, on average 2^-24.5 of |a b| (an fp32 multiply's own rounding is up to 2^-24);
a dot product accumulated from them in fp32 (six roundings of the accumulator per 16 terms) is as close to the fp64 result as the
term-by-term fp32 dot product.  (The reference computes these
layers in fp32: UNet/model.py:28-48; this pins what "fp32-grade" means for the route that replaces the fp32 matrix instruction.)"""
import numpy as np


def trunc_bf16(v):
    return (v.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)


def split3(v):
    h = trunc_bf16(v)
    a = (v - h).astype(np.float32)          # exact: the low 16 mantissa bits of v
    m = trunc_bf16(a)
    b = (a - m).astype(np.float32)
    l = trunc_bf16(b)
    return h, m, l, (b - l).astype(np.float32)


def samples(rng, n):
    v = rng.standard_normal(n).astype(np.float32) * np.float32(10.0) ** rng.integers(-20, 20, n).astype(np.float32)
    edge = np.array([0x3F800000, 0x3F7FFFFF, 0x3F808000, 0x3F807FFF, 0x3F80FFFF, 0x00800000, 0x7F7FFFFF, 0x3F800001], dtype=np.uint32).view(np.float32)
    return np.concatenate([v, edge, -edge])


def test_three_truncated_pieces_are_an_exact_split():
    v = samples(np.random.default_rng(1), 200000)
    h, m, l, rest = split3(v)
    assert np.all(rest == 0)                                                     # 8 + 8 + 8 significant bits cover the 24-bit significand
    assert np.array_equal((h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64)).astype(np.float32), v)
    for p in (h, m, l):
        assert np.all((p.view(np.uint32) & np.uint32(0xFFFF)) == 0)              # each piece is a bf16 value
        assert np.all((p == 0) | (np.sign(p) == np.sign(v)))                     # and carries the sign of the value


def test_six_piece_products_are_exact_and_leave_one_rounding():
    rng = np.random.default_rng(2)
    a, b = samples(rng, 100000), samples(rng, 100000)
    keep = np.isfinite(a.astype(np.float64) * b.astype(np.float64)) & (np.abs(a.astype(np.float64) * b.astype(np.float64)) > 1e-30) \
        & (np.abs(a.astype(np.float64) * b.astype(np.float64)) < 1e30)
    a, b = a[keep], b[keep]
    ah, am, al, _ = split3(a); bh, bm, bl, _ = split3(b)
    pairs = [(ah, bh), (ah, bm), (am, bh), (ah, bl), (al, bh), (am, bm)]
    total = np.zeros(a.shape, np.float64)
    for x, y in pairs:
        p64 = x.astype(np.float64) * y.astype(np.float64)
        assert np.array_equal((x * y).astype(np.float64), p64)                   # 8 x 8 significant bits: exact in an fp32 accumulator
        total += p64
    exact = a.astype(np.float64) * b.astype(np.float64)
    rel = (exact - total) / exact                                                # the dropped am bl + al bm + al bl, relative to a b
    # truncation pieces carry the sign of their value, so every dropped term has the sign of the product: the six-product sum is short by
    # |m| < 2^-7 |a|, |l| < 2^-15 |a|  =>  at most 2 * 2^-22 = 2^-21 of |a b|, on average 2^-24.5 (4e-8) -- a bias towards zero the size of
    # an fp32 multiply's own rounding (which is at most 2^-24, without bias)
    assert np.all(rel >= 0) and np.max(rel) < 2.0 ** -21 and 2.0 ** -25.5 < np.mean(rel) < 2.0 ** -24, (np.max(rel), np.mean(rel))
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
    assert e6 < 1.25 * e1 and e6 < 1e-7, (e6, e1)
    assert e3 > 3 * e6, (e3, e6)                                                  # ... which is what the per-kernel error bounds of the GPU tests reject
