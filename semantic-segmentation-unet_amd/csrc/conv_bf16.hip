// 3x3 convolution on the bf16 matrix cores (BASELINE config 4: "bf16 forward/backward with fp32 master weights").
//
// Reference semantics: the same UNet._conv_layer convolution as the fp32 kernels (UNet/model.py:28-37); the reference keeps a
// mixed-precision policy in comments only (UNet/train.py:52-54), so the contract here is the usual one of that policy: the
// operands of the contraction (activations, weights) are rounded to bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32), products
// are exact and accumulated in fp32 (v_mfma_f32_32x32x16_bf16), bias / ReLU / BatchNorm / loss / Adam stay fp32 on fp32
// master weights.  Activations stay fp32 in HBM in this stage: the kernel converts while it stages its input patch.
//
// Implicit GEMM, one workgroup (4 waves, one per SIMD, 256 accumulator registers) = 16 x 32 output pixels x CT = 128 (or 64)
// output channels; the reduction runs in chunks of 16 input channels x 9 taps:
//   * input patch 18 x 34 pixels x 16 channels: fp32 buffer loads (out-of-image lanes return 0 through the buffer's range
//     check, no branches) into registers one chunk ahead, converted and written as two bf16 planes [k half][pixel][8] -- an
//     MFMA A fragment (32 consecutive pixels of a row, 8 channels per lane) is one conflict-free ds_read_b128 whose address
//     differs between taps by an immediate;
//   * weights pre-packed on the device as [chunk][tap][k half][Cout][8] bf16, so a chunk's share is 1-KB pieces that LDS-DMA
//     copies verbatim and a B fragment is again one linear ds_read_b128;
//   * per chunk and wave 144 (72) MFMAs from 18 A + 36 (18) B fragment reads: every A fragment (patch row q, column shift b)
//     feeds the up to three taps a with output row q - a, every B fragment four output rows; the stream is explicit
//     (asm volatile) with the next group's reads issued behind the current group's MFMAs and counted lgkmcnt waits.
// The data gradient is the same kernel on dz with weights packed flipped / transposed (mode 1).
#include "common.h"
#include <stdlib.h>

#ifndef UNET_CB_ABLATE
#define UNET_CB_ABLATE 0        /* diagnostic builds (scripts/build_variant.sh): see conv_bf16_body */
#endif
#ifndef UNET_CBS_ABLATE
#define UNET_CBS_ABLATE 0       /* diagnostic builds of the persistent kernels (results wrong): 1 no DMA, 2 no MFMA stream, 4 no epilogue, 8 no patch DMA, 16 no weight DMA, 32 patch DMA from contiguous memory */
#endif

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_b;

struct ConvBf16Args {
    const float* x; const uint16_t* wp; const float* bias; float* out;
    int ldx, ldo, N, H, W, Cin, Cout, relu;
    int tby, tbx, n_px, n_co;
    unsigned x_bytes;
    int in16;                    // the input tensor is stored as bf16 (ldx in elements): staged without conversion
    int out16, r16;              // the output / the producer's saved activation (STATS 2) is stored as bf16 (ldo / bn_ldr in elements)
    float* stat_part;            // STATS 1: BatchNorm sums of the output (sum y, sum y^2); STATS 2: BatchNorm-backward sums (sum dx, sum dx * r)
    const float* bn_r; int bn_ldr, bn_c0, bn_c1;     // STATS 2: saved activation of the producer layer, whose dy is dx[..., c0:c1)
    const float* in_scale; const float* in_shift;    // NORM: the input is a producer's conv output r; the operand is bf16(scale * r + shift)
};

constexpr int kPW = 34, kPlane = 640 * 16, kXP = 2 * kPlane;        // patch row length (pixels), bytes of one k-half plane, of the patch

#define CB_RD128(dst, base, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst) : "v"(base), "n"(off))
// A read of the interleaved x image: the lanes whose k half sits in the upper 16-byte slot of its pixel are a wave-wide constant per
// fragment (an SGPR pair from a ballot), so the slot select is one v_cndmask next to the read and costs no vector register
#define CB_RD128_SEL(dst, base0, base1, lanemask, off) do { unsigned t_; asm volatile("v_cndmask_b32 %1, %2, %3, %4\n\tds_read_b128 %0, %1 offset:%5" \
                                                        : "=&v"(dst), "=&v"(t_) : "v"(base0), "v"(base1), "s"(lanemask), "n"(off)); } while (0)
// (the 64-channel-tile kernels are at their register budget and the 18 SGPR pairs spill there: they take the lane's own 18-bit mask and
// two VALU instructions per read instead; same-box A/B of 64->64 @512^2 data gradient + sums: 0.246 vs 0.233 ms)
#define CB_RD128_BFE(dst, base, mask, bit, off) do { unsigned t_; asm volatile("v_bfe_u32 %1, %3, %4, 1\n\tv_lshl_add_u32 %1, %1, 4, %2\n\tds_read_b128 %0, %1 offset:%5" \
                                                        : "=&v"(dst), "=&v"(t_) : "v"(base), "v"(mask), "n"(bit), "n"(off)); } while (0)
#define CB_MFMA(accv, av, bv) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(accv) : "v"(av), "v"(bv) : "memory")

template <int NCO> struct CbFrags { bf16x8 a[2]; bf16x8 b[2][3][NCO]; };

__device__ __forceinline__ unsigned cb_pack2(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// the same, pinned in program order: in the staging path the conversion must stay BEHIND the MFMA stream (it carries the
// compiler's wait for the prefetched loads with it)
__device__ __forceinline__ unsigned cb_pack2_pinned(float lo, float hi) {
    unsigned r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

// wait until at most `n` LDS operations are outstanding; ties every fragment the following MFMAs read to the wait (each
// register exactly once: a repeated "+v" operand would be copied BEFORE the wait)
#define CB_WAIT_TIED(n, ...) asm volatile("s_waitcnt lgkmcnt(" #n ")" : __VA_ARGS__)
#define CB_TIES4(a, b) "+v"(a), "+v"(b[0][0]), "+v"(b[1][0]), "+v"(b[2][0]), "+v"(b[0][1]), "+v"(b[1][1]), "+v"(b[2][1]), \
                       "+v"(b[0][2]), "+v"(b[1][2]), "+v"(b[2][2]), "+v"(b[0][3]), "+v"(b[1][3]), "+v"(b[2][3])
#define CB_TIES2(a, b) "+v"(a), "+v"(b[0][0]), "+v"(b[1][0]), "+v"(b[2][0]), "+v"(b[0][1]), "+v"(b[1][1]), "+v"(b[2][1])
template <int NCO> __device__ __forceinline__ void cb_wait_ab(int issued, bf16x8& a, bf16x8 (&b)[3][NCO]);
template <> __device__ __forceinline__ void cb_wait_ab<4>(int issued, bf16x8& a, bf16x8 (&b)[3][4]) {
    if (issued == 3) CB_WAIT_TIED(3, CB_TIES4(a, b)); else if (issued == 2) CB_WAIT_TIED(2, CB_TIES4(a, b)); else CB_WAIT_TIED(1, CB_TIES4(a, b));
}
template <> __device__ __forceinline__ void cb_wait_ab<2>(int issued, bf16x8& a, bf16x8 (&b)[3][2]) {
    if (issued == 3) CB_WAIT_TIED(3, CB_TIES2(a, b)); else if (issued == 2) CB_WAIT_TIED(2, CB_TIES2(a, b)); else CB_WAIT_TIED(1, CB_TIES2(a, b));
}

// Slot (16-byte position in the row of one (tap, k half)) of output-channel column q in the LDS weight image, and the column kept at a slot.
// A lane reads column q = NCO j + c, so a linear image would put a ds_read_b128 lane group on 4 (NCO = 4) or 8 (NCO = 2) slots.  The groups
// are fixed by the hardware: lanes {0-3, 12-15, 20-27} and {4-11, 16-19, 28-31} of each half-wave (MI355X_MICROARCH.md, LDS).
//   NCO = 4: a rotation inside every aligned group of 16 columns -- conflict-free for those groups.
//   NCO = 2: that rotation left every group 2-way conflicted (SQ_LDS_BANK_CONFLICT 0.09 of the cycles against 0.28 active, also in the MFMA-only
//            build; the 128-channel tiles read 0.000): instead the 16 lanes of group g reading sub-tile c get the 16 slots of bank row 2 g + c.
template <int NCO> __device__ __forceinline__ int cb_wslot(int q) {
    if constexpr (NCO == 4) return (q & ~15) + ((q + (q >> 4)) & 15);
    else {
        const int j = q >> 1, c = q & 1;
        const bool g0 = j < 4 || (j >= 12 && j < 16) || (j >= 20 && j < 28);
        const int r = g0 ? (j < 4 ? j : j < 16 ? j - 8 : j - 12) : (j < 12 ? j - 4 : j < 20 ? j - 8 : j - 16);
        return 16 * (2 * (g0 ? 0 : 1) + c) + r;
    }
}
template <int NCO> __device__ __forceinline__ int cb_wcol(int pos) {
    if constexpr (NCO == 4) return (pos & ~15) + ((pos - (pos >> 4)) & 15);
    else {
        const int row = pos >> 4, r = pos & 15, c = row & 1;
        const int j = (row >> 1) == 0 ? (r < 4 ? r : r < 8 ? r + 8 : r + 12) : (r < 8 ? r + 4 : r < 12 ? r + 8 : r + 16);
        return 2 * j + c;
    }
}

// one chunk (16 input channels x 9 taps) from LDS stage ST: 36 * NCO MFMAs per wave.  Groups (bs, q) = (column shift, patch
// row); group g issues the A fragment of group g + 1 and its share of the next column shift's B fragments, waits for its own
// (counted: LDS operations retire in order) and runs its MFMAs.
// XL = 1 (the persistent kernels): the x image is [pixel][k half][8] -- 32 bytes per pixel, the two halves of a pixel swapped where bit 3 of
// the pixel index is set (keeps the 16 lanes of a ds_read_b128 group on 16 different 16-byte slots although the lane stride is 32 bytes).
// The A fragment of group g is pixel P0 + d_g of this lane; asel[g] = the lanes whose k half sits in the upper slot (a ballot, SGPR pair).
template <int NCO, int ST, int STAGE_BYTES, int XL = 0, class Fill>
__device__ __forceinline__ void cb_compute(f32x16 (&acc)[4][NCO], unsigned a_base0, const unsigned (&b_base0)[NCO], Fill&& fill,
                                           const unsigned long long* asel = nullptr, unsigned amask = 0) {
    constexpr int CT = 32 * NCO;
    constexpr int AO = 0, BO = 0;                                   // (the stage offset does not fit the 16-bit immediate)
    constexpr int APX = XL ? 32 : 16;                                // bytes between consecutive pixels in the A read address
    const unsigned a_base = a_base0 + ST * STAGE_BYTES;
    const unsigned a_base16 = a_base + 16;
    unsigned b_base[NCO];                                           // one base per sub-tile: its columns sit swizzled in the image
#pragma unroll
    for (int c = 0; c < NCO; ++c) b_base[c] = b_base0[c] + ST * STAGE_BYTES;
    constexpr int BPG = NCO / 2;                                   // next-shift B reads per group: 3 * NCO over 6 groups
    CbFrags<NCO> fr;
#pragma unroll
    for (int k = 0; k < 3 * NCO; ++k)
        CB_RD128(fr.b[0][k / NCO][k % NCO], b_base[k % NCO], BO + (3 * (k / NCO) + 0) * 2 * CT * 16);
    if (XL && NCO == 2) CB_RD128_BFE(fr.a[0], a_base, amask, 0, AO + 0);
    else if (XL) CB_RD128_SEL(fr.a[0], a_base, a_base16, asel[0], AO + 0); else CB_RD128(fr.a[0], a_base, AO + 0);
#pragma unroll
    for (int g = 0; g < 18; ++g) {
        const int bs = g / 6, q = g % 6;
        int issued = 0;
        if (g + 1 < 18) {
            const int bs2 = (g + 1) / 6, q2 = (g + 1) % 6;
            if (XL && NCO == 2) CB_RD128_BFE(fr.a[(g + 1) & 1], a_base, amask, g + 1, AO + (q2 * kPW + bs2) * APX);
            else if (XL) CB_RD128_SEL(fr.a[(g + 1) & 1], a_base, a_base16, asel[g + 1], AO + (q2 * kPW + bs2) * APX);
            else    CB_RD128(fr.a[(g + 1) & 1], a_base, AO + (q2 * kPW + bs2) * APX);
            ++issued;
        }
        if (bs < 2) {
#pragma unroll
            for (int k = q * BPG; k < (q + 1) * BPG; ++k) {
                CB_RD128(fr.b[(bs + 1) & 1][k / NCO][k % NCO], b_base[k % NCO], BO + (3 * (k / NCO) + bs + 1) * 2 * CT * 16);
                ++issued;
            }
        }
        // everything issued before this group has landed once at most `issued` operations are outstanding
        if (q == 0) cb_wait_ab<NCO>(issued, fr.a[g & 1], fr.b[bs & 1]);
        else {
            if (issued == 3)      asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(fr.a[g & 1]));
            else if (issued == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fr.a[g & 1]));
            else if (issued == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(fr.a[g & 1]));
            else                  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fr.a[g & 1]));
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int r = q - a;
            if (r < 0 || r > 3) continue;
#pragma unroll
            for (int c = 0; c < NCO; ++c) CB_MFMA(acc[r][c], fr.a[g & 1], fr.b[bs & 1][a][c]);
        }
        fill(g);          // the caller's share of the next chunk's staging, issued in the shadow of this group's MFMAs
    }
}

// Epilogue of one tile (shared by the per-tile and the persistent kernels): bias, ReLU, stores, fused BatchNorm sums.
// `red` = LDS scratch for the cross-wave sum of the statistics (CT * 8 floats, free of in-flight traffic), `row` = the tile's row of
// the partial-sum buffer.
// B16 = 1: the output and the saved activation are known to be bf16 tensors at compile time (the fp32 variants of the loads and
// stores, and the uniform branches that choose between them per store, drop out).
// EDGE = 0: the tile lies inside the image (every BASELINE shape: H % 16 == 0, W % 32 == 0) -- the per-column / per-row selects drop out
// (they were 588 of the 2550 vector instructions of a 128-channel tile's backward-sums epilogue).
// The bias is NOT added here: the accumulators start at the bias (cb_init_acc), which takes the place of their zero fill.
template <int NCO, int STATS, int B16 = 0, int EDGE = 1>
__device__ __forceinline__ void cb_epilogue(const ConvBf16Args& p, f32x16 (&acc)[4][NCO], int img, int ty0, int tx0, int co0, int row,
                                            float* red, int tid, int wv, int li, int lh) {
    constexpr int CT = 32 * NCO;
    // (every per-lane offset below is a function of the lane alone once the edge selects are gone: pinned here, or the compiler hoists ~40
    // of them out of a persistent caller's tile loop and spills the DMA source pointers of the chunk loop instead)
    if (!EDGE) asm volatile("" : "+v"(li), "+v"(lh));
    const bool out16 = B16 || p.out16, r16 = B16 || p.r16;
    // epilogue: accumulator register e of (row r, sub-tile c) = pixel (ty0 + 4 wv + r, tx0 + (e&3) + 8 (e>>2) + 4 lh), channel
    // co0 + NCO li + c (see the weight image): one buffer store per pixel and lane.  The 16 per-lane offsets (column, channel
    // group) are computed once, the row goes into the scalar offset; columns past the image edge get an offset the range check
    // drops.  out16 / r16: the output / the producer's saved activation is a bf16 tensor (same indexing, 2-byte elements).
    // (A variant with the column part in the scalar offset and scalar branches for the image edge made every launch 15-20 % slower
    // in a same-box A/B -- 64 basic blocks instead of one straight store stream -- and was dropped.)
    typedef unsigned ovec_t __attribute__((ext_vector_type(NCO)));
    typedef unsigned hvec_t __attribute__((ext_vector_type(NCO / 2)));
    float st1[NCO], st2[NCO];
    f32x2 st1p[NCO / 2], st2p[NCO / 2];                               // EDGE 0: the same sums, a channel pair per register pair
    const int oes = out16 ? 2 : 4, res = r16 ? 2 : 4;
    const __amdgpu_buffer_rsrc_t srd_o = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, (int)((size_t)p.N * p.H * p.W * p.ldo * oes), 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_r = __builtin_amdgcn_make_buffer_rsrc((void*)(STATS == 2 ? p.bn_r : p.out), 0,
                                                                            STATS == 2 ? (int)((size_t)p.N * p.H * p.W * p.bn_ldr * res) : 0, 0x00020000);
    const int cl = co0 + NCO * li;                                    // this lane's first channel
    const bool with_r = STATS == 2 && cl >= p.bn_c0 && cl < p.bn_c1;  // (c0, c1 multiples of 64: all NCO channels or none)
    // store offset of (row r, element e) = lane part (column 4 lh, the lane's channels) + row + column part, formed per store (two
    // vector instructions): kept as 16 precomputed registers they would crowd out the saved-activation window below
    const int vo = (4 * lh * p.ldo + NCO * li) * oes;
    const int wlim = p.W - tx0 - 4 * lh;                              // column (e & 3) + 8 (e >> 2) of this lane is inside the image iff < wlim
    // saved-activation loads (STATS 2): lane offset = column 4 lh and the lane's channels, or an offset the range check rejects when
    // the lane has no channel in [c0, c1); the column part (e & 3) + 8 (e >> 2) goes into the scalar offset; whether a column is
    // inside the image is uniform up to the half-wave (both halves / only lh = 0 / none): a select between three registers
    const int vr = with_r ? (4 * lh * p.bn_ldr + (cl - p.bn_c0)) * res : (int)0x80000000, vr_lo = lh ? (int)0x80000000 : vr;
    const int wrem = p.W - tx0;
    auto r_voff = [&](int e) { const int c0 = (e & 3) + 8 * (e >> 2); return !EDGE || c0 + 4 < wrem ? vr : (c0 < wrem ? vr_lo : (int)0x80000000); };
    const float lo = p.relu ? 0.f : -INFINITY;                        // (STATS 2 is the data gradient: no bias, no ReLU -- not even the v_max)
#pragma unroll
    for (int c = 0; c < NCO; ++c) { st1[c] = 0.f; st2[c] = 0.f; st1p[c >> 1] = f32x2{0.f, 0.f}; st2p[c >> 1] = f32x2{0.f, 0.f}; }
    // STATS 2 with a bf16 saved activation (the default storage): the 16 loads of a row are issued TWO rows ahead of the row being
    // written (rows 0 and 1 before any arithmetic, row r + 2 before row r is processed), so the memory latency is paid about once
    // per tile instead of eight times -- the per-row half-batches of the fp32 path below cost 0.24 ms of a 0.50 ms launch on
    // 64->64 @512^2, where the main loop (4 chunks) is too short to hide anything.
    constexpr int RING = NCO == 4 ? 3 : 4;          // rows of the saved activation in flight (NCO = 2: the whole tile)
    hvec_t rvh[RING][16];
    const bool hoist = STATS == 2 && r16 && !(UNET_CB_ABLATE & 2);
    auto issue_row = [&](int r) {
        const int gy = ty0 + 4 * wv + r;
        const int sr = ((img * p.H + (!EDGE || gy < p.H ? gy : 0)) * p.W + tx0) * p.bn_ldr * 2;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int vofs = !EDGE || gy < p.H ? r_voff(e) : (int)0x80000000, so_e = sr + ((e & 3) + 8 * (e >> 2)) * p.bn_ldr * 2;
            if constexpr (NCO == 4) rvh[r % RING][e] = __builtin_amdgcn_raw_buffer_load_b64(srd_r, vofs, so_e, 0);
            else                    rvh[r % RING][e][0] = __builtin_amdgcn_raw_buffer_load_b32(srd_r, vofs, so_e, 0);
        }
    };
    if (hoist) {
#pragma unroll
        for (int r = 0; r < RING - 1; ++r) issue_row(r);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int gy = ty0 + 4 * wv + r;
        const bool row_ok = !EDGE || gy < p.H;                                       // uniform per wave
        const int pix0 = (img * p.H + gy) * p.W + tx0;
        const int so = (pix0 * p.ldo + co0) * oes;
        if (hoist && r + RING - 1 < 4) issue_row(r + RING - 1);
        if (!row_ok) continue;
        // STATS 2, fp32 saved activation: two batches of 8 pixels per row, each batch's loads in flight together
#pragma unroll
        for (int eh = 0; eh < 2; ++eh) {
            float rv[8][NCO];
            if (hoist) {
#pragma unroll
                for (int e8 = 0; e8 < 8; ++e8)
#pragma unroll
                    for (int c = 0; c < NCO; ++c) {
                        const unsigned hw = (unsigned)rvh[r % RING][8 * eh + e8][c >> 1];
                        rv[e8][c] = __builtin_bit_cast(float, (c & 1) ? (hw & 0xffff0000u) : (hw << 16));
                    }
            } else if (UNET_CB_ABLATE & 2) {
#pragma unroll
                for (int e8 = 0; e8 < 8; ++e8)
#pragma unroll
                    for (int c = 0; c < NCO; ++c) rv[e8][c] = 1.f;
            } else if (STATS == 2) {
                const int sr = pix0 * p.bn_ldr * res;
#pragma unroll
                for (int e8 = 0; e8 < 8; ++e8) {
                    const int e = 8 * eh + e8, so_e = sr + ((e & 3) + 8 * (e >> 2)) * p.bn_ldr * res;
                    if (r16) {
                        hvec_t h;
                        if constexpr (NCO == 4) h = __builtin_amdgcn_raw_buffer_load_b64(srd_r, r_voff(e), so_e, 0);
                        else                    h[0] = __builtin_amdgcn_raw_buffer_load_b32(srd_r, r_voff(e), so_e, 0);
#pragma unroll
                        for (int c = 0; c < NCO; ++c)
                            rv[e8][c] = __builtin_bit_cast(float, (c & 1) ? ((unsigned)h[c >> 1] & 0xffff0000u) : ((unsigned)h[c >> 1] << 16));
                    } else {
                        ovec_t f;
                        if constexpr (NCO == 4) f = __builtin_amdgcn_raw_buffer_load_b128(srd_r, r_voff(e), so_e, 0);
                        else                    f = __builtin_amdgcn_raw_buffer_load_b64(srd_r, r_voff(e), so_e, 0);
#pragma unroll
                        for (int c = 0; c < NCO; ++c) rv[e8][c] = __builtin_bit_cast(float, (unsigned)f[c]);
                    }
                }
            }
#pragma unroll
            for (int e8 = 0; e8 < 8; ++e8) {
                const int e = 8 * eh + e8, ce = (e & 3) + 8 * (e >> 2);
                const bool colok = !EDGE || ce < wlim;
                float v[NCO];
#pragma unroll
                for (int c = 0; c < NCO; ++c) {
                    asm("v_accvgpr_read_b32 %0, %1" : "=v"(v[c]) : "a"(acc[r][c][e]));
                    // (an asm v_max: fmaxf() on a value the compiler cannot see through is preceded by a canonicalising v_max v, v, v)
                    if (STATS != 2) asm("v_max_f32 %0, %1, %2" : "=v"(v[c]) : "v"(v[c]), "v"(lo));
                    if (EDGE && STATS == 1 && colok) { st1[c] += v[c]; st2[c] += v[c] * v[c]; }
                    if (EDGE && STATS == 2 && colok) { st1[c] += v[c]; st2[c] += v[c] * rv[e8][c]; }
                }
                if (!EDGE && STATS != 0) {
                    // interior tiles: the sums of a channel PAIR as packed fp32 instructions, written out -- left to itself the compiler pairs the
                    // unconditional updates too, but across pixels, and keeps a row of values alive for it (scratch traffic inside the chunk loop)
#pragma unroll
                    for (int c = 0; c < NCO; c += 2) {
                        f32x2 vv = {v[c], v[c + 1]};
                        f32x2 ww = vv;
                        if (STATS == 2) { ww[0] = rv[e8][c]; ww[1] = rv[e8][c + 1]; }
                        asm("v_pk_add_f32 %0, %0, %1" : "+v"(st1p[c >> 1]) : "v"(vv));
                        asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(st2p[c >> 1]) : "v"(vv), "v"(ww));
                    }
                }
                if ((UNET_CB_ABLATE & 1) && v[0] != 1.2345e38f) continue;
                const int ovoff = colok ? vo + (so + ce * p.ldo * oes) : (int)0x80000000;
                if (out16) {
                    hvec_t h;
#pragma unroll
                    for (int c = 0; c < NCO; c += 2) h[c >> 1] = cb_pack2(v[c], v[c + 1]);
                    if constexpr (NCO == 4) __builtin_amdgcn_raw_buffer_store_b64(h, srd_o, ovoff, 0, UNET_NT_AUX(UNET_NT_CONV16));
                    else                    __builtin_amdgcn_raw_buffer_store_b32(h[0], srd_o, ovoff, 0, UNET_NT_AUX(UNET_NT_CONV16));
                } else {
                    ovec_t ov;
#pragma unroll
                    for (int c = 0; c < NCO; ++c) ov[c] = __builtin_bit_cast(unsigned, v[c]);
                    if constexpr (NCO == 4) __builtin_amdgcn_raw_buffer_store_b128(ov, srd_o, ovoff, 0, 0);
                    else                    __builtin_amdgcn_raw_buffer_store_b64(ov, srd_o, ovoff, 0, 0);
                }
            }
        }
    }
    if (STATS != 0) {
        // per-channel sums of this tile (of the fp32 values, before any rounding of the stored tensor): the two half-waves (same
        // channels, different pixels), then the four waves through LDS in a fixed order ->
        // stat_part[channel / 64][row = pixel tile][channel % 64][2]
#pragma unroll
        for (int c = 0; c < NCO; ++c) {
            if (!EDGE) { st1[c] = st1p[c >> 1][c & 1]; st2[c] = st2p[c >> 1][c & 1]; }
            st1[c] += __shfl_xor(st1[c], 32); st2[c] += __shfl_xor(st2[c], 32);
            if (lh == 0) { red[((wv * 32 + li) * NCO + c) * 2] = st1[c]; red[((wv * 32 + li) * NCO + c) * 2 + 1] = st2[c]; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // (not __syncthreads(): no vmcnt(0) -- a persistent caller has prefetches in flight)
        if (tid < CT) {                                               // tid = channel within the tile = NCO * lane + c
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int w4 = 0; w4 < 4; ++w4) { a += red[(w4 * CT + tid) * 2]; b += red[(w4 * CT + tid) * 2 + 1]; }
            const int ch = co0 + tid;
            float* o = p.stat_part + (((size_t)(ch >> 6) * p.n_px + row) * 64 + (ch & 63)) * 2;
            o[0] = a; o[1] = b;
        }
    }
}

// Accumulators of a tile start at the bias of their output channel (lane li, sub-tile c: channel co0 + NCO li + c) instead of at zero: the
// 256 (128) v_accvgpr_write of the fill are there anyway, and the epilogue loses as many v_add.
template <int NCO>
__device__ __forceinline__ void cb_init_acc(f32x16 (&acc)[4][NCO], const float (&bv)[NCO]) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < NCO; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                // (written as the instruction: a plain `acc = bv` is 256 VGPR copies to the register allocator, which spills them)
                float t;
                asm("v_accvgpr_write_b32 %0, %1" : "=a"(t) : "v"(bv[c]));
                acc[r][c][e] = t;
            }
}

// NORM: BatchNorm-apply on load.  The input tensor is the producer layer's conv output r (pre-BatchNorm), and the staging path forms
// the operand bf16(fma(scale[c], r, shift[c])) -- the very instructions unet_bn_apply_any (norm.hip) + this kernel's own staging
// conversion would have executed on a materialised BatchNorm output, so the result is bit-identical to that two-pass form while
// the y tensor is never written or read; positions outside the image stay exact zeros (the padding applies to the BatchNorm OUTPUT).
template <int NCO, int STATS, int NORM = 0>
__device__ __forceinline__ void conv_bf16_body(const ConvBf16Args& p) {
    constexpr int CT = 32 * NCO;
    constexpr int WB = 18 * CT * 16;                                // bytes of a chunk's weights for this tile
    constexpr int STAGE = kXP + WB;
    constexpr int NPIECE = WB / 1024, KW = (NPIECE + 3) / 4;
    __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;

    // tile: consecutive ids stay on one XCD (workgroups are dealt round-robin over the 8 XCDs), output-channel tile slowest,
    // so an XCD's L2 holds few weight tiles at a time
    int t = blockIdx.x;
    const int total = p.n_px * p.n_co;
    if ((total & 7) == 0) t = (t & 7) * (total >> 3) + (t >> 3);
    const int cot = t / p.n_px; int px = t % p.n_px;
    const int bx = px % p.tbx; px /= p.tbx;
    const int by = px % p.tby; const int img = px / p.tby;
    const int co0 = cot * CT, ty0 = 16 * by, tx0 = 32 * bx;
    const int nchunks = p.Cin / 16;

    // staging duty: float4 quad f of patch pixels (tid >> 2) + 64 j; out-of-image (and past-the-patch) lanes get an offset the
    // buffer's range check rejects, so they load zeros
    unsigned voff[10];
    int okmask = 0;                                                   // NORM: bit j = staged pixel j of this thread is inside the image
    const int f = tid & 3;
#pragma unroll
    for (int j = 0; j < 10; ++j) {
        const int pp = (tid >> 2) + 64 * j;
        const int gy = ty0 - 1 + pp / kPW, gx = tx0 - 1 + pp % kPW;
        const bool ok = pp < 18 * kPW && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        voff[j] = ok ? (unsigned)((((size_t)(img * p.H + gy) * p.W + gx) * p.ldx) * (p.in16 ? 2 : 4) + (p.in16 ? (f >> 1) * 16 : f * 16)) : 0x80000000u;
        okmask |= ok ? (1 << j) : 0;
    }
    // (compiler-visible buffer loads: it places the vmcnt wait in front of the first use; with inline-asm loads a register copy
    // of a destination can be scheduled ahead of a hand-written wait)
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_b*)smem;
    const unsigned wr_base = lds0 + (unsigned)((f >> 1) * kPlane + (tid >> 2) * 16 + (f & 1) * 8);
    const unsigned a_base = lds0 + (unsigned)(lh * kPlane + (4 * wv * kPW + li) * 16);
    // Sub-tile c of a lane is output channel NCO li + c: a lane's NCO accumulator sub-tiles are NCO consecutive channels and the
    // epilogue stores them with one instruction per pixel (16 / 8 bytes fp32, 8 / 4 bytes bf16).  The B fragment of sub-tile c
    // therefore reads column q = NCO j + c in lane j -- a 64- (32-) byte lane stride, 4-way bank conflicts on a linear image --
    // so column q is kept at slot cb_wslot(q), a permutation that puts every hardware lane group of a ds_read_b128 on 16 different slots
    // and costs the DMA nothing (its lanes still read within one contiguous 1-KB run).
    unsigned b_base[NCO];
#pragma unroll
    for (int c = 0; c < NCO; ++c) {
        const int q = NCO * li + c;
        b_base[c] = lds0 + (unsigned)(kXP + lh * CT * 16 + cb_wslot<NCO>(q) * 16);
    }
    // weight pieces wv + 4k of a chunk: piece = 64 columns x 16 B of one (tap, k half); [tap][half][Cout][8] in memory
    unsigned woff[KW];
#pragma unroll
    for (int k = 0; k < KW; ++k) {
        const int id = wv + 4 * k;
        const int row = id / (CT / 64), blk = id % (CT / 64);
        const int pos = 64 * blk + lane;                              // slot in the image -> the column stored there
        const int col = cb_wcol<NCO>(pos);
        woff[k] = (unsigned)((row * p.Cout + co0 + col) * 16);
    }
    const size_t wchunk = (size_t)18 * p.Cout * 16;
    const char* wsrc = reinterpret_cast<const char*>(p.wp);

    // Staging of the NEXT chunk rides inside the current chunk's MFMA stream (18 MFMA groups per chunk): the 10 input loads and
    // the weight DMAs two per group in groups 0-4, the 10 conversions + LDS writes two per group in groups 13-17 -- the compiler
    // waits with vmcnt(0) in front of the first conversion, so every load must be 8 groups (64+ MFMAs) old by then.  Issued in a
    // block around the stream they cost 23 % of a deep layer (ablation: 512->512 @64^2 compute alone 0.092 ms, data movement
    // alone 0.068 ms, together 0.122 ms).
    f32x4 stg[10];
    f32x4 nsc = {1.f, 1.f, 1.f, 1.f}, nsh = {0.f, 0.f, 0.f, 0.f};       // NORM: scale / shift of this thread's 4 channels of the chunk in flight
    auto issue_x1 = [&](int chunk, int j) {
        stg[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd, (int)voff[j], chunk * (p.in16 ? 32 : 64), 0));
        if (NORM && j == 9) {        // (the thread's channels of a chunk: 16 chunk + 4 f .. + 3, for either storage)
            nsc = *reinterpret_cast<const f32x4*>(p.in_scale + chunk * 16 + 4 * f);
            nsh = *reinterpret_cast<const f32x4*>(p.in_shift + chunk * 16 + 4 * f);
        }
    };
    auto issue_w1 = [&](int chunk, int stage, int k) {
        if (NPIECE % 4 == 0 || wv + 4 * k < NPIECE)
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(wsrc + (size_t)chunk * wchunk + woff[k]),
                                             (lds_void_b*)(smem + stage * STAGE + kXP + (wv + 4 * k) * 1024), 16, 0, 0);
    };
    auto write_x1 = [&](int stage, int j) {
        const unsigned wb = wr_base + (unsigned)(stage * STAGE + j * 1024);
        asm volatile("" : "+v"(stg[j]));          // pins every use of the loaded registers (the bf16 selects too) at this point of the stream
        uint2 v;
        if (NORM) {
            float r0, r1, r2, r3;
            if (p.in16) {
                const unsigned lo = __builtin_bit_cast(unsigned, (f & 1) ? stg[j].z : stg[j].x), hi = __builtin_bit_cast(unsigned, (f & 1) ? stg[j].w : stg[j].y);
                r0 = __builtin_bit_cast(float, lo << 16); r1 = __builtin_bit_cast(float, lo & 0xffff0000u);
                r2 = __builtin_bit_cast(float, hi << 16); r3 = __builtin_bit_cast(float, hi & 0xffff0000u);
            } else { r0 = stg[j].x; r1 = stg[j].y; r2 = stg[j].z; r3 = stg[j].w; }
            asm volatile("" : "+v"(nsc), "+v"(nsh));
            v.x = cb_pack2_pinned(fmaf(nsc.x, r0, nsh.x), fmaf(nsc.y, r1, nsh.y));
            v.y = cb_pack2_pinned(fmaf(nsc.z, r2, nsh.z), fmaf(nsc.w, r3, nsh.w));
            const unsigned m = (unsigned)(-((okmask >> j) & 1));          // zero padding of the BatchNorm OUTPUT
            v.x &= m; v.y &= m;
        } else {
            v.x = cb_pack2_pinned(stg[j].x, stg[j].y); v.y = cb_pack2_pinned(stg[j].z, stg[j].w);
            if (p.in16) {
                v.x = __builtin_bit_cast(unsigned, (f & 1) ? stg[j].z : stg[j].x); v.y = __builtin_bit_cast(unsigned, (f & 1) ? stg[j].w : stg[j].y);
            }
        }
        asm volatile("ds_write_b64 %0, %1" :: "v"(wb), "v"(v) : "memory");
    };
    auto issue_x = [&](int chunk) {
#pragma unroll
        for (int j = 0; j < 10; ++j) issue_x1(chunk, j);
    };
    auto issue_w = [&](int chunk, int stage) {
#pragma unroll
        for (int k = 0; k < KW; ++k) issue_w1(chunk, stage, k);
    };
    auto write_x = [&](int stage) {
#pragma unroll
        for (int j = 0; j < 10; ++j) write_x1(stage, j);
    };

    // diagnostic builds (scripts/build_variant.sh): bit 0 = no epilogue stores, bit 1 = no saved-activation loads in the STATS 2
    // epilogue, bit 2 = no chunk loop, bit 3 = no prologue loads
    f32x16 acc[4][NCO];
    {
        float bv[NCO];
#pragma unroll
        for (int c = 0; c < NCO; ++c) bv[c] = (STATS != 2 && p.bias) ? p.bias[co0 + NCO * li + c] : 0.f;
        cb_init_acc<NCO>(acc, bv);
    }
    if (!(UNET_CB_ABLATE & 8)) { issue_x(0); issue_w(0, 0); }
    write_x(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int c = 0; c < ((UNET_CB_ABLATE & 4) ? 0 : nchunks); c += 2) {                            // Cin % 32 == 0: an even number of chunks
        cb_compute<NCO, 0, STAGE>(acc, a_base, b_base, [&](int g) {
            if (g < 5) { issue_x1(c + 1, 2 * g); issue_x1(c + 1, 2 * g + 1); }
            if (2 * g < KW) issue_w1(c + 1, 1, 2 * g);
            if (2 * g + 1 < KW) issue_w1(c + 1, 1, 2 * g + 1);
            if (g >= 13) { write_x1(1, 2 * (g - 13)); write_x1(1, 2 * (g - 13) + 1); }
        });
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const int cn = c + 2 < nchunks ? c + 2 : c;                  // past the end: refill with a valid chunk, never read
        cb_compute<NCO, 1, STAGE>(acc, a_base, b_base, [&](int g) {
            if (g < 5) { issue_x1(cn, 2 * g); issue_x1(cn, 2 * g + 1); }
            if (2 * g < KW) issue_w1(cn, 0, 2 * g);
            if (2 * g + 1 < KW) issue_w1(cn, 0, 2 * g + 1);
            if (g >= 13) { write_x1(0, 2 * (g - 13)); write_x1(0, 2 * (g - 13) + 1); }
        });
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

    cb_epilogue<NCO, STATS>(p, acc, img, ty0, tx0, co0, t % p.n_px, reinterpret_cast<float*>(smem), tid, wv, li, lh);
}

// ---- persistent form for bf16-stored inputs ------------------------------------------------------------------------------------------
// The per-tile kernel above pays, per tile and serialised inside the workgroup, a prologue (first loads from HBM: 0.06 ms per launch on
// 64->64 @512^2), the chunk loop (0.10 ms: only 4 chunks with 64 reduce channels) and the epilogue (0.06 ms).  With the input stored as
// bf16 the staging needs no conversion, so the patch goes global -> LDS by LDS-DMA exactly like the weights (16 bytes per lane = the 8
// channels of one pixel and k half: one LDS slot of the [k half][pixel][8] image; lanes outside the image read a channel-indexed zero
// page), no staging registers, no conversion instructions.  The workgroup then walks tiles t, t + grid, ... and treats (tile, chunk) as
// ONE stream: while the last chunk of a tile is computed, chunk 0 of the NEXT tile is already on its way into the other LDS stage, so
// only a workgroup's first tile pays a prologue, and the epilogue's memory latency overlaps the landing of that chunk.
// Measured (same box, 8 x 512^2): 64->64 fwd+sums 0.223 -> 0.205 ms, 128->64 dgrad+sums 0.514 -> 0.438 ms, 128->128 @256^2 -7 %,
// the training step 16.15 -> 15.71 ms.
// A deeper variant was built and dropped: ONE workgroup per CU with FOUR stages (DMA three chunks ahead, counted vmcnt waits) ran
// 64->64 in 0.230 ms against 0.203 ms for this one.  Its ablations say why (ms per launch): DMA alone 0.102 (16-byte pieces at a
// 128-byte pixel stride read at 24 GB/s per CU, HBM and L2 weights together), MFMA stream + LDS reads alone 0.111, both 0.146,
// + epilogue arithmetic 0.208, + stores 0.230: with one workgroup per CU the epilogue (700 VALU instructions and 64 4-byte-per-lane
// stores per wave and tile) is serial with the matrix pipe, while a second resident workgroup hides most of it -- latency was not the
// limiter, so the extra stages bought nothing.
// Round 3 repeated the question for the 128-channel tiles (one workgroup per CU) with ablation builds (UNET_CBS_ABLATE) and one counter pass;
// 1024->512 @64^2 forward + sums, ms per launch: MFMA stream alone 0.162 (matrix pipe 78 % busy at 2.07 GHz), + epilogue 0.167, + DMA instead
// 0.209, everything 0.219 (66 % busy at 1.98 GHz); the DMA alone 0.099.  So the DMA costs 0.047 although it would fit under the MFMA stream twice.
// A THREE-stage patch ring (patch of chunk q + 2 and weights of chunk q + 1 issued during chunk q, counted vmcnt wait that leaves the
// patch pieces in flight, 136 KB of LDS) passed every test and ran the same 0.222 ms: the loss is not the latency of the patch DMA.  What is
// left unexplained sits between the weight DMA, the LDS it shares with 54 fragment reads per chunk and wave, and the clock (-4 % with DMA on).
// The 128-channel tile on EIGHT waves (waves 0-3 output channels 0-63, waves 4-7 channels 64-127, 128 accumulator registers each, patch and
// weights staged once for both: two waves per SIMD without the doubled patch traffic of two 64-channel workgroups) was built, passed the kernel
// tests and measured against this form on every layer: 64->128 @256^2 forward -10 %, 128->128 -3 %, 128->64 @512^2 data gradient -6 %, every layer
// with >= 256 reduce channels +0-2 %: about -0.06 ms per step for three more kernels -- removed.  It fits the counter finding below: these
// kernels are short of power, not of latency cover.
// Two more suspects cleared the same day: the patch pieces read from one CONTIGUOUS 20 KB block per chunk instead of 32 bytes per pixel row
// (UNET_CBS_ABLATE bit 32) cost the same, so it is not the scattered source; one DMA instruction per MFMA group instead of three in groups
// 0..4 is worth 1-3 % (kept: see fill), so it is not mainly back-to-back issue either.
__device__ __attribute__((aligned(256))) uint16_t g_zero_page_b[4096 + 64];       // zero source that out-of-image patch pixels walk over (per channel)

// EDGE = 0: H % 16 == 0 and W % 32 == 0 -- no tile crosses the image border, the epilogue's per-column selects are compiled out (cb_epilogue)
template <int NCO, int STATS, int EDGE = 1>
__device__ __forceinline__ void conv_bf16_stream_body(const ConvBf16Args& p) {
    constexpr int CT = 32 * NCO;
    constexpr int WB = 18 * CT * 16;
    constexpr int STAGE = kXP + WB;
    constexpr int NPIECE = WB / 1024, KW = (NPIECE + 3) / 4;
    constexpr int KX = 5;                                            // 2 planes x 10 pieces of 64 pixels = 20 x-pieces per chunk, 5 per wave
    __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE + 1024 * NCO];
    float* const red = reinterpret_cast<float*>(smem + 2 * STAGE);   // statistics scratch of the epilogue: never a DMA target
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int total = p.n_px * p.n_co;
    const int nchunks = p.Cin / 16;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_b*)smem;
    // x image of a stage: [patch pixel 0..639][2 x 16 B]: slot (h ^ bit 3 of the pixel index) holds k half h (see cb_compute, XL = 1)
    const int P0 = 4 * wv * kPW + li;
    const unsigned a_base = lds0 + (unsigned)(P0 * 32);
    unsigned long long asel[18];                                     // fragment g: the lanes that read the upper slot (wave-uniform: SGPR pairs)
#pragma unroll
    for (int g = 0; g < 18; ++g) asel[g] = __builtin_amdgcn_ballot_w64(((((P0 + (g % 6) * kPW + g / 6) >> 3) & 1) ^ lh) != 0);
    unsigned amask = 0;
#pragma unroll
    for (int g = 0; g < 18; ++g) amask |= (unsigned)((((P0 + (g % 6) * kPW + g / 6) >> 3) & 1) ^ lh) << g;
    unsigned b_base[NCO];
#pragma unroll
    for (int c = 0; c < NCO; ++c) {
        const int q = NCO * li + c;
        b_base[c] = lds0 + (unsigned)(kXP + lh * CT * 16 + cb_wslot<NCO>(q) * 16);
    }
    const size_t wchunk = (size_t)18 * p.Cout * 16;

    struct Tile { int t, img, ty0, tx0, co0; };
    auto tile_at = [&](int logical) {
        int t = logical;
        if ((total & 7) == 0) t = (t & 7) * (total >> 3) + (t >> 3);         // consecutive ids of one XCD (grid is a multiple of 8 or == total)
        Tile c; c.t = t;
        const int cot = t / p.n_px; int px = t % p.n_px;
        const int bx = px % p.tbx; px /= p.tbx;
        const int by = px % p.tby; c.img = px / p.tby;
        c.co0 = cot * CT; c.ty0 = 16 * by; c.tx0 = 32 * bx;
        return c;
    };
    // per-lane DMA sources of a tile: x-piece id = wv + 4 k covers patch pixels 32 id .. 32 id + 31, two lanes per pixel: the instruction
    // reads 32 x 32 contiguous bytes.  (16 bytes per lane from 64 different pixels -- one k half per instruction -- cost 0.03-0.07 ms
    // per launch more: four times the L2 requests for the same bytes.)
    struct Src { const char* x[KX]; const char* w; };
    auto sources = [&](const Tile& c) {
        Src s;
#pragma unroll
        for (int k = 0; k < KX; ++k) {
            const int id = wv + 4 * k, pp = 32 * id + (lane >> 1), h = (lane & 1) ^ ((pp >> 3) & 1);
            const int gy = c.ty0 - 1 + pp / kPW, gx = c.tx0 - 1 + pp % kPW;
            const bool ok = pp < 18 * kPW && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            s.x[k] = ok ? reinterpret_cast<const char*>(p.x) + (((size_t)(c.img * p.H + gy) * p.W + gx) * p.ldx) * 2 + h * 16
                        : reinterpret_cast<const char*>(g_zero_page_b) + h * 16;
            // (diagnostic: the same byte count from a CONTIGUOUS 20 KB block per tile and chunk -- wrong data, isolates the cost of the 32-byte pieces)
            if (UNET_CBS_ABLATE & 32) s.x[k] = reinterpret_cast<const char*>(p.x) + ((size_t)(c.t % p.n_px) * 20480 * (p.Cin / 16) + (size_t)id * 1024 + lane * 16) % ((size_t)p.x_bytes - 65536);
        }
        s.w = reinterpret_cast<const char*>(p.wp) + (size_t)c.co0 * 16;
        return s;
    };
    unsigned woff[KW];                                               // weight piece wv + 4 k: (row, 64-column block), swizzled column order (see the per-tile kernel)
#pragma unroll
    for (int k = 0; k < KW; ++k) {
        const int id = wv + 4 * k;
        const int row = id / (CT / 64), blk = id % (CT / 64);
        const int pos = 64 * blk + lane;
        const int col = cb_wcol<NCO>(pos);
        woff[k] = (unsigned)((row * p.Cout + col) * 16);
    }
    auto issue_x1 = [&](const Src& s, int chunk, int stage, int k) {
        const int id = wv + 4 * k;
        const char* src = s.x[k] + (size_t)chunk * 32;
        if (UNET_CBS_ABLATE & 32)
            src = reinterpret_cast<const char*>(p.x) + ((size_t)(s.x[k] - reinterpret_cast<const char*>(p.x)) + (size_t)chunk * 20480) % ((size_t)p.x_bytes - 65536);
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(src), (lds_void_b*)(smem + stage * STAGE + id * 1024), 16, 0, 0);
    };
    auto issue_w1 = [&](const Src& s, int chunk, int stage, int k) {
        if (NPIECE % 4 == 0 || wv + 4 * k < NPIECE)
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(s.w + (size_t)chunk * wchunk + woff[k]),
                                             (lds_void_b*)(smem + stage * STAGE + kXP + (wv + 4 * k) * 1024), 16, 0, 0);
    };
    auto fill = [&](const Src& s, int chunk, int stage, int g) {    // this wave's share of a chunk's DMA, spread over the MFMA groups
        if (UNET_CBS_ABLATE & 1) return;
        // ONE piece per MFMA group: a global_load_lds costs the issuing wave ~15 cycles behind an MFMA but ~64 directly behind another one (3.1);
        // three per group (patch + two weight pieces in groups 0..4) made the patch pieces 3.5 x as expensive as the weight pieces per instruction
        static_assert(KW + KX <= 18, "one DMA instruction per MFMA group");
        // (round 5, A/B: the patch pieces -- the first touch of their cache lines, ~1900 cycles from HBM -- in the chunk's FIRST groups instead
        // of its last: forward -1 .. -3 % on most layers, bott_b's data gradient +4 %: within noise of zero over the step, not kept)
        if (g < KW) { if (!(UNET_CBS_ABLATE & 16)) issue_w1(s, chunk, stage, g); }
        else if (g - KW < KX) { if (!(UNET_CBS_ABLATE & 8)) issue_x1(s, chunk, stage, g - KW); }
    };

    const int G = (int)gridDim.x;
    Tile cur = tile_at((int)blockIdx.x);
    // bias of a tile's output channels: loaded one tile ahead by asm statements (a builtin load would be sunk to its use, behind the epilogue)
    // and complete at the vmcnt(0) that closes the chunk it was issued in
    float bnx[NCO];
    auto load_bias = [&](const Tile& c) {
#pragma unroll
        for (int k = 0; k < NCO; ++k) bnx[k] = 0.f;
        if (STATS != 2 && p.bias) {
            const float* bp = p.bias + c.co0 + NCO * li;
#pragma unroll
            for (int k = 0; k < NCO; ++k) asm volatile("global_load_dword %0, %1, off offset:%2" : "=v"(bnx[k]) : "v"(bp), "n"(4 * k) : "memory");
        }
    };
    load_bias(cur);
    {
        const Src s0 = sources(cur);
#pragma unroll
        for (int g = 0; g < 18; ++g) fill(s0, 0, 0, g);             // prologue: chunk 0 of the first tile
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int logical = (int)blockIdx.x; logical < total; logical += G) {
        Tile nxt = cur;
        Src sf = sources(cur);                                       // what the DMA stream (one chunk ahead of the MFMAs) reads; recomputed per tile
                                                                     // rather than kept in registers across the epilogue
        f32x16 acc[4][NCO];
#pragma unroll
        for (int k = 0; k < NCO; ++k) asm volatile("" : "+v"(bnx[k]));      // (uses stay behind the wait that completed the loads)
        cb_init_acc<NCO>(acc, bnx);
        for (int c = 0; c < nchunks; c += 2) {                       // Cin % 32 == 0: an even number of chunks, chunk c lives in stage c & 1
            if (UNET_CBS_ABLATE & 2) { for (int g = 0; g < 18; ++g) fill(sf, c + 1, 1, g); }
            else cb_compute<NCO, 0, STAGE, 1>(acc, a_base, b_base, [&](int g) { fill(sf, c + 1, 1, g); }, asel, amask);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            int cn = c + 2;
            if (cn == nchunks) {                                     // the stream continues with chunk 0 of the workgroup's next tile
                cn = 0;                                              // (after the last tile: chunk 0 of this one again -- valid memory, never read)
                if (logical + G < total) { nxt = tile_at(logical + G); sf = sources(nxt); }
                load_bias(nxt);                                      // lands with this chunk's DMA (vmcnt(0) below)
            }
            if (UNET_CBS_ABLATE & 2) { for (int g = 0; g < 18; ++g) fill(sf, cn, 0, g); }
            else cb_compute<NCO, 1, STAGE, 1>(acc, a_base, b_base, [&](int g) { fill(sf, cn, 0, g); }, asel, amask);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        // (one epilogue per kernel: a uniform branch between an interior and an edge copy made the allocator stage the accumulators through scratch)
        if (!(UNET_CBS_ABLATE & 4)) cb_epilogue<NCO, STATS, 1, EDGE>(p, acc, cur.img, cur.ty0, cur.tx0, cur.co0, cur.t % p.n_px, red, tid, wv, li, lh);
        cur = nxt;
    }
}

// (the 64-channel-tile kernels are built for TWO workgroups per CU -- 128 + 128 registers, 76 KB of LDS each: they serve the
// full-resolution layers, which run on HBM, and a second workgroup's loads and stores fill the first one's epilogue: 64->64 @512^2
// 0.32 -> 0.27 ms in a same-box A/B)
__global__ __launch_bounds__(256, 1) void conv_bf16_kernel_128(ConvBf16Args p) { conv_bf16_body<4, 0>(p); }
__global__ __launch_bounds__(256, 2) void conv_bf16_kernel_64(ConvBf16Args p) { conv_bf16_body<2, 0>(p); }
__global__ __launch_bounds__(256, 1) void conv_bf16_stats_kernel_128(ConvBf16Args p) { conv_bf16_body<4, 1>(p); }
__global__ __launch_bounds__(256, 2) void conv_bf16_stats_kernel_64(ConvBf16Args p) { conv_bf16_body<2, 1>(p); }
__global__ __launch_bounds__(256, 1) void conv_bf16_bnbwd_kernel_128(ConvBf16Args p) { conv_bf16_body<4, 2>(p); }
__global__ __launch_bounds__(256, 2) void conv_bf16_bnbwd_kernel_64(ConvBf16Args p) { conv_bf16_body<2, 2>(p); }
// persistent forms (bf16-stored input, no BatchNorm-apply on load)
__global__ __launch_bounds__(256, 1) void conv_bf16_stream_kernel_128(ConvBf16Args p) { conv_bf16_stream_body<4, 0>(p); }
__global__ __launch_bounds__(256, 2) void conv_bf16_stream_kernel_64(ConvBf16Args p) { conv_bf16_stream_body<2, 0>(p); }
__global__ __launch_bounds__(256, 1) void conv_bf16_stream_stats_kernel_128(ConvBf16Args p) { conv_bf16_stream_body<4, 1>(p); }
__global__ __launch_bounds__(256, 2) void conv_bf16_stream_stats_kernel_64(ConvBf16Args p) { conv_bf16_stream_body<2, 1>(p); }
__global__ __launch_bounds__(256, 1) void conv_bf16_stream_bnbwd_kernel_128(ConvBf16Args p) { conv_bf16_stream_body<4, 2>(p); }
__global__ __launch_bounds__(256, 2) void conv_bf16_stream_bnbwd_kernel_64(ConvBf16Args p) { conv_bf16_stream_body<2, 2>(p); }
// ... and for images whose tiles all lie inside (every BASELINE shape)
__global__ __launch_bounds__(256, 1) void conv_bf16_stream_in_kernel_128(ConvBf16Args p) { conv_bf16_stream_body<4, 0, 0>(p); }
__global__ __launch_bounds__(256, 2) void conv_bf16_stream_in_kernel_64(ConvBf16Args p) { conv_bf16_stream_body<2, 0, 0>(p); }
__global__ __launch_bounds__(256, 1) void conv_bf16_stream_in_stats_kernel_128(ConvBf16Args p) { conv_bf16_stream_body<4, 1, 0>(p); }
__global__ __launch_bounds__(256, 2) void conv_bf16_stream_in_stats_kernel_64(ConvBf16Args p) { conv_bf16_stream_body<2, 1, 0>(p); }
__global__ __launch_bounds__(256, 1) void conv_bf16_stream_in_bnbwd_kernel_128(ConvBf16Args p) { conv_bf16_stream_body<4, 2, 0>(p); }
__global__ __launch_bounds__(256, 2) void conv_bf16_stream_in_bnbwd_kernel_64(ConvBf16Args p) { conv_bf16_stream_body<2, 2, 0>(p); }
// forward with BatchNorm-apply on load (NORM)
__global__ __launch_bounds__(256, 1) void conv_bf16_norm_kernel_128(ConvBf16Args p) { conv_bf16_body<4, 0, 1>(p); }
__global__ __launch_bounds__(256, 2) void conv_bf16_norm_kernel_64(ConvBf16Args p) { conv_bf16_body<2, 0, 1>(p); }
__global__ __launch_bounds__(256, 1) void conv_bf16_norm_stats_kernel_128(ConvBf16Args p) { conv_bf16_body<4, 1, 1>(p); }
__global__ __launch_bounds__(256, 2) void conv_bf16_norm_stats_kernel_64(ConvBf16Args p) { conv_bf16_body<2, 1, 1>(p); }

// fp32 HWIO weights -> bf16 [chunk][tap][k half][out channel][8]: mode 0 forward (reduce over Cin), mode 1 data gradient
// (reduce over Cout, taps flipped, output channel = the layer's input channel)
__global__ void conv_bf16_pack_kernel(const float* __restrict__ w, uint16_t* __restrict__ wp, int Cin, int Cout, int mode) {
    const int red = mode ? Cout : Cin, outc = mode ? Cin : Cout;
    const long n = (long)(red / 8) * 9 * outc;                       // one thread = 8 reduce channels of one (tap, output channel)
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int o = (int)(i % outc); long rest = i / outc;
    const int kh = (int)(rest % 2); rest /= 2;
    const int tap = (int)(rest % 9); const int chunk = (int)(rest / 9);
    const int k0 = chunk * 16 + kh * 8;
    unsigned v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float a, b;
        if (mode == 0) { a = w[((size_t)tap * Cin + k0 + 2 * e) * Cout + o]; b = w[((size_t)tap * Cin + k0 + 2 * e + 1) * Cout + o]; }
        else           { a = w[((size_t)(8 - tap) * Cin + o) * Cout + k0 + 2 * e]; b = w[((size_t)(8 - tap) * Cin + o) * Cout + k0 + 2 * e + 1]; }
        v[e] = cb_pack2(a, b);
    }
    reinterpret_cast<uint4*>(wp)[i] = make_uint4(v[0], v[1], v[2], v[3]);
}

int conv_bf16_cus() {
    static int cus = 0;
    if (!cus) { int dev = 0; hipDeviceProp_t pr; (void)hipGetDevice(&dev); cus = (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256; }
    return cus;
}


struct ConvBf16Stats { int mode; float* part; size_t bytes; const float* r; int ldr, c0, c1, r16; };

int run_conv_bf16(const float* x, int ldx, const void* wp, const float* bias, float* out, int ldo,
                  int N, int H, int W, int Cin, int Cout, int relu, hipStream_t st, const ConvBf16Stats* stats = nullptr, int in16 = 0,
                  int out16 = 0, const float* in_scale = nullptr, const float* in_shift = nullptr, int max_workgroups = 0) {
    ConvBf16Args a{};
    a.in_scale = in_scale; a.in_shift = in_shift;
    a.in16 = in16; a.out16 = out16; a.r16 = stats ? stats->r16 : 0;
    a.x = x; a.wp = (const uint16_t*)wp; a.bias = bias; a.out = out; a.ldx = ldx; a.ldo = ldo;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.relu = relu;
    a.tby = (H + 15) / 16; a.tbx = (W + 31) / 32; a.n_px = N * a.tby * a.tbx;
    a.x_bytes = (unsigned)((size_t)N * H * W * ldx * (in16 ? 2 : 4));
    // 128-channel tiles halve the input traffic; 64-channel tiles when they would leave compute units idle
    // (64-channel tiles for the K = 64 layers with 128 outputs -- two workgroups per CU hiding each other's epilogue -- were A/B-tested:
    // 64->128 @256^2 forward 0.111 -> 0.107 ms, 64->128 @512^2 data gradient + sums 0.401 -> 0.421 ms: no gain, wide tiles stay)
#ifndef UNET_CB_NARROW_MAX
#define UNET_CB_NARROW_MAX 0     /* diagnostic builds: 64-channel tiles (two workgroups per CU) for every layer with Cout <= this */
#endif
    const bool wide = Cout % 128 == 0 && (long)a.n_px * (Cout / 128) >= conv_bf16_cus() && Cout > UNET_CB_NARROW_MAX;
    a.n_co = Cout / (wide ? 128 : 64);
    const int mode = stats ? stats->mode : 0;
    if (mode) {
        if (stats->bytes < (size_t)(Cout / 64) * a.n_px * 128 * sizeof(float)) return UNET_ENOSPC;
        a.stat_part = stats->part; a.bn_r = stats->r; a.bn_ldr = stats->ldr; a.bn_c0 = stats->c0; a.bn_c1 = stats->c1;
    }
    const dim3 grid((unsigned)(a.n_px * a.n_co));
    if (in16 && out16 && (mode != 2 || a.r16) && !in_scale && Cin <= 4096) {
        // persistent form: one resident wave of workgroups walks the tiles (the grid is the whole tile count when that is smaller)
        const long slots = (long)unet_grid_slots(conv_bf16_cus(), max_workgroups) * (wide ? 1 : 2);
        const dim3 pgrid((unsigned)((long)grid.x < slots ? (long)grid.x : slots));
        if (H % 16 == 0 && W % 32 == 0) {
            if (mode == 0) { if (wide) conv_bf16_stream_in_kernel_128<<<pgrid, 256, 0, st>>>(a); else conv_bf16_stream_in_kernel_64<<<pgrid, 256, 0, st>>>(a); }
            else if (mode == 1) { if (wide) conv_bf16_stream_in_stats_kernel_128<<<pgrid, 256, 0, st>>>(a); else conv_bf16_stream_in_stats_kernel_64<<<pgrid, 256, 0, st>>>(a); }
            else { if (wide) conv_bf16_stream_in_bnbwd_kernel_128<<<pgrid, 256, 0, st>>>(a); else conv_bf16_stream_in_bnbwd_kernel_64<<<pgrid, 256, 0, st>>>(a); }
            return UNET_LAUNCH_STATUS();
        }
        if (mode == 0) { if (wide) conv_bf16_stream_kernel_128<<<pgrid, 256, 0, st>>>(a); else conv_bf16_stream_kernel_64<<<pgrid, 256, 0, st>>>(a); }
        else if (mode == 1) { if (wide) conv_bf16_stream_stats_kernel_128<<<pgrid, 256, 0, st>>>(a); else conv_bf16_stream_stats_kernel_64<<<pgrid, 256, 0, st>>>(a); }
        else { if (wide) conv_bf16_stream_bnbwd_kernel_128<<<pgrid, 256, 0, st>>>(a); else conv_bf16_stream_bnbwd_kernel_64<<<pgrid, 256, 0, st>>>(a); }
        return UNET_LAUNCH_STATUS();
    }
    if (in_scale && mode == 0) { if (wide) conv_bf16_norm_kernel_128<<<grid, 256, 0, st>>>(a); else conv_bf16_norm_kernel_64<<<grid, 256, 0, st>>>(a); }
    else if (in_scale && mode == 1) { if (wide) conv_bf16_norm_stats_kernel_128<<<grid, 256, 0, st>>>(a); else conv_bf16_norm_stats_kernel_64<<<grid, 256, 0, st>>>(a); }
    else if (in_scale) return UNET_EINVAL;
    else if (mode == 0) { if (wide) conv_bf16_kernel_128<<<grid, 256, 0, st>>>(a); else conv_bf16_kernel_64<<<grid, 256, 0, st>>>(a); }
    else if (mode == 1) { if (wide) conv_bf16_stats_kernel_128<<<grid, 256, 0, st>>>(a); else conv_bf16_stats_kernel_64<<<grid, 256, 0, st>>>(a); }
    else { if (wide) conv_bf16_bnbwd_kernel_128<<<grid, 256, 0, st>>>(a); else conv_bf16_bnbwd_kernel_64<<<grid, 256, 0, st>>>(a); }
    return UNET_LAUNCH_STATUS();
}

}  // namespace

extern "C" int unet_conv3x3_bf16_supported(int N, int H, int W, int Cin, int Cout) {
    return (N > 0 && H > 0 && W > 0 && Cin % 64 == 0 && Cout % 64 == 0 && (size_t)N * H * W * Cin * 4 < ((size_t)1 << 31)) ? 1 : 0;
}

extern "C" size_t unet_conv3x3_bf16_packed_bytes(int Cin, int Cout) { return (size_t)9 * Cin * Cout * 2; }

// mode 0: forward operand; mode 1: data-gradient operand (call unet_conv3x3_dgrad_bf16 with it)
extern "C" int unet_conv3x3_bf16_pack_weights(const float* w, void* packed, int Cin, int Cout, int mode, void* stream) {
    UNET_CHECK_ARG(w && packed && Cin % 64 == 0 && Cout % 64 == 0 && (mode == 0 || mode == 1) && unet_aligned16(packed));
    const long n = (long)9 * Cin * Cout / 8;
    conv_bf16_pack_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(w, (uint16_t*)packed, Cin, Cout, mode);
    return UNET_LAUNCH_STATUS();
}

// rows of the statistics partials of unet_conv3x3_fwd_bf16 / unet_conv3x3_dgrad_bf16 (one per 16 x 32 pixel tile)
extern "C" int unet_conv3x3_bf16_stats_rows(int N, int H, int W, int Cin, int Cout) {
    if (!unet_conv3x3_bf16_supported(N, H, W, Cin, Cout)) return 0;
    return N * ((H + 15) / 16) * ((W + 31) / 32);
}

// ---- weight gradient on the bf16 matrix cores ------------------------------------------------------------------------------
//   dw[a,b,ci,co] = sum_{n,y,x} bf16(xin[n,y+a-1,x+b-1,ci]) * bf16(dz[n,y,x,co]),   fp32 accumulation
// The contraction runs over pixels, so both MFMA operands need 8 consecutive PIXELS of one channel per lane while memory (and
// the LDS images, [32-channel group][pixel][32 channels] bf16) are channel-minor: the fragments are gathered by
// ds_read_b64_tr_b16 (gfx950's transposing LDS read, two per fragment), which also makes the nine taps free -- a tap is a
// row slot and a pixel offset in the read address, nothing is staged per tap.
// Workgroup = 64 input x 64 output channels x 9 taps (wave = one 32 x 32 pair, 9 accumulators) walking down a 32-pixel-wide
// strip of one image two rows per step: per step 2 new input rows (ring of 6) and 2 new dz rows are staged (fp32 buffer loads
// one step ahead, zeros outside the image through the buffer range check, converted to bf16 on the way into LDS) and every
// wave runs 36 MFMAs.  Strips x row chunks are split over workgroups; partial sums are added in a fixed order.
namespace {

typedef int i32x2 __attribute__((ext_vector_type(2)));

struct WgBf16Args {
    const float* x; const float* dz; float* out;
    int ldx, lddz, N, H, W, Cin, Cout;
    int n_ci, n_co, splits, tbx, rpc, cps, n_sc;
    unsigned x_bytes, dz_bytes;
    int x16, z16;                // xin / dz stored as bf16 (ldx / lddz in elements)
};

#ifndef UNET_WG_DEEP
#define UNET_WG_DEEP 1          /* DMA form: rows staged three steps ahead (0: two, the round-3 rings) */
#endif

#define WG_RDTR(dst, base, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=&v"(dst) : "v"(base), "n"(off))
#define WG_WAIT_CASE(n) case n: asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory"); break;
__device__ __forceinline__ void wg_wait_lgkm(int n) {
    switch (n) { WG_WAIT_CASE(0) WG_WAIT_CASE(1) WG_WAIT_CASE(2) WG_WAIT_CASE(3) WG_WAIT_CASE(4) WG_WAIT_CASE(5) WG_WAIT_CASE(6)
                 WG_WAIT_CASE(7) WG_WAIT_CASE(8) WG_WAIT_CASE(9) WG_WAIT_CASE(10) WG_WAIT_CASE(11) WG_WAIT_CASE(12)
                 default: asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
}

// counted lgkmcnt wait tied to the fragments about to be consumed (each register once)
#define WG_TIED_CASE4(n) case n: asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(a), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3)); break;
__device__ __forceinline__ void wg_wait_tied4(int n, bf16x8& a, bf16x8& b0, bf16x8& b1, bf16x8& b2, bf16x8& b3) {
    switch (n) { WG_TIED_CASE4(0) WG_TIED_CASE4(1) WG_TIED_CASE4(2) WG_TIED_CASE4(3) WG_TIED_CASE4(4) WG_TIED_CASE4(5) WG_TIED_CASE4(6) WG_TIED_CASE4(7)
                 default: asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(a), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3)); }
}
#define WG_TIED_CASE2(n) case n: asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(a), "+v"(b0), "+v"(b1)); break;
__device__ __forceinline__ void wg_wait_tied2(int n, bf16x8& a, bf16x8& b0, bf16x8& b1) {
    switch (n) { WG_TIED_CASE2(0) WG_TIED_CASE2(1) WG_TIED_CASE2(2) WG_TIED_CASE2(3) WG_TIED_CASE2(4) WG_TIED_CASE2(5) WG_TIED_CASE2(6) WG_TIED_CASE2(7)
                 default: asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(a), "+v"(b0), "+v"(b1)); }
}

// DMA = 1 (both operands stored as bf16): the staged rows are copies of memory, so they go global -> LDS by LDS-DMA (16 bytes per lane,
// five 1 KB pieces per row and wave; out-of-image pixels and rows read a zero page) -- no staging registers, no conversions, no ds_write
// (ablation of the register path on 512->512 @64^2, ms per launch: MFMA stream 0.096, staging loads 0.022, conversion + LDS writes 0.046
// -- nine ds_write_b64 per lane and step from ONE wave per SIMD run at a fraction of the LDS store rate -- barrier 0.005, fixed 0.035).
// The rings grow to 8 input rows / 6 dz rows and the DMA of pass s + 3 is issued at step s: two steps of latency cover instead of one.
// Round 6: ... and to 10 / 8 rows of 34 pixels (the DMA form needs no padding columns: 18 x 4352 B = 76.5 KB, still two workgroups per CU), the
// DMA of pass s + 4 issued at step s: THREE steps between a row's issue and its first use.  All channel pairs of a strip run on one XCD at the
// same time, so every workgroup meets every row's first touch of HBM (2-3 us under load) -- two steps (~2.3 us) did not cover it.
template <int DMA>
__device__ __forceinline__ void wgrad_bf16_body(const WgBf16Args& p) {
    constexpr int DEEP = DMA && UNET_WG_DEEP;
    constexpr int XR = DEEP ? 10 : DMA ? 8 : 6, ZR = DEEP ? 8 : DMA ? 6 : 4;      // ring rows: input / dz
    constexpr int ZP = ZR / 2;                                       // dz row pairs in the ring
    constexpr int AHEAD = DEEP ? 4 : 3;                              // step s issues the DMA of pass s + AHEAD
    constexpr int PIX = DEEP ? 34 : 36;                              // pixel slots of a staged row (34 input / 32 dz columns used)
    constexpr int kWgGrp = PIX * 64;                                 // one 32-channel group of a row
    constexpr int kWgXRow = 2 * kWgGrp, kWgDzRow = kWgXRow;          // bytes of one staged row: 2 channel groups x PIX pixels x 64 B
    constexpr int kXRingB = XR * kWgXRow;
    __shared__ __attribute__((aligned(1024))) char smem[(XR + ZR) * kWgXRow];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int npairs = p.n_ci * p.n_co;
    // workgroups are dealt round-robin over the 8 XCDs: keep all channel pairs of one pixel split on one XCD (and adjacent in
    // launch order), so that its L2 serves the 2..16-fold re-reads of the strip's rows instead of the fabric
    int pair, split;
    if ((p.splits & 7) == 0) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        pair = idx % npairs; split = xcd + 8 * (idx / npairs);
    } else {
        pair = blockIdx.x % npairs; split = blockIdx.x / npairs;
    }
    const int ci0 = (pair / p.n_co) * 64, co0 = (pair % p.n_co) * 64;
    const int cisub = wv & 1, cosub = wv >> 1;

    // staging roles (wave-uniform): waves 0,1 stage input row rho = wave, waves 2,3 the dz row rho = wave - 2; thread = 4 channels
    // (quad q) of 9 (input: patch columns 9o..9o+8) or 8 (dz) consecutive pixels
    const bool is_x = wv < 2;
    const int rho = wv & 1, o = (lane >> 4) & 3, q = lane & 15;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_b*)smem;
    const unsigned wr_lane = (unsigned)((q >> 3) * kWgGrp + 9 * o * 64 + (q & 7) * 8);
    // fragment gathers: 16-lane group g4 = (k half, channel half); lane 4 q4 + pp supplies pixel row q4, channels 4 pp .. 4 pp + 3
    const int g4 = lane >> 4, q4 = (lane >> 2) & 3, pp = lane & 3;
    const unsigned frag_lane = (unsigned)((8 * (g4 >> 1) + q4) * 64 + (16 * (g4 & 1) + 4 * pp) * 2);
    const unsigned a_lane = lds0 + cisub * kWgGrp + frag_lane;
    const unsigned b_lane = lds0 + kXRingB + cosub * kWgGrp + frag_lane;

    const int xes = p.x16 ? 2 : 4, zes = p.z16 ? 2 : 4;
    const char* xb_ptr = reinterpret_cast<const char*>(p.x) + (size_t)ci0 * xes;
    const char* zb_ptr = reinterpret_cast<const char*>(p.dz) + (size_t)co0 * zes;
    const int xrec = (int)(p.x_bytes - (unsigned)ci0 * xes), zrec = (int)(p.dz_bytes - (unsigned)co0 * zes);

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    // Staging passes.  Pass j of a strip chunk carries input rows y0-1+2j, y0+2j (ring slots 2j, 2j+1 mod 6) and, for j >= 1, dz
    // rows y0+2(j-1), +1 (dz slots 2((j-1)&1), +1); compute step s needs passes <= s+1: iteration s issues pass s+2, computes
    // step s and commits pass s+2.  The kernel is built for TWO workgroups per CU (8 taps in 128 AGPRs, the ninth in VGPRs, 112
    // VGPRs, 46 KB of LDS): a second workgroup -- or another stream's kernel: the BatchNorm passes and the 64-channel-tile data
    // gradient co-reside with it -- fills the load-latency and barrier gaps.  (The one-workgroup form ran its loads three passes
    // ahead in a register ring instead; alone it was as fast, inside the step 0.8 ms slower because nothing could share a CU.)
    f32x4 stg[1][9];
    unsigned voff[9];
    int y0 = 0, y_end = 0;
    // (both roles run the SAME instruction sequence -- 9 loads, 9 conversions and writes per pass, the dz role's ninth column is
    // padding -- and differ only in scalar operands: with role branches the compiler merged their tails and turned the register
    // ring into a dynamically indexed scratch array)
    const char* role_ptr = is_x ? xb_ptr : zb_ptr;
    const bool is16 = is_x ? p.x16 != 0 : p.z16 != 0;                // bf16-stored operand: same 16-byte loads, half of each used as it is
    const int role_es = is16 ? 2 : 4;
    const int role_rec = is_x ? xrec : zrec, role_rowbytes = p.W * (is_x ? p.ldx : p.lddz) * role_es;
    auto issue = [&](f32x4 (&sg)[9], int j) {
        const int row = is_x ? y0 - 1 + 2 * j + rho : y0 + 2 * (j - 1) + rho;
        const bool ok = is_x ? (row >= 0 && row < p.H) : (row >= y0 && row < y_end);
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)role_ptr, 0, ok ? role_rec : 0, 0x00020000);
        const int so = (row < 0 ? 0 : row) * role_rowbytes;
#pragma unroll
        for (int t = 0; t < 9; ++t) sg[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff[t], so, 0));
    };
    // convert and write pass j
    auto commit = [&](const f32x4 (&sg)[9], int j) {
        const int slot = is_x ? (2 * j + rho) % XR : XR + 2 * ((j + 1) & 1) + rho;     // dz slots follow the input-row slots
        const unsigned wb = lds0 + wr_lane + (unsigned)(slot * kWgXRow);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            uint2 v; v.x = cb_pack2_pinned(sg[t].x, sg[t].y); v.y = cb_pack2_pinned(sg[t].z, sg[t].w);
            if (is16) {          // 16-byte aligned load = 8 channels: this thread's quad is the low or the high half
                v.x = __builtin_bit_cast(unsigned, (q & 1) ? sg[t].z : sg[t].x); v.y = __builtin_bit_cast(unsigned, (q & 1) ? sg[t].w : sg[t].y);
            }
            asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(wb), "v"(v), "n"(t * 64) : "memory");
        }
    };
    // DMA form: a row slot (2 groups x PIX pixels x 64 B = 288 / 272 16-byte pieces) is filled by five wave-instructions; piece 64 k + lane =
    // (group, pixel, 16-byte quarter of the pixel's 32 channels) -- the LDS image is a copy of memory
    constexpr int NI = 5, NPIECE = 2 * PIX * 4;
    unsigned poff[NI]; unsigned pok = 0;                              // per-lane source offset inside the image row 0 / validity bits (per strip)
    int dgrp[NI], dpix[NI], dj[NI];
    if constexpr (DMA) {
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int P = 64 * k + lane;
            dgrp[k] = P / (4 * PIX); const int rem = P % (4 * PIX); dpix[k] = rem >> 2; dj[k] = rem & 3;
        }
    }
    // compute step s from LDS: 4 groups (output row yy, k-step ks) x 9 taps; A fragments three ahead, the next group's B at the group start
    auto compute = [&](int s) {
        unsigned xb[4], zb[2];
#pragma unroll
        for (int r = 0; r < 4; ++r) xb[r] = a_lane + (unsigned)(((2 * s + r) % XR) * kWgXRow);
        zb[0] = b_lane + (unsigned)(((s % ZP) * 2) * kWgDzRow); zb[1] = zb[0] + kWgDzRow;
        // One asm block per step, generated by scripts/gen_wgrad_bf16_step.py: 56 transposing reads, 36 MFMAs, counted waits (LDS operations
        // retire in order).  Input row r of the step serves tap a = r of output row 0 AND tap a = r - 1 of output row 1, so each of the 24
        // A fragments (4 rows x 3 column shifts x 2 k-steps) is read ONCE and feeds both MFMAs (round 2 read 36: one per MFMA -- 1 KB of
        // fragment reads per MFMA and wave is the LDS's whole rate at a busy matrix pipe); both dz rows of a k-step are live together.
        // The fragments live in fixed registers v[80:111] -- an A ring of four, the four B fragments of the step -- because a 128-bit
        // MFMA operand has to be assembled from two 64-bit reads: as separate asm statements that took compiler copies plus s_nop
        // padding in front of every MFMA (57 cycles per MFMA measured); inside one block nothing can be scheduled between a read and its use.
        // Measured (round 3, same box, 13 layers): the shared A fragments are worth 1-3 %; placing the five DMA instructions of pass s + 3
        // INSIDE this block (one per seven MFMAs, `--dma` of the generator) instead of in front of it changed nothing (2.128 vs 2.128 ms,
        // 2.150 vs 2.142 ms over the layers) although the no-DMA ablation is 20 % faster: what the ablation removes is the WAIT for the
        // rows (vmcnt), i.e. the L2 / fabric delivery rate of a 64 x 64 tile's 288 flop per staged byte, not the issue slots.
        asm volatile(
            "ds_read_b64_tr_b16 v[96:97], %[z0] offset:0\n\t"
            "ds_read_b64_tr_b16 v[98:99], %[z0] offset:256\n\t"
            "ds_read_b64_tr_b16 v[80:81], %[x0] offset:0\n\t"
            "ds_read_b64_tr_b16 v[82:83], %[x0] offset:256\n\t"
            "ds_read_b64_tr_b16 v[84:85], %[x0] offset:64\n\t"
            "ds_read_b64_tr_b16 v[86:87], %[x0] offset:320\n\t"
            "ds_read_b64_tr_b16 v[88:89], %[x0] offset:128\n\t"
            "ds_read_b64_tr_b16 v[90:91], %[x0] offset:384\n\t"
            "s_waitcnt lgkmcnt(6)\n\t"
            "s_waitcnt lgkmcnt(4)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c0], v[80:83], v[96:99], %[c0]\n\t"
            "ds_read_b64_tr_b16 v[100:101], %[z1] offset:0\n\t"
            "ds_read_b64_tr_b16 v[102:103], %[z1] offset:256\n\t"
            "ds_read_b64_tr_b16 v[92:93], %[x1] offset:0\n\t"
            "ds_read_b64_tr_b16 v[94:95], %[x1] offset:256\n\t"
            "s_waitcnt lgkmcnt(6)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c1], v[84:87], v[96:99], %[c1]\n\t"
            "ds_read_b64_tr_b16 v[80:81], %[x1] offset:64\n\t"
            "ds_read_b64_tr_b16 v[82:83], %[x1] offset:320\n\t"
            "s_waitcnt lgkmcnt(6)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c2], v[88:91], v[96:99], %[c2]\n\t"
            "ds_read_b64_tr_b16 v[84:85], %[x1] offset:128\n\t"
            "ds_read_b64_tr_b16 v[86:87], %[x1] offset:384\n\t"
            "s_waitcnt lgkmcnt(4)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c3], v[92:95], v[96:99], %[c3]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c0], v[92:95], v[100:103], %[c0]\n\t"
            "ds_read_b64_tr_b16 v[88:89], %[x2] offset:0\n\t"
            "ds_read_b64_tr_b16 v[90:91], %[x2] offset:256\n\t"
            "s_waitcnt lgkmcnt(4)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c4], v[80:83], v[96:99], %[c4]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c1], v[80:83], v[100:103], %[c1]\n\t"
            "ds_read_b64_tr_b16 v[92:93], %[x2] offset:64\n\t"
            "ds_read_b64_tr_b16 v[94:95], %[x2] offset:320\n\t"
            "s_waitcnt lgkmcnt(4)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c5], v[84:87], v[96:99], %[c5]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c2], v[84:87], v[100:103], %[c2]\n\t"
            "ds_read_b64_tr_b16 v[80:81], %[x2] offset:128\n\t"
            "ds_read_b64_tr_b16 v[82:83], %[x2] offset:384\n\t"
            "s_waitcnt lgkmcnt(4)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c6], v[88:91], v[96:99], %[c6]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c3], v[88:91], v[100:103], %[c3]\n\t"
            "ds_read_b64_tr_b16 v[84:85], %[x3] offset:0\n\t"
            "ds_read_b64_tr_b16 v[86:87], %[x3] offset:256\n\t"
            "s_waitcnt lgkmcnt(4)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c7], v[92:95], v[96:99], %[c7]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c4], v[92:95], v[100:103], %[c4]\n\t"
            "ds_read_b64_tr_b16 v[88:89], %[x3] offset:64\n\t"
            "ds_read_b64_tr_b16 v[90:91], %[x3] offset:320\n\t"
            "s_waitcnt lgkmcnt(4)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c8], v[80:83], v[96:99], %[c8]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c5], v[80:83], v[100:103], %[c5]\n\t"
            "ds_read_b64_tr_b16 v[92:93], %[x3] offset:128\n\t"
            "ds_read_b64_tr_b16 v[94:95], %[x3] offset:384\n\t"
            "s_waitcnt lgkmcnt(4)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c6], v[84:87], v[100:103], %[c6]\n\t"
            "ds_read_b64_tr_b16 v[104:105], %[z0] offset:1024\n\t"
            "ds_read_b64_tr_b16 v[106:107], %[z0] offset:1280\n\t"
            "ds_read_b64_tr_b16 v[80:81], %[x0] offset:1024\n\t"
            "ds_read_b64_tr_b16 v[82:83], %[x0] offset:1280\n\t"
            "s_waitcnt lgkmcnt(6)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c7], v[88:91], v[100:103], %[c7]\n\t"
            "ds_read_b64_tr_b16 v[84:85], %[x0] offset:1088\n\t"
            "ds_read_b64_tr_b16 v[86:87], %[x0] offset:1344\n\t"
            "s_waitcnt lgkmcnt(6)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c8], v[92:95], v[100:103], %[c8]\n\t"
            "ds_read_b64_tr_b16 v[88:89], %[x0] offset:1152\n\t"
            "ds_read_b64_tr_b16 v[90:91], %[x0] offset:1408\n\t"
            "s_waitcnt lgkmcnt(6)\n\t"
            "s_waitcnt lgkmcnt(4)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c0], v[80:83], v[104:107], %[c0]\n\t"
            "ds_read_b64_tr_b16 v[108:109], %[z1] offset:1024\n\t"
            "ds_read_b64_tr_b16 v[110:111], %[z1] offset:1280\n\t"
            "ds_read_b64_tr_b16 v[92:93], %[x1] offset:1024\n\t"
            "ds_read_b64_tr_b16 v[94:95], %[x1] offset:1280\n\t"
            "s_waitcnt lgkmcnt(6)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c1], v[84:87], v[104:107], %[c1]\n\t"
            "ds_read_b64_tr_b16 v[80:81], %[x1] offset:1088\n\t"
            "ds_read_b64_tr_b16 v[82:83], %[x1] offset:1344\n\t"
            "s_waitcnt lgkmcnt(6)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c2], v[88:91], v[104:107], %[c2]\n\t"
            "ds_read_b64_tr_b16 v[84:85], %[x1] offset:1152\n\t"
            "ds_read_b64_tr_b16 v[86:87], %[x1] offset:1408\n\t"
            "s_waitcnt lgkmcnt(4)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c3], v[92:95], v[104:107], %[c3]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c0], v[92:95], v[108:111], %[c0]\n\t"
            "ds_read_b64_tr_b16 v[88:89], %[x2] offset:1024\n\t"
            "ds_read_b64_tr_b16 v[90:91], %[x2] offset:1280\n\t"
            "s_waitcnt lgkmcnt(4)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c4], v[80:83], v[104:107], %[c4]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c1], v[80:83], v[108:111], %[c1]\n\t"
            "ds_read_b64_tr_b16 v[92:93], %[x2] offset:1088\n\t"
            "ds_read_b64_tr_b16 v[94:95], %[x2] offset:1344\n\t"
            "s_waitcnt lgkmcnt(4)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c5], v[84:87], v[104:107], %[c5]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c2], v[84:87], v[108:111], %[c2]\n\t"
            "ds_read_b64_tr_b16 v[80:81], %[x2] offset:1152\n\t"
            "ds_read_b64_tr_b16 v[82:83], %[x2] offset:1408\n\t"
            "s_waitcnt lgkmcnt(4)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c6], v[88:91], v[104:107], %[c6]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c3], v[88:91], v[108:111], %[c3]\n\t"
            "ds_read_b64_tr_b16 v[84:85], %[x3] offset:1024\n\t"
            "ds_read_b64_tr_b16 v[86:87], %[x3] offset:1280\n\t"
            "s_waitcnt lgkmcnt(4)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c7], v[92:95], v[104:107], %[c7]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c4], v[92:95], v[108:111], %[c4]\n\t"
            "ds_read_b64_tr_b16 v[88:89], %[x3] offset:1088\n\t"
            "ds_read_b64_tr_b16 v[90:91], %[x3] offset:1344\n\t"
            "s_waitcnt lgkmcnt(4)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c8], v[80:83], v[104:107], %[c8]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c5], v[80:83], v[108:111], %[c5]\n\t"
            "ds_read_b64_tr_b16 v[92:93], %[x3] offset:1152\n\t"
            "ds_read_b64_tr_b16 v[94:95], %[x3] offset:1408\n\t"
            "s_waitcnt lgkmcnt(4)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c6], v[84:87], v[108:111], %[c6]\n\t"
            "s_waitcnt lgkmcnt(2)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c7], v[88:91], v[108:111], %[c7]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c8], v[92:95], v[108:111], %[c8]\n\t"
            : [c0] "+a"(acc[0]), [c1] "+a"(acc[1]), [c2] "+a"(acc[2]), [c3] "+a"(acc[3]), [c4] "+a"(acc[4]), [c5] "+a"(acc[5]), [c6] "+a"(acc[6]), [c7] "+a"(acc[7]),
              [c8] "+v"(acc[8])
            : [x0] "v"(xb[0]), [x1] "v"(xb[1]), [x2] "v"(xb[2]), [x3] "v"(xb[3]), [z0] "v"(zb[0]), [z1] "v"(zb[1])
            : "memory", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99",
              "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111");
    };

    auto dma_issue = [&](int j) {
        const int row = is_x ? y0 - 1 + 2 * j + rho : y0 + 2 * (j - 1) + rho;
        const bool rok = is_x ? (row >= 0 && row < p.H) : (row >= y0 && row < y_end);
        const int slot = is_x ? (2 * j + rho) % XR : XR + 2 * ((j + ZP - 1) % ZP) + rho;  // dz rows of step j - 1: slots 2 ((j - 1) % ZP), + 1
        const char* rb = role_ptr + (size_t)(rok ? row : 0) * role_rowbytes;
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const char* src = (rok && ((pok >> k) & 1)) ? rb + poff[k] : reinterpret_cast<const char*>(g_zero_page_b) + dj[k] * 16;
            if (k < 4 || lane < NPIECE - 256)                         // pieces NPIECE .. 319 would land in the next slot
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(src), (lds_void_b*)(smem + slot * kWgXRow + 1024 * k), 16, 0, 0);
        }
    };

    for (int sc = split; sc < p.n_sc; sc += p.splits) {
        const int strip = sc / p.cps, chunk = sc % p.cps;
        const int img = strip / p.tbx, x0 = 32 * (strip % p.tbx);
        y0 = chunk * p.rpc;
        y_end = y0 + p.rpc < p.H ? y0 + p.rpc : p.H;
        const int steps = (y_end - y0 + 1) / 2;
        if constexpr (DMA) {
            pok = 0;
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                const int gx = is_x ? x0 - 1 + dpix[k] : x0 + dpix[k];
                const bool ok = 64 * k + lane < NPIECE && dpix[k] < (is_x ? 34 : 32) && gx >= 0 && gx < p.W;
                poff[k] = ok ? (unsigned)((((size_t)img * p.H * p.W + gx) * (is_x ? p.ldx : p.lddz)) * 2 + dgrp[k] * 64 + dj[k] * 16) : 0u;
                pok |= ok ? (1u << k) : 0u;
            }
            // fill: passes 0, 1 landed, passes 2 .. AHEAD - 1 in flight.  (The previous strip ended with a barrier and drained its DMAs.)
#pragma unroll
            for (int j = 0; j < AHEAD; ++j) dma_issue(j);
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"((AHEAD - 2) * NI) : "memory");
            for (int s = 0; s < steps; ++s) {
                if (!(UNET_CB_ABLATE & 2)) dma_issue(s + AHEAD);
                if (!(UNET_CB_ABLATE & 1)) compute(s);
                // pass s + 2 has to be here for step s + 1; the passes behind it (one, or two in the deep form) stay in flight
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"((AHEAD - 2) * NI) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // DMAs still in flight target slots the next strip fills
            continue;
        }
        // per-thread offsets inside a row (the row goes into the scalar offset); columns outside the image / the patch: rejected
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int pc = 9 * o + t;                                 // staged column: input patch column (image column x0 - 1 + pc) / dz column x0 + pc
            const int gx = is_x ? x0 - 1 + pc : x0 + pc;
            const bool ok = pc < (is_x ? 34 : 32) && gx >= 0 && gx < p.W;
            voff[t] = ok ? (unsigned)((((size_t)img * p.H * p.W + gx) * (is_x ? p.ldx : p.lddz)) * role_es + (is16 ? (q >> 1) * 16 : q * 16)) : 0x80000000u;
        }
        // fill: passes 0 and 1 committed, 2 and 3 in flight.  (The previous strip's last step ended with a barrier: every wave is
        // done reading the rings.)
        issue(stg[0], 0); commit(stg[0], 0);
        issue(stg[0], 1); commit(stg[0], 1);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        for (int s = 0; s < steps; ++s) {
            if (!(UNET_CB_ABLATE & 2)) issue(stg[0], s + 2);
            if (!(UNET_CB_ABLATE & 1)) compute(s);
            if (!(UNET_CB_ABLATE & 4)) commit(stg[0], s + 2);
            if (!(UNET_CB_ABLATE & 8)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // passes still in flight belong to nobody
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

    // accumulator register e of tap t = (input channel ci0 + 32 cisub + (e&3) + 8 (e>>2) + 4 lh, output channel co0 + 32 cosub + li)
    float* o_base = p.out + (size_t)split * 9 * p.Cin * p.Cout;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int ci = ci0 + 32 * cisub + (e & 3) + 8 * (e >> 2) + 4 * lh;
            float v;
            if (t < 8) asm("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(acc[t][e]));
            else v = acc[t][e];                       // the ninth tap accumulates in VGPRs: 8 x 16 AGPRs + 128 VGPRs = two workgroups per CU
            o_base[((size_t)t * p.Cin + ci) * p.Cout + co0 + 32 * cosub + li] = v;
        }
}

__global__ __launch_bounds__(256, 2) void wgrad_bf16_kernel(WgBf16Args p) { wgrad_bf16_body<0>(p); }
__global__ __launch_bounds__(256, 2) void wgrad_bf16_dma_kernel(WgBf16Args p) { wgrad_bf16_body<1>(p); }

__global__ __launch_bounds__(256) void wgrad_bf16_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, long n4, int splits, int sl) {
    __shared__ f32x4 part[256];
    const int per = 256 / sl;
    const int o = threadIdx.x % per, sj = threadIdx.x / per;
    const long i = (long)blockIdx.x * per + o;
    const int k0 = (int)((long)splits * sj / sl), k1 = (int)((long)splits * (sj + 1) / sl);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i < n4)
        for (int k = k0; k < k1; ++k) s += reinterpret_cast<const f32x4*>(ws)[(size_t)k * n4 + i];
    if (sl == 1) { if (i < n4) reinterpret_cast<f32x4*>(dw)[i] = s; return; }
    part[threadIdx.x] = s;
    __syncthreads();
    if (sj == 0 && i < n4) {
        for (int j = 1; j < sl; ++j) s += part[j * per + o];
        reinterpret_cast<f32x4*>(dw)[i] = s;
    }
}

void wgrad_bf16_plan(WgBf16Args& a, int max_workgroups) {
    a.n_ci = a.Cin / 64; a.n_co = a.Cout / 64;
    const int npairs = a.n_ci * a.n_co;
    a.tbx = (a.W + 31) / 32;
    const int strips = a.N * a.tbx;
    const int slots = unet_grid_slots(conv_bf16_cus(), max_workgroups);  // one workgroup per CU (two: 4-12 % faster alone, nothing in the step -- DESIGN.md 3b)
    const int target = (slots + npairs - 1) / npairs;                    // strip-chunks wanted so that every slot has a workgroup
    int cps = (target + strips - 1) / strips; if (cps < 1) cps = 1;
    int rpc = (a.H + cps - 1) / cps; rpc += rpc & 1; if (rpc < 2) rpc = 2;
    a.rpc = rpc; a.cps = (a.H + rpc - 1) / rpc; a.n_sc = strips * a.cps;
    int splits = slots / npairs; if (splits < 1) splits = 1; if (splits > a.n_sc) splits = a.n_sc;
    a.splits = splits;
}

}  // namespace

extern "C" int unet_conv3x3_wgrad_bf16_supported(int N, int H, int W, int Cin, int Cout) {
    return (N > 0 && H > 0 && W > 0 && Cin % 64 == 0 && Cout % 64 == 0 && (size_t)N * H * W * (Cin > Cout ? Cin : Cout) * 4 < ((size_t)1 << 31)) ? 1 : 0;
}

extern "C" size_t unet_conv3x3_wgrad_bf16_workspace_wg(int N, int H, int W, int Cin, int Cout, int max_workgroups) {
    if (!unet_conv3x3_wgrad_bf16_supported(N, H, W, Cin, Cout)) return 0;
    WgBf16Args a{}; a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    wgrad_bf16_plan(a, max_workgroups);
    return a.splits > 1 ? (size_t)a.splits * 9 * Cin * Cout * sizeof(float) : 16;
}
extern "C" size_t unet_conv3x3_wgrad_bf16_workspace(int N, int H, int W, int Cin, int Cout) {
    return unet_conv3x3_wgrad_bf16_workspace_wg(N, H, W, Cin, Cout, 0);
}

// dw[a,b,ci,co] (HWIO, UNet/model.py:31) = sum_{n,y,x} bf16(xin[n,y+a-1,x+b-1,ci]) * bf16(dz[n,y,x,co]); x_bf16 / dz_bf16: that
// operand is already stored as bf16 (leading dimension in elements).  max_workgroups: cap on the one-wave grid (common.h
// unet_grid_slots); the workspace follows it (.._workspace_wg)
extern "C" int unet_conv3x3_wgrad_bf16_wg(const void* xin, int ldx, int x_bf16, const void* dz, int lddz, int dz_bf16, float* dw,
                                          int N, int H, int W, int Cin, int Cout, int max_workgroups, void* ws, size_t ws_bytes, void* stream) {
    UNET_CHECK_ARG(xin && dz && dw && ws && unet_conv3x3_wgrad_bf16_supported(N, H, W, Cin, Cout));
    UNET_CHECK_ARG((!x_bf16 || ldx % 8 == 0) && (!dz_bf16 || lddz % 8 == 0));        // 16-byte aligned pixels
    UNET_CHECK_ARG(ldx >= Cin && lddz >= Cout && ldx % 4 == 0 && lddz % 4 == 0 && unet_aligned16(xin) && unet_aligned16(dz) && unet_aligned16(dw) && unet_aligned16(ws));
    UNET_CHECK_ARG((size_t)N * H * W * ldx * 4 < ((size_t)1 << 31) && (size_t)N * H * W * lddz * 4 < ((size_t)1 << 31));
    if (ws_bytes < unet_conv3x3_wgrad_bf16_workspace_wg(N, H, W, Cin, Cout, max_workgroups)) return UNET_ENOSPC;
    WgBf16Args a{};
    a.x = (const float*)xin; a.dz = (const float*)dz; a.ldx = ldx; a.lddz = lddz; a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.x16 = x_bf16 ? 1 : 0; a.z16 = dz_bf16 ? 1 : 0;
    wgrad_bf16_plan(a, max_workgroups);
    a.out = a.splits > 1 ? (float*)ws : dw;
    a.x_bytes = (unsigned)((size_t)N * H * W * ldx * (a.x16 ? 2 : 4)); a.dz_bytes = (unsigned)((size_t)N * H * W * lddz * (a.z16 ? 2 : 4));
    hipStream_t st = (hipStream_t)stream;
    // both operands stored as bf16: the staged rows are plain copies of memory -> LDS-DMA staging; otherwise the register-staged
    // kernel converts fp32 operands on the way into LDS
#ifdef UNET_WGRAD_BF16_NO_DMA    /* diagnostic build (scripts/bf16_onload_ab.py): what a weight gradient with BatchNorm-apply on load would at least cost */
    if (false) {}
#else
    if (a.x16 && a.z16) wgrad_bf16_dma_kernel<<<(unsigned)(a.n_ci * a.n_co * a.splits), 256, 0, st>>>(a);
#endif
    else                       wgrad_bf16_kernel<<<(unsigned)(a.n_ci * a.n_co * a.splits), 256, 0, st>>>(a);
    int rc = UNET_LAUNCH_STATUS(); if (rc) return rc;
    if (a.splits > 1) {
        const long n4 = (long)9 * Cin * Cout / 4;
        int sl = 1;
        while (sl < 16 && 2 * sl <= a.splits && n4 * sl < 256 * 1024) sl *= 2;
        wgrad_bf16_reduce_kernel<<<(unsigned)((n4 * sl + 255) / 256), 256, 0, st>>>((const float*)ws, dw, n4, a.splits, sl);
        rc = UNET_LAUNCH_STATUS();
    }
    return rc;
}

extern "C" int unet_conv3x3_wgrad_bf16(const void* xin, int ldx, int x_bf16, const void* dz, int lddz, int dz_bf16, float* dw,
                                          int N, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream) {
    return unet_conv3x3_wgrad_bf16_wg(xin, ldx, x_bf16, dz, lddz, dz_bf16, dw, N, H, W, Cin, Cout, 0, ws, ws_bytes, stream);
}

// out[n,i,j,co] = relu?(bias[co] + sum_{a,b,ci} bf16(x[n,i+a-1,j+b-1,ci]) * bf16(W[a,b,ci,co])), fp32 accumulation.
// x_bf16: the input is stored as bf16 (ldx in elements); in_scale / in_shift (nullable, Cin floats each, 16-byte aligned): BatchNorm
// apply on load -- x is the producer's conv output r and the operand is bf16(scale * r + shift) inside the image, 0 outside;
// out_bf16: the output is stored as bf16; stat_part nullable: BatchNorm sums of the output, [Cout/64][rows][64][2] = (sum y, sum y^2)
// per 16 x 32 pixel tile, for unet_bn_train_finalize_partials
// max_workgroups: cap on the persistent grid (common.h unet_grid_slots; the statistics rows are per tile and do not change)
extern "C" int unet_conv3x3_fwd_bf16_wg(const void* x, int ldx, int x_bf16, const float* in_scale, const float* in_shift, const void* wp,
                                        const float* bias, void* out, int ldo, int out_bf16,
                                        int N, int H, int W, int Cin, int Cout, int relu, float* stat_part, size_t stat_bytes, int max_workgroups, void* stream) {
    UNET_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr) && (!in_scale || (unet_aligned16(in_scale) && unet_aligned16(in_shift))));
    UNET_CHECK_ARG(x && wp && out && unet_conv3x3_bf16_supported(N, H, W, Cin, Cout));
    UNET_CHECK_ARG(ldx >= Cin && ldo >= Cout && ldx % 4 == 0 && unet_aligned16(x) && unet_aligned16(wp));
    UNET_CHECK_ARG((size_t)N * H * W * ldx * 4 < ((size_t)1 << 31) && (size_t)N * H * W * ldo * 4 < ((size_t)1 << 31));
    UNET_CHECK_ARG(!x_bf16 || ldx % 8 == 0);
    const ConvBf16Stats s{1, stat_part, stat_bytes, nullptr, 0, 0, 0, 0};
    return run_conv_bf16((const float*)x, ldx, wp, bias, (float*)out, ldo, N, H, W, Cin, Cout, relu, (hipStream_t)stream, stat_part ? &s : nullptr,
                         x_bf16 ? 1 : 0, out_bf16 ? 1 : 0, in_scale, in_shift, max_workgroups);
}
extern "C" int unet_conv3x3_fwd_bf16(const void* x, int ldx, int x_bf16, const float* in_scale, const float* in_shift, const void* wp,
                                     const float* bias, void* out, int ldo, int out_bf16,
                                     int N, int H, int W, int Cin, int Cout, int relu, float* stat_part, size_t stat_bytes, void* stream) {
    return unet_conv3x3_fwd_bf16_wg(x, ldx, x_bf16, in_scale, in_shift, wp, bias, out, ldo, out_bf16, N, H, W, Cin, Cout, relu, stat_part, stat_bytes, 0, stream);
}

// data gradient with every option: dz_bf16 (dz stored as bf16), r_prev / stat_part nullable (producer's BatchNorm-backward sums)
extern "C" int unet_conv3x3_dgrad_bf16_wg(const void* dz, int lddz, int dz_bf16, const void* wpd, void* dx, int lddx, int dx_bf16,
                                          int N, int H, int W, int Cin, int Cout, const void* r_prev, int ldr, int r_bf16, int c0, int c1,
                                          float* stat_part, size_t stat_bytes, int max_workgroups, void* stream) {
    UNET_CHECK_ARG(dz && wpd && dx && unet_conv3x3_bf16_supported(N, H, W, Cout, Cin));
    UNET_CHECK_ARG(lddz >= Cout && lddx >= Cin && lddz % 4 == 0 && unet_aligned16(dz) && unet_aligned16(wpd));
    UNET_CHECK_ARG((r_prev == nullptr) == (stat_part == nullptr));
    UNET_CHECK_ARG(!r_prev || (c0 >= 0 && c0 < c1 && c1 <= Cin && c0 % 64 == 0 && c1 % 64 == 0 && ldr >= c1 - c0));
    UNET_CHECK_ARG((size_t)N * H * W * lddz * 4 < ((size_t)1 << 31) && (size_t)N * H * W * lddx * 4 < ((size_t)1 << 31));
    UNET_CHECK_ARG(!dz_bf16 || lddz % 8 == 0);
    const ConvBf16Stats s{2, stat_part, stat_bytes, (const float*)r_prev, ldr, c0, c1, r_bf16 ? 1 : 0};
    return run_conv_bf16((const float*)dz, lddz, wpd, nullptr, (float*)dx, lddx, N, H, W, Cout, Cin, 0, (hipStream_t)stream, r_prev ? &s : nullptr,
                         dz_bf16 ? 1 : 0, dx_bf16 ? 1 : 0, nullptr, nullptr, max_workgroups);
}
extern "C" int unet_conv3x3_dgrad_bf16(const void* dz, int lddz, int dz_bf16, const void* wpd, void* dx, int lddx, int dx_bf16,
                                          int N, int H, int W, int Cin, int Cout, const void* r_prev, int ldr, int r_bf16, int c0, int c1,
                                          float* stat_part, size_t stat_bytes, void* stream) {
    return unet_conv3x3_dgrad_bf16_wg(dz, lddz, dz_bf16, wpd, dx, lddx, dx_bf16, N, H, W, Cin, Cout, r_prev, ldr, r_bf16, c0, c1, stat_part, stat_bytes, 0, stream);
}

// ---- 2x2 / stride-2 transposed convolution on the bf16 matrix cores (forward and data gradient) -----------------------------------
//   forward  (UNet/model.py:39-48): z[n,2i+a,2j+b,co] = bias[co] + sum_ci bf16(x[n,i,j,ci]) * bf16(W[a,b,co,ci])
//   gradient:                       dx[n,i,j,ci] = sum_{a,b,co} bf16(dz[n,2i+a,2j+b,co]) * bf16(W[a,b,co,ci])
// Both are GEMMs over the INPUT-resolution pixels: forward K = Cin, N = 4 taps x Cout with the result scattered to the four
// sub-lattices of the output; gradient K = 4 taps x Cout (a k-chunk never straddles a tap: its gather is a scalar offset), N = Cin.
// One body: workgroup = 16 x 32 pixels x CT = 128 (64) columns, k-chunks of 32 (two MFMA k-steps), the staging / LDS images /
// fragment reads of the 3x3 kernel without the halo and the taps.  With so little arithmetic per byte these kernels live on
// the LDS fill and HBM, not on the matrix pipe; what bf16 buys is 16x less MFMA time next to that traffic.
namespace {

struct ConvtBf16Args {
    const float* x; const uint16_t* wp; const float* bias; float* out;
    int ldx, ldo, N, H, W, K, Ncol, Cout;          // H, W: input resolution; K, Ncol: GEMM depth / width; Cout: the layer's (tap = column / Cout)
    int tby, tbx, n_px, n_co;
    unsigned x_bytes; int in16;
    float* stat_part;
    const float* bn_r; int bn_ldr;               // MODE 2 + STATS 2: the producer's saved activation (all Ncol channels)
    int out16, r16;                              // the output / the producer's saved activation is stored as bf16 (ldo / bn_ldr in elements)
};

constexpr int kCtXP = 2 * 2 * 512 * 16;             // [k-step][k half][pixel][8] bf16

// XDMA = 1 (the input is a bf16 tensor): the chunk's input image is filled by LDS-DMA, 64 contiguous bytes (the pixel's 32 channels of
// the chunk) per pixel and 16 pixels per wave-instruction, and laid out [pixel][4 x 16 B] with the four pieces of a pixel rotated by
// (pixel >> 2) & 3 (XOR): the A fragment of (k-step ks, k half lh) is piece 2 ks + lh, and with a 64-byte lane stride only that XOR keeps
// the 16 lanes of a ds_read_b128 group on 16 different slots.  The rotation depends on the lane alone (pixel = 32 row + lane column and
// 32 row / 4 = 0 mod 4), so the read addresses stay two per-lane constants.  (The first DMA form of these kernels copied the plane
// layout of the register path -- 16 bytes per lane from 64 different pixels -- and gained nothing: four times the L2 requests.)
template <int NCO, int STATS, int MODE, int XDMA = 0>
__device__ __forceinline__ void convt_bf16_body(const ConvtBf16Args& p) {
    constexpr int CT = 32 * NCO;
    constexpr int WB = 2 * 2 * CT * 16;
    constexpr int STAGE = kCtXP + WB;
    constexpr int NPIECE = WB / 1024, KW = (NPIECE + 3) / 4;
    __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    int t = blockIdx.x;
    const int total = p.n_px * p.n_co;
    if ((total & 7) == 0) t = (t & 7) * (total >> 3) + (t >> 3);
    // Column tile fastest (round 6; the 3x3 kernels keep it slowest): the 2..16 column tiles of one pixel tile run next to each other on one
    // XCD, whose L2 then serves the input tile's second .. n-th read -- with the column tile slowest those reads were half a launch apart
    // (same-box micro-benchmark, four BASELINE shapes: forward + sums 0.383 -> 0.343 ms, data gradient + sums 0.370 -> 0.343 ms).
#ifndef UNET_CT_COLFAST
#define UNET_CT_COLFAST 1
#endif
    const int cot = UNET_CT_COLFAST ? t % p.n_co : t / p.n_px; int px = UNET_CT_COLFAST ? t / p.n_co : t % p.n_px;
    const int tpx = px;
    const int bx = px % p.tbx; px /= p.tbx;
    const int by = px % p.tby; const int img = px / p.tby;
    const int n0 = cot * CT, ty0 = 16 * by, tx0 = 32 * bx;
    const int nchunks = p.K / 32;
    const int es = p.in16 ? 2 : 4;

    // staging duty: quad f (4 of the chunk's 32 channels) of pixels (tid >> 3) + 32 j
    unsigned voff[16];
    const int f = tid & 7;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int pp = (tid >> 3) + 32 * j;
        const int gy = ty0 + (pp >> 5), gx = tx0 + (pp & 31);
        const bool ok = gy < p.H && gx < p.W;
        const size_t pix = MODE == 2 ? ((size_t)(img * 2 * p.H + 2 * gy) * (2 * p.W) + 2 * gx) : ((size_t)(img * p.H + gy) * p.W + gx);
        voff[j] = ok ? (unsigned)(pix * p.ldx * es + (p.in16 ? (f >> 1) * 16 : f * 16)) : 0x80000000u;
    }
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_b*)smem;
    // image [k-step = f >> 2][k half = (f >> 1) & 1][pixel][16 B], low / high 8 bytes by f & 1
    const unsigned wr_base = lds0 + (unsigned)(((f >> 2) * 2 + ((f >> 1) & 1)) * 8192 + (tid >> 3) * 16 + (f & 1) * 8);
    const unsigned a_base = XDMA ? lds0 + (unsigned)((4 * wv * 32 + li) * 64 + ((lh ^ ((li >> 2) & 3)) * 16))
                                 : lds0 + (unsigned)(lh * 8192 + (4 * wv * 32 + li) * 16);
    const unsigned a_base_k1 = XDMA ? lds0 + (unsigned)((4 * wv * 32 + li) * 64 + (((2 + lh) ^ ((li >> 2) & 3)) * 16)) : a_base + 16384;
    const unsigned b_base = lds0 + (unsigned)(kCtXP + lh * CT * 16 + li * 16);
    unsigned woff[KW];
#pragma unroll
    for (int k = 0; k < KW; ++k) {
        const int id = wv + 4 * k;
        const int row = id / (CT / 64), blk = id % (CT / 64);      // row = k-step * 2 + k half
        woff[k] = (unsigned)((row * p.Ncol + n0 + 64 * blk) * 16 + lane * 16);
    }
    const size_t wchunk = (size_t)4 * p.Ncol * 16;
    const char* wsrc = reinterpret_cast<const char*>(p.wp);

    f32x4 stg[XDMA ? 1 : 16];
    // XDMA: this wave's 8 pieces of a chunk cover pixels 16 (wv + 4 k) .. + 15, four lanes per pixel
    const char* dsrc[8];
    if constexpr (XDMA) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int pp = 16 * (wv + 4 * k) + (lane >> 2), j = (lane & 3) ^ ((pp >> 2) & 3);
            const int gy = ty0 + (pp >> 5), gx = tx0 + (pp & 31);
            const size_t pix = MODE == 2 ? ((size_t)(img * 2 * p.H + 2 * gy) * (2 * p.W) + 2 * gx) : ((size_t)(img * p.H + gy) * p.W + gx);
            dsrc[k] = (gy < p.H && gx < p.W) ? reinterpret_cast<const char*>(p.x) + pix * p.ldx * 2 + j * 16 : nullptr;
        }
    }
    auto issue_x = [&](int chunk, int stage) {
        int so;
        if (MODE == 2) {       // chunk = 32 channels of one tap (a, b): the tap's pixel offset and the channel offset are scalars
            const int k0 = chunk * 32, tap = k0 / p.Cout, c0 = k0 % p.Cout;
            so = (((tap >> 1) * 2 * p.W + (tap & 1)) * p.ldx + c0) * es;
        } else so = chunk * 32 * es;
        if constexpr (XDMA) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const char* src = dsrc[k] ? dsrc[k] + so : reinterpret_cast<const char*>(g_zero_page_b) + (lane & 3) * 16;
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(src),
                                                 (lds_void_b*)(smem + stage * STAGE + (wv + 4 * k) * 1024), 16, 0, 0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                stg[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd, (int)voff[j], so, 0));
        }
    };
    auto issue_w = [&](int chunk, int stage) {
        const char* src = wsrc + (size_t)chunk * wchunk;
#pragma unroll
        for (int k = 0; k < KW; ++k)
            if (NPIECE % 4 == 0 || wv + 4 * k < NPIECE)
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(src + woff[k]),
                                                 (lds_void_b*)(smem + stage * STAGE + kCtXP + (wv + 4 * k) * 1024), 16, 0, 0);
    };
    auto write_x = [&](int stage) {
        if constexpr (XDMA) return;
        const unsigned wb = wr_base + (unsigned)(stage * STAGE);
#pragma unroll
        for (int j = 0; j < (XDMA ? 1 : 16); ++j) {
            uint2 v; v.x = cb_pack2_pinned(stg[j].x, stg[j].y); v.y = cb_pack2_pinned(stg[j].z, stg[j].w);
            if (p.in16) {
                v.x = __builtin_bit_cast(unsigned, (f & 1) ? stg[j].z : stg[j].x); v.y = __builtin_bit_cast(unsigned, (f & 1) ? stg[j].w : stg[j].y);
            }
            asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(wb), "v"(v), "n"(j * 512) : "memory");
        }
    };
    f32x16 acc[4][NCO];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < NCO; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][c][e] = 0.f;
    // one chunk from stage ST: 2 k-steps x (4 pixel rows x NCO column tiles); all 8 + 2 NCO fragment reads up front, consumed in order
    auto compute = [&](unsigned ab, unsigned bb, unsigned ab1) {
        bf16x8 fa[2][4], fb[2][NCO];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int c = 0; c < NCO; ++c) CB_RD128(fb[ks][c], bb, ks * 2 * CT * 16 + c * 512);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (XDMA) { if (ks == 0) CB_RD128(fa[ks][r], ab, r * 2048); else CB_RD128(fa[ks][r], ab1, r * 2048); }
                else CB_RD128(fa[ks][r], ab, ks * 16384 + r * 512);
            }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // reads still allowed in flight when fa[ks][r] is needed: everything issued after it
                const int after = (1 - ks) * (NCO + 4) + (3 - r);
                if (NCO == 4) {
                    if (after >= 8) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(fa[ks][r]), "+v"(fb[ks][0]), "+v"(fb[ks][1]), "+v"(fb[ks][2]), "+v"(fb[ks][3]));
                    else wg_wait_tied4(after, fa[ks][r], fb[ks][0], fb[ks][1], fb[ks][2], fb[ks][3]);
                } else {
                    wg_wait_tied2(after, fa[ks][r], fb[ks][0], fb[ks][1]);
                }
#pragma unroll
                for (int c = 0; c < NCO; ++c) CB_MFMA(acc[r][c], fa[ks][r], fb[ks][c]);
            }
    };

    issue_x(0, 0); issue_w(0, 0);
    write_x(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int c = 0; c < nchunks; c += 2) {                            // K % 64 == 0: an even number of chunks
        issue_x(c + 1, 1); issue_w(c + 1, 1);
        compute(a_base, b_base, a_base_k1);
        write_x(1);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const int cn = c + 2 < nchunks ? c + 2 : c;
        issue_x(cn, 0); issue_w(cn, 0);
        compute(a_base + STAGE, b_base + STAGE, a_base_k1 + STAGE);
        write_x(0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

    // epilogue.  Forward: column n = tap * Cout + co -> output pixel (2 y + a, 2 x + b); gradient: plain [pixel][ci].
    // A lane holds ONE channel (cb + li) of 16 pixels per row.  fp32 output: one 4-byte store per element.  bf16 output (out16): lanes
    // 2j and 2j+1 exchange one value per pixel pair (DPP quad_perm swap) so that the even lane stores channels (2j, 2j+1) of the even
    // pixel and the odd lane the same channel pair of the odd pixel -- 4-byte stores again, half as many.  The producer's saved
    // activation (STATS 2) of a sub-tile is loaded for all four rows before the arithmetic (one exposed latency per sub-tile, not
    // four); stored as bf16 (r16) a lane reads the dword holding its channel pair and keeps its half.
    float st1[NCO], st2[NCO];
    const int oH = MODE == 1 ? 2 * p.H : p.H, oW = MODE == 1 ? 2 * p.W : p.W, pstep = MODE == 1 ? 2 : 1;
    const int oes = p.out16 ? 2 : 4, res = p.r16 ? 2 : 4;
    const __amdgpu_buffer_rsrc_t srd_o = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, (int)((size_t)p.N * oH * oW * p.ldo * oes), 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_r = __builtin_amdgcn_make_buffer_rsrc((void*)(STATS == 2 ? p.bn_r : p.out), 0,
                                                                            STATS == 2 ? (int)((size_t)p.N * p.H * p.W * p.bn_ldr * res) : 0, 0x00020000);
    const int lpar = li & 1;
    int ovoff[16], rvoff[16];
    bool colok[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int col = (e & 3) + 8 * (e >> 2) + 4 * lh;
        colok[e] = tx0 + col < p.W;
        // out16: the store of pixel pair (e & ~1, e | 1) is issued at the even e by every lane; even lanes write the even pixel, odd lanes the odd one
        const int colp = (e & ~1) + lpar, colq = (colp & 3) + 8 * (colp >> 2) + 4 * lh;
        if (p.out16) ovoff[e] = (tx0 + colq < p.W) ? (pstep * colq * p.ldo + (li & ~1)) * 2 : (int)0x80000000;
        else         ovoff[e] = colok[e] ? (pstep * col * p.ldo + li) * 4 : (int)0x80000000;
        rvoff[e] = (STATS == 2 && colok[e]) ? (p.r16 ? (col * p.bn_ldr + (li & ~1)) * 2 : (col * p.bn_ldr + li) * 4) : (int)0x80000000;
    }
#pragma unroll
    for (int c = 0; c < NCO; ++c) {
        const int nb = n0 + 32 * c;                                  // first column of this sub-tile
        const int tap = MODE == 1 ? nb / p.Cout : 0, cb = MODE == 1 ? nb % p.Cout : nb;
        const float bv = (MODE == 1 && p.bias) ? p.bias[cb + li] : 0.f;
        st1[c] = 0.f; st2[c] = 0.f;
        unsigned rvh[4][16];
        if (STATS == 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gy = ty0 + 4 * wv + r;
                const int sr = (((img * p.H + (gy < p.H ? gy : 0)) * p.W + tx0) * p.bn_ldr + cb) * res;
#pragma unroll
                for (int e = 0; e < 16; ++e) rvh[r][e] = __builtin_amdgcn_raw_buffer_load_b32(srd_r, gy < p.H ? rvoff[e] : (int)0x80000000, sr, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gy = ty0 + 4 * wv + r;
            if (gy >= p.H) continue;
            const int opix = MODE == 1 ? ((img * oH + 2 * gy + (tap >> 1)) * oW + 2 * tx0 + (tap & 1)) : ((img * p.H + gy) * p.W + tx0);
            const int so = (opix * p.ldo + cb) * oes;
            float vv[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v;
                asm("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(acc[r][c][e]));
                v += bv;
                vv[e] = v;
                if (STATS != 0 && colok[e]) {
                    float rv = 0.f;
                    if (STATS == 2) rv = p.r16 ? __builtin_bit_cast(float, lpar ? (rvh[r][e] & 0xffff0000u) : (rvh[r][e] << 16)) : __builtin_bit_cast(float, rvh[r][e]);
                    st1[c] += v; st2[c] += STATS == 1 ? v * v : v * rv;
                }
            }
            if (p.out16) {
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    // even lane keeps pixel e and receives its neighbour's value of pixel e; odd lane keeps pixel e+1 and receives pixel e+1
                    const float send = lpar ? vv[e] : vv[e + 1];
                    const float recv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0xB1, 0xF, 0xF, false));
                    const unsigned pk = lpar ? cb_pack2(recv, vv[e + 1]) : cb_pack2(vv[e], recv);
                    __builtin_amdgcn_raw_buffer_store_b32(pk, srd_o, ovoff[e], so, UNET_NT_AUX(UNET_NT_CONV16));
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vv[e]), srd_o, ovoff[e], so, 0);
            }
        }
    }
    if (STATS != 0) {
        // per-channel sums; forward: one partial row per (pixel tile, tap), so the four taps of a channel are four rows
        float* red = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int c = 0; c < NCO; ++c) {
            st1[c] += __shfl_xor(st1[c], 32); st2[c] += __shfl_xor(st2[c], 32);
            if (lh == 0) { red[((wv * NCO + c) * 32 + li) * 2] = st1[c]; red[((wv * NCO + c) * 32 + li) * 2 + 1] = st2[c]; }
        }
        __syncthreads();
        if (tid < CT) {
            const int c = tid >> 5, l = tid & 31;
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int w4 = 0; w4 < 4; ++w4) { a += red[((w4 * NCO + c) * 32 + l) * 2]; b += red[((w4 * NCO + c) * 32 + l) * 2 + 1]; }
            const int ncol = n0 + tid;
            const int ch = MODE == 1 ? ncol % p.Cout : ncol, tap = MODE == 1 ? ncol / p.Cout : 0;
            const int rows = MODE == 1 ? 4 * p.n_px : p.n_px, row = MODE == 1 ? 4 * tpx + tap : tpx;
            float* o = p.stat_part + (((size_t)(ch >> 6) * rows + row) * 64 + (ch & 63)) * 2;
            o[0] = a; o[1] = b;
        }
    }
}

}  // namespace

namespace {

__global__ __launch_bounds__(256, 1) void convt_bf16_fwd_kernel_128(ConvtBf16Args p) { convt_bf16_body<4, 0, 1>(p); }
__global__ __launch_bounds__(256, 1) void convt_bf16_fwd_kernel_64(ConvtBf16Args p) { convt_bf16_body<2, 0, 1>(p); }
__global__ __launch_bounds__(256, 1) void convt_bf16_fwd_stats_kernel_128(ConvtBf16Args p) { convt_bf16_body<4, 1, 1>(p); }
__global__ __launch_bounds__(256, 1) void convt_bf16_fwd_stats_kernel_64(ConvtBf16Args p) { convt_bf16_body<2, 1, 1>(p); }
__global__ __launch_bounds__(256, 1) void convt_bf16_dgrad_kernel_128(ConvtBf16Args p) { convt_bf16_body<4, 0, 2>(p); }
__global__ __launch_bounds__(256, 1) void convt_bf16_dgrad_kernel_64(ConvtBf16Args p) { convt_bf16_body<2, 0, 2>(p); }
__global__ __launch_bounds__(256, 1) void convt_bf16_dgrad_bnbwd_kernel_128(ConvtBf16Args p) { convt_bf16_body<4, 2, 2>(p); }
__global__ __launch_bounds__(256, 1) void convt_bf16_dgrad_bnbwd_kernel_64(ConvtBf16Args p) { convt_bf16_body<2, 2, 2>(p); }
// bf16-stored input: LDS-DMA staging
__global__ __launch_bounds__(256, 1) void convt_bf16_fwd_dma_kernel_128(ConvtBf16Args p) { convt_bf16_body<4, 0, 1, 1>(p); }
#ifndef UNET_CT_OCC
#define UNET_CT_OCC 2           /* workgroups per CU of the 64-column forward kernels (bf16-stored input) */
#endif
__global__ __launch_bounds__(256, UNET_CT_OCC) void convt_bf16_fwd_dma_kernel_64(ConvtBf16Args p) { convt_bf16_body<2, 0, 1, 1>(p); }
__global__ __launch_bounds__(256, 1) void convt_bf16_fwd_stats_dma_kernel_128(ConvtBf16Args p) { convt_bf16_body<4, 1, 1, 1>(p); }
__global__ __launch_bounds__(256, UNET_CT_OCC) void convt_bf16_fwd_stats_dma_kernel_64(ConvtBf16Args p) { convt_bf16_body<2, 1, 1, 1>(p); }
__global__ __launch_bounds__(256, 1) void convt_bf16_dgrad_dma_kernel_128(ConvtBf16Args p) { convt_bf16_body<4, 0, 2, 1>(p); }
__global__ __launch_bounds__(256, 1) void convt_bf16_dgrad_dma_kernel_64(ConvtBf16Args p) { convt_bf16_body<2, 0, 2, 1>(p); }
__global__ __launch_bounds__(256, 1) void convt_bf16_dgrad_bnbwd_dma_kernel_128(ConvtBf16Args p) { convt_bf16_body<4, 2, 2, 1>(p); }
__global__ __launch_bounds__(256, 1) void convt_bf16_dgrad_bnbwd_dma_kernel_64(ConvtBf16Args p) { convt_bf16_body<2, 2, 2, 1>(p); }

// B[K][Ncol] of the GEMM -> bf16 [K/32][k-step][k half][Ncol][8]; the Keras kernel [2][2][Cout][Cin] is B^T for the forward
// (k = ci, column = tap * Cout + co) and B itself for the data gradient (k = tap * Cout + co, column = ci)
__global__ void convt_bf16_pack_kernel(const float* __restrict__ w, uint16_t* __restrict__ wp, int K, int Ncol, int transposed) {
    const long n = (long)(K / 8) * Ncol;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int col = (int)(i % Ncol); const int k0 = (int)(i / Ncol) * 8;       // (chunk, k-step, k half) flattened = k0 / 8
    unsigned v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int k = k0 + 2 * e;
        const float a = transposed ? w[(size_t)col * K + k] : w[(size_t)k * Ncol + col];
        const float b = transposed ? w[(size_t)col * K + k + 1] : w[(size_t)(k + 1) * Ncol + col];
        v[e] = cb_pack2(a, b);
    }
    reinterpret_cast<uint4*>(wp)[i] = make_uint4(v[0], v[1], v[2], v[3]);
}

int run_convt_bf16(int mode, const void* x, int ldx, int in16, const void* wp, const float* bias, float* out, int ldo,
                   int N, int H, int W, int Cin, int Cout, float* stat_part, size_t stat_bytes, const float* r_prev, int ldr, hipStream_t st,
                   int out16 = 0, int r16 = 0) {
    ConvtBf16Args a{};
    a.out16 = out16; a.r16 = r16;
    a.x = (const float*)x; a.wp = (const uint16_t*)wp; a.bias = bias; a.out = out; a.ldx = ldx; a.ldo = ldo;
    a.N = N; a.H = H; a.W = W; a.Cout = Cout; a.in16 = in16;
    a.K = mode == 1 ? Cin : 4 * Cout; a.Ncol = mode == 1 ? 4 * Cout : Cin;
    a.tby = (H + 15) / 16; a.tbx = (W + 31) / 32; a.n_px = N * a.tby * a.tbx;
    a.x_bytes = (unsigned)((size_t)N * H * W * (mode == 2 ? 4 : 1) * ldx * (in16 ? 2 : 4));
    // Forward with K <= 256 (up_1, up_2): 4-8 chunks of 32 MFMAs between a prologue whose first loads come from HBM and an epilogue that writes
    // 128 KB -- one workgroup per CU leaves every one of those latencies exposed.  64-column tiles at TWO workgroups per CU (128 accumulators,
    // 72 KB of LDS) hide them in each other; the input tile's extra reads come from L2 (column tile fastest).  Same box, forward + sums:
    // up_1 0.141 -> 0.112 ms, up_2 0.089 -> 0.079; at K >= 512 the wide tile wins (up_3 0.063 either way, up_4 0.050 vs 0.056), and the data
    // gradient's 64-column kernel with BatchNorm-backward sums does not fit 256 registers (it spills: 0.141 -> 0.161 on up_1).
    const bool narrow_k = UNET_CT_OCC > 1 && in16 && mode == 1 && a.K <= 256;
    const bool wide = a.Ncol % 128 == 0 && (long)a.n_px * (a.Ncol / 128) >= conv_bf16_cus() && !narrow_k;
    a.n_co = a.Ncol / (wide ? 128 : 64);
    a.stat_part = stat_part; a.bn_r = r_prev; a.bn_ldr = ldr;
    if (stat_part) {
        const int chans = mode == 1 ? Cout : Cin, rows = (mode == 1 ? 4 : 1) * a.n_px;
        if (stat_bytes < (size_t)(chans / 64) * rows * 128 * sizeof(float)) return UNET_ENOSPC;
    }
    const dim3 grid((unsigned)(a.n_px * a.n_co));
    if (in16) {                                     // bf16-stored input: LDS-DMA staging (64 contiguous bytes per pixel)
        if (mode == 1) {
            if (stat_part) { if (wide) convt_bf16_fwd_stats_dma_kernel_128<<<grid, 256, 0, st>>>(a); else convt_bf16_fwd_stats_dma_kernel_64<<<grid, 256, 0, st>>>(a); }
            else           { if (wide) convt_bf16_fwd_dma_kernel_128<<<grid, 256, 0, st>>>(a); else convt_bf16_fwd_dma_kernel_64<<<grid, 256, 0, st>>>(a); }
        } else {
            if (stat_part) { if (wide) convt_bf16_dgrad_bnbwd_dma_kernel_128<<<grid, 256, 0, st>>>(a); else convt_bf16_dgrad_bnbwd_dma_kernel_64<<<grid, 256, 0, st>>>(a); }
            else           { if (wide) convt_bf16_dgrad_dma_kernel_128<<<grid, 256, 0, st>>>(a); else convt_bf16_dgrad_dma_kernel_64<<<grid, 256, 0, st>>>(a); }
        }
        return UNET_LAUNCH_STATUS();
    }
    if (mode == 1) {
        if (stat_part) { if (wide) convt_bf16_fwd_stats_kernel_128<<<grid, 256, 0, st>>>(a); else convt_bf16_fwd_stats_kernel_64<<<grid, 256, 0, st>>>(a); }
        else           { if (wide) convt_bf16_fwd_kernel_128<<<grid, 256, 0, st>>>(a); else convt_bf16_fwd_kernel_64<<<grid, 256, 0, st>>>(a); }
    } else {
        if (stat_part) { if (wide) convt_bf16_dgrad_bnbwd_kernel_128<<<grid, 256, 0, st>>>(a); else convt_bf16_dgrad_bnbwd_kernel_64<<<grid, 256, 0, st>>>(a); }
        else           { if (wide) convt_bf16_dgrad_kernel_128<<<grid, 256, 0, st>>>(a); else convt_bf16_dgrad_kernel_64<<<grid, 256, 0, st>>>(a); }
    }
    return UNET_LAUNCH_STATUS();
}

}  // namespace

// Cin, Cout multiples of 64; tensors < 2 GiB
extern "C" int unet_convT2x2_bf16_supported(int N, int H, int W, int Cin, int Cout) {
    return (N > 0 && H > 0 && W > 0 && Cin % 64 == 0 && Cout % 64 == 0 && (size_t)N * H * W * 4 * Cout * 4 < ((size_t)1 << 31) &&
            (size_t)N * H * W * Cin * 4 < ((size_t)1 << 31)) ? 1 : 0;
}
extern "C" size_t unet_convT2x2_bf16_packed_bytes(int Cin, int Cout) { return (size_t)4 * Cin * Cout * 2; }
// mode 0: forward operand, mode 1: data-gradient operand; w is the Keras kernel [2][2][Cout][Cin] (UNet/model.py:41-46)
extern "C" int unet_convT2x2_bf16_pack_weights(const float* w, void* packed, int Cin, int Cout, int mode, void* stream) {
    UNET_CHECK_ARG(w && packed && Cin % 64 == 0 && Cout % 64 == 0 && (mode == 0 || mode == 1) && unet_aligned16(packed));
    const int K = mode == 0 ? Cin : 4 * Cout, Ncol = mode == 0 ? 4 * Cout : Cin;
    const long n = (long)(K / 8) * Ncol;
    convt_bf16_pack_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(w, (uint16_t*)packed, K, Ncol, mode == 0 ? 1 : 0);
    return UNET_LAUNCH_STATUS();
}
// rows of the statistics partials: forward 4 per 16x32 input-pixel tile (one per tap), data gradient 1
extern "C" int unet_convT2x2_bf16_stats_rows(int N, int H, int W, int Cin, int Cout, int dgrad) {
    if (!unet_convT2x2_bf16_supported(N, H, W, Cin, Cout)) return 0;
    return (dgrad ? 1 : 4) * N * ((H + 15) / 16) * ((W + 31) / 32);
}
// z[n,2i+a,2j+b,co] = bias[co] + sum_ci x[n,i,j,ci] W[a,b,co,ci] (operands rounded to bf16, fp32 accumulation); H, W: input size;
// x_bf16 / out_bf16: x / the output stored as bf16 (leading dimensions in elements; ldo even); stat_part nullable: BatchNorm sums of
// z (taken before the rounding of a bf16 output), [Cout/64][rows][64][2]
extern "C" int unet_convT2x2_fwd_bf16(const void* x, int ldx, int x_bf16, const void* wp, const float* bias, void* out, int ldo, int out_bf16,
                                         int N, int H, int W, int Cin, int Cout, float* stat_part, size_t stat_bytes, void* stream) {
    UNET_CHECK_ARG(x && wp && out && unet_convT2x2_bf16_supported(N, H, W, Cin, Cout) && (!out_bf16 || (ldo % 2 == 0 && (reinterpret_cast<uintptr_t>(out) & 3u) == 0)));
    UNET_CHECK_ARG(ldx >= Cin && ldo >= Cout && ldx % 4 == 0 && (!x_bf16 || ldx % 8 == 0) && unet_aligned16(x) && unet_aligned16(wp));
    UNET_CHECK_ARG((size_t)N * H * W * ldx * 4 < ((size_t)1 << 31) && (size_t)N * H * W * 4 * ldo * 4 < ((size_t)1 << 31));
    return run_convt_bf16(1, x, ldx, x_bf16 ? 1 : 0, wp, bias, (float*)out, ldo, N, H, W, Cin, Cout, stat_part, stat_bytes, nullptr, 0, (hipStream_t)stream, out_bf16 ? 1 : 0, 0);
}
// dx[n,i,j,ci] = sum_{a,b,co} dz[n,2i+a,2j+b,co] W[a,b,co,ci]; r_prev / stat_part nullable: BatchNorm-backward sums (sum dx, sum dx * r_prev)
// of the layer that produced x (all Cin channels), [Cin/64][rows][64][2]
// dx_bf16 / r_bf16: dx / r_prev stored as bf16
extern "C" int unet_convT2x2_dgrad_bf16(const void* dz, int lddz, int dz_bf16, const void* wpd, void* dx, int lddx, int dx_bf16,
                                           int N, int H, int W, int Cin, int Cout, const void* r_prev, int ldr, int r_bf16,
                                           float* stat_part, size_t stat_bytes, void* stream) {
    UNET_CHECK_ARG(dz && wpd && dx && unet_convT2x2_bf16_supported(N, H, W, Cin, Cout));
    UNET_CHECK_ARG((!dx_bf16 || (lddx % 2 == 0 && (reinterpret_cast<uintptr_t>(dx) & 3u) == 0)) && (!r_bf16 || (ldr % 2 == 0 && (reinterpret_cast<uintptr_t>(r_prev) & 3u) == 0)));
    UNET_CHECK_ARG(lddz >= Cout && lddx >= Cin && lddz % 4 == 0 && (!dz_bf16 || lddz % 8 == 0) && unet_aligned16(dz) && unet_aligned16(wpd));
    UNET_CHECK_ARG((r_prev == nullptr) == (stat_part == nullptr) && (!r_prev || ldr >= Cin));
    UNET_CHECK_ARG((size_t)N * H * W * 4 * lddz * 4 < ((size_t)1 << 31) && (size_t)N * H * W * lddx * 4 < ((size_t)1 << 31));
    return run_convt_bf16(2, dz, lddz, dz_bf16 ? 1 : 0, wpd, nullptr, (float*)dx, lddx, N, H, W, Cin, Cout, stat_part, stat_bytes, (const float*)r_prev, ldr, (hipStream_t)stream,
                          dx_bf16 ? 1 : 0, r_bf16 ? 1 : 0);
}

// ---- transposed-conv weight gradient on the bf16 matrix cores -------------------------------------------------------------------
//   dw[a,b,co,ci] = sum_{n,i,j} bf16(dz[n,2i+a,2j+b,co]) * bf16(x[n,i,j,ci])        (Keras kernel layout [2][2][Cout][Cin])
// Contraction over input pixels, operands gathered by the transposing LDS read as in wgrad_bf16_kernel; the four taps are the
// four stride-2 sub-lattices of dz, i.e. a row slot plus a pixel offset with DOUBLED pixel stride in the read address.
// Workgroup = CO_T (128 / 64) output x 128 input channels x 4 taps; unit of work = one input row segment of 32 pixels (+ its two
// dz rows of 64 pixels): every thread stages the same 4 + CO_T/8 loads per unit (no roles), one unit ahead; units are dealt to the
// workgroups of a channel tile round-robin and the partial sums added in a fixed order.  Per unit 32 (16) MFMAs per wave against
// 80 (48) KB of fp32 operands: the kernel runs on HBM / L2 bandwidth, which is the point -- the fp32 kernel ran on the matrix pipe.
namespace {

struct CtWgBf16Args {
    const float* x; const float* dz; float* out;
    int ldx, lddz, N, H, W, Cin, Cout;
    int n_co, n_ci, splits, tbx, n_units;
    unsigned x_bytes, dz_bytes; int x16, z16;
};

template <int MT>      // 32-channel blocks of dz per wave; the workgroup covers CO_T = 64 * MT output channels
__device__ __forceinline__ void convt_wgrad_bf16_body(const CtWgBf16Args& p) {
    constexpr int CO_T = 64 * MT, ZG = CO_T / 32;                    // dz channel groups of 32
    constexpr int XROW = 4 * 32 * 64, ZROW = ZG * 64 * 64;           // bytes of the staged x row (4 groups x 32 px) / one dz row (ZG groups x 64 px)
    constexpr int STAGE = XROW + 2 * ZROW;
    constexpr int XL = 4, ZL = 2 * 64 * (CO_T / 4) / 256;            // loads per thread: x 32 px x 32 quads, dz 2 rows x 64 px x CO_T/4 quads
    __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wv & 1, wn = wv >> 1;
    const int npairs = p.n_co * p.n_ci;
    const int pair = blockIdx.x % npairs, split = blockIdx.x / npairs;
    const int co0 = (pair / p.n_ci) * CO_T, ci0 = (pair % p.n_ci) * 128;
    const int xes = p.x16 ? 2 : 4, zes = p.z16 ? 2 : 4;
    const __amdgpu_buffer_rsrc_t srd_x = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(p.x) + (size_t)ci0 * xes), 0,
                                                                            (int)(p.x_bytes - (unsigned)ci0 * xes), 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_z = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(p.dz) + (size_t)co0 * zes), 0,
                                                                            (int)(p.dz_bytes - (unsigned)co0 * zes), 0x00020000);
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_b*)smem;
    // staging: x load j covers quad xq of pixel xp + 8 j; dz load j covers quad zq of (row, pixel) index zi + (256 / ZQ) j
    constexpr int ZQ = CO_T / 4;
    const int xq = tid & 31, xp = tid >> 5, zq = tid % ZQ, zi = tid / ZQ;
    const unsigned xw = lds0 + (unsigned)((xq >> 3) * (32 * 64) + xp * 64 + (xq & 7) * 8);
    const unsigned zw = lds0 + XROW + (unsigned)((zq >> 3) * (64 * 64) + (zq & 7) * 8);
    // fragment gathers (see wgrad_bf16_kernel): x pixels at stride 64 B, dz pixels of one sub-lattice at stride 128 B
    const int g4 = lane >> 4, q4 = (lane >> 2) & 3, pp = lane & 3;
    const unsigned chan = (unsigned)((16 * (g4 & 1) + 4 * pp) * 2);
    const unsigned b_lane = lds0 + (unsigned)((2 * wn) * (32 * 64) + (8 * (g4 >> 1) + q4) * 64) + chan;
    const unsigned a_lane = lds0 + XROW + (unsigned)((MT * wm) * (64 * 64) + (8 * (g4 >> 1) + q4) * 128) + chan;

    f32x16 acc[4][MT][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[t][m][c][e] = 0.f;

    f32x4 sx[XL], sz[ZL];
    auto issue = [&](int u) {
        const int uc = u < p.n_units ? u : split;                    // past the end: a valid unit, never used
        const int seg = uc % p.tbx, row = uc / p.tbx;                // row = n * H + i
        const int x0 = 32 * seg;
        const int img = row / p.H, i = row % p.H;
#pragma unroll
        for (int j = 0; j < XL; ++j) {
            const int gx = x0 + xp + 8 * j;
            const unsigned vo = gx < p.W ? (unsigned)((((size_t)row * p.W + gx) * p.ldx) * xes + (p.x16 ? (xq >> 1) * 16 : xq * 16)) : 0x80000000u;
            sx[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_x, (int)vo, 0, 0));
        }
#pragma unroll
        for (int j = 0; j < ZL; ++j) {
            const int idx = zi + (256 / ZQ) * j;                      // 0 .. 127 = dz row (0/1) * 64 + pixel
            const int zr = idx >> 6, zp = idx & 63;
            const int gx = 2 * x0 + zp;
            const unsigned vo = gx < 2 * p.W ? (unsigned)((((size_t)(img * 2 * p.H + 2 * i + zr) * (2 * p.W) + gx) * p.lddz) * zes + (p.z16 ? (zq >> 1) * 16 : zq * 16)) : 0x80000000u;
            sz[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_z, (int)vo, 0, 0));
        }
    };
    auto commit = [&](int stage) {
#pragma unroll
        for (int j = 0; j < XL; ++j) {
            uint2 v; v.x = cb_pack2_pinned(sx[j].x, sx[j].y); v.y = cb_pack2_pinned(sx[j].z, sx[j].w);
            if (p.x16) { v.x = __builtin_bit_cast(unsigned, (xq & 1) ? sx[j].z : sx[j].x); v.y = __builtin_bit_cast(unsigned, (xq & 1) ? sx[j].w : sx[j].y); }
            asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(xw + (unsigned)(stage * STAGE)), "v"(v), "n"(j * 8 * 64) : "memory");
        }
#pragma unroll
        for (int j = 0; j < ZL; ++j) {
            const int idx = zi + (256 / ZQ) * j;
            const unsigned off = (unsigned)((idx >> 6) * ZROW + (idx & 63) * 64);
            uint2 v; v.x = cb_pack2_pinned(sz[j].x, sz[j].y); v.y = cb_pack2_pinned(sz[j].z, sz[j].w);
            if (p.z16) { v.x = __builtin_bit_cast(unsigned, (zq & 1) ? sz[j].z : sz[j].x); v.y = __builtin_bit_cast(unsigned, (zq & 1) ? sz[j].w : sz[j].y); }
            asm volatile("ds_write_b64 %0, %1" :: "v"(zw + off + (unsigned)(stage * STAGE)), "v"(v) : "memory");
        }
    };
    auto compute = [&](int stage) {
        const unsigned ab = a_lane + (unsigned)(stage * STAGE), bb = b_lane + (unsigned)(stage * STAGE);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            i32x2 fa[4][MT][2], fb[2][2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                WG_RDTR(fb[c][0], bb, c * (32 * 64) + ks * 1024);
                WG_RDTR(fb[c][1], bb, c * (32 * 64) + ks * 1024 + 256);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    WG_RDTR(fa[t][m][0], ab, (t >> 1) * ZROW + m * (64 * 64) + (32 * ks + (t & 1)) * 64);
                    WG_RDTR(fa[t][m][1], ab, (t >> 1) * ZROW + m * (64 * 64) + (32 * ks + (t & 1)) * 64 + 512);
                }
            // every fragment register is tied to the wait (a copy the compiler makes of one must not be scheduled ahead of it)
            if constexpr (MT == 2)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[1][0]), "+v"(fb[1][1]),
                             "+v"(fa[0][0][0]), "+v"(fa[0][0][1]), "+v"(fa[0][1][0]), "+v"(fa[0][1][1]), "+v"(fa[1][0][0]), "+v"(fa[1][0][1]), "+v"(fa[1][1][0]), "+v"(fa[1][1][1]),
                             "+v"(fa[2][0][0]), "+v"(fa[2][0][1]), "+v"(fa[2][1][0]), "+v"(fa[2][1][1]), "+v"(fa[3][0][0]), "+v"(fa[3][0][1]), "+v"(fa[3][1][0]), "+v"(fa[3][1][1]));
            else
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[1][0]), "+v"(fb[1][1]),
                             "+v"(fa[0][0][0]), "+v"(fa[0][0][1]), "+v"(fa[1][0][0]), "+v"(fa[1][0][1]), "+v"(fa[2][0][0]), "+v"(fa[2][0][1]), "+v"(fa[3][0][0]), "+v"(fa[3][0][1]));
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    bf16x8 av;
                    { i32x4 t4; t4.x = fa[t][m][0].x; t4.y = fa[t][m][0].y; t4.z = fa[t][m][1].x; t4.w = fa[t][m][1].y; av = __builtin_bit_cast(bf16x8, t4); }
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        bf16x8 bv;
                        { i32x4 t4; t4.x = fb[c][0].x; t4.y = fb[c][0].y; t4.z = fb[c][1].x; t4.w = fb[c][1].y; bv = __builtin_bit_cast(bf16x8, t4); }
                        // (s_nop: wait states for compiler-made copies in front of an inline-asm MFMA, see wgrad_bf16_kernel)
                        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[t][m][c]) : "v"(av), "v"(bv) : "memory");
                    }
                }
        }
    };

    int u = split;
    issue(u);
    commit(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    int stage = 0;
    for (; u < p.n_units; u += p.splits) {
        issue(u + p.splits);
        compute(stage);
        commit(stage ^ 1);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        stage ^= 1;
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    // accumulator register e of (tap t, block m, sub-tile c) = (output channel co0 + 32 (MT wm + m) + row(e, lh), input channel ci0 + 32 (2 wn + c) + li)
    float* o_base = p.out + (size_t)split * 4 * p.Cout * p.Cin;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int co = co0 + 32 * (MT * wm + m) + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    float v;
                    asm("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(acc[t][m][c][e]));
                    o_base[((size_t)t * p.Cout + co) * p.Cin + ci0 + 32 * (2 * wn + c) + li] = v;
                }
}

}  // namespace

namespace {

#ifndef UNET_CTWG_OCC
#define UNET_CTWG_OCC 1         /* workgroups per CU of the 64-output-channel weight-gradient kernel (2: and twice the pixel splits) */
#endif
__global__ __launch_bounds__(256, 1) void convt_wgrad_bf16_kernel_128(CtWgBf16Args p) { convt_wgrad_bf16_body<2>(p); }
__global__ __launch_bounds__(256, UNET_CTWG_OCC) void convt_wgrad_bf16_kernel_64(CtWgBf16Args p) { convt_wgrad_bf16_body<1>(p); }

void convt_wgrad_bf16_plan(CtWgBf16Args& a, int max_workgroups) {
    // 64-output-channel tiles by default: 128 AGPRs + 103 VGPRs, so the kernel shares a CU with the other stream's kernels (the
    // 128-channel tile takes all 256 AGPRs; alone it is as fast, in the step 0.13 ms slower -- compile with -DUNET_CONVT_WGRAD_WIDE to get it)
    // Round 6 tried a second workgroup of its OWN per CU (-DUNET_CTWG_OCC=2: launch bounds (256, 2), twice the pixel splits -- a unit is 16
    // MFMAs per wave between two barriers with register-staged operands): 0.413 -> 0.349 ms over the four BASELINE shapes stand-alone and
    // -0.13 ms on a single-stream step, but +0.05 ms on the two-stream step the product runs (12.740 -> 12.792 ms, four alternating runs):
    // the slots it fills are the ones the other stream's kernels were using.  Left at one.
#ifdef UNET_CONVT_WGRAD_WIDE
    const int cot = a.Cout % 128 == 0 ? 128 : 64;
#else
    const int cot = 64;
#endif
    a.n_co = a.Cout / cot; a.n_ci = a.Cin / 128;
    a.tbx = (a.W + 31) / 32; a.n_units = a.N * a.H * a.tbx;
    const int npairs = a.n_co * a.n_ci;
    int splits = unet_grid_slots(conv_bf16_cus(), max_workgroups) * (cot == 64 ? UNET_CTWG_OCC : 1) / npairs; if (splits < 1) splits = 1; if (splits > a.n_units) splits = a.n_units;
    a.splits = splits;
}

}  // namespace

extern "C" int unet_convT2x2_wgrad_bf16_supported(int N, int H, int W, int Cin, int Cout) {
    return (unet_convT2x2_bf16_supported(N, H, W, Cin, Cout) && Cin % 128 == 0) ? 1 : 0;
}
extern "C" size_t unet_convT2x2_wgrad_bf16_workspace_wg(int N, int H, int W, int Cin, int Cout, int max_workgroups) {
    if (!unet_convT2x2_wgrad_bf16_supported(N, H, W, Cin, Cout)) return 0;
    CtWgBf16Args a{}; a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    convt_wgrad_bf16_plan(a, max_workgroups);
    return a.splits > 1 ? (size_t)a.splits * 4 * Cout * Cin * sizeof(float) : 16;
}
extern "C" size_t unet_convT2x2_wgrad_bf16_workspace(int N, int H, int W, int Cin, int Cout) {
    return unet_convT2x2_wgrad_bf16_workspace_wg(N, H, W, Cin, Cout, 0);
}
// dw[a,b,co,ci] = sum_{n,i,j} dz[n,2i+a,2j+b,co] * xin[n,i,j,ci], operands rounded to bf16 (or stored as bf16), fp32 accumulation
extern "C" int unet_convT2x2_wgrad_bf16_wg(const void* xin, int ldx, int x_bf16, const void* dz, int lddz, int dz_bf16, float* dw,
                                           int N, int H, int W, int Cin, int Cout, int max_workgroups, void* ws, size_t ws_bytes, void* stream) {
    UNET_CHECK_ARG(xin && dz && dw && ws && unet_convT2x2_wgrad_bf16_supported(N, H, W, Cin, Cout));
    UNET_CHECK_ARG(ldx >= Cin && lddz >= Cout && ldx % 4 == 0 && lddz % 4 == 0 && (!x_bf16 || ldx % 8 == 0) && (!dz_bf16 || lddz % 8 == 0));
    UNET_CHECK_ARG(unet_aligned16(xin) && unet_aligned16(dz) && unet_aligned16(dw) && unet_aligned16(ws));
    UNET_CHECK_ARG((size_t)N * H * W * ldx * 4 < ((size_t)1 << 31) && (size_t)N * H * W * 4 * lddz * 4 < ((size_t)1 << 31));
    if (ws_bytes < unet_convT2x2_wgrad_bf16_workspace_wg(N, H, W, Cin, Cout, max_workgroups)) return UNET_ENOSPC;
    CtWgBf16Args a{};
    a.x = (const float*)xin; a.dz = (const float*)dz; a.ldx = ldx; a.lddz = lddz; a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.x16 = x_bf16 ? 1 : 0; a.z16 = dz_bf16 ? 1 : 0;
    convt_wgrad_bf16_plan(a, max_workgroups);
    a.out = a.splits > 1 ? (float*)ws : dw;
    a.x_bytes = (unsigned)((size_t)N * H * W * ldx * (a.x16 ? 2 : 4)); a.dz_bytes = (unsigned)((size_t)N * H * W * 4 * lddz * (a.z16 ? 2 : 4));
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)(a.n_co * a.n_ci * a.splits));
    if (a.n_co * 128 == Cout) convt_wgrad_bf16_kernel_128<<<grid, 256, 0, st>>>(a);
    else                      convt_wgrad_bf16_kernel_64<<<grid, 256, 0, st>>>(a);
    int rc = UNET_LAUNCH_STATUS(); if (rc) return rc;
    if (a.splits > 1) {
        const long n4 = (long)4 * Cout * Cin / 4;
        int sl = 1;
        while (sl < 16 && 2 * sl <= a.splits && n4 * sl < 256 * 1024) sl *= 2;
        wgrad_bf16_reduce_kernel<<<(unsigned)((n4 * sl + 255) / 256), 256, 0, st>>>((const float*)ws, dw, n4, a.splits, sl);
        rc = UNET_LAUNCH_STATUS();
    }
    return rc;
}

extern "C" int unet_convT2x2_wgrad_bf16(const void* xin, int ldx, int x_bf16, const void* dz, int lddz, int dz_bf16, float* dw,
                                           int N, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream) {
    return unet_convT2x2_wgrad_bf16_wg(xin, ldx, x_bf16, dz, lddz, dz_bf16, dw, N, H, W, Cin, Cout, 0, ws, ws_bytes, stream);
}

// ---- all weight packs of a step in one launch ---------------------------------------------------------------------------------------
// jobs[njobs][6] int64: {fp32 kernel, forward operand, data-gradient operand, Cin | Cout << 32, kind (0: 3x3, 1: transposed conv),
// first 256-thread block of this job}; a job covers both operands (the two packs of a layer read the same weights).
namespace {

__global__ __launch_bounds__(256) void bf16_pack_batch_kernel(const long long* __restrict__ jobs, int njobs) {
    int j = 0;
    while (j + 1 < njobs && (long long)blockIdx.x >= jobs[(j + 1) * 6 + 5]) ++j;
    const long long* jb = jobs + j * 6;
    const float* w = reinterpret_cast<const float*>(jb[0]);
    uint4* dst0 = reinterpret_cast<uint4*>(jb[1]); uint4* dst1 = reinterpret_cast<uint4*>(jb[2]);
    const int Cin = (int)(jb[3] & 0xffffffffll), Cout = (int)(jb[3] >> 32), kind = (int)jb[4];
    const long i = ((long)blockIdx.x - jb[5]) * 256 + threadIdx.x;
    if (kind == 0) {
        // 3x3: thread = 8 reduce channels of one (tap, output channel); forward reduces over Cin, the data gradient over Cout
        const long n = (long)9 * Cin * Cout / 8;
        if (i >= n) return;
#pragma unroll
        for (int mode = 0; mode < 2; ++mode) {
            const int outc = mode ? Cin : Cout;
            const int o = (int)(i % outc); long rest = i / outc;
            const int kh = (int)(rest % 2); rest /= 2;
            const int tap = (int)(rest % 9); const int chunk = (int)(rest / 9);
            const int k0 = chunk * 16 + kh * 8;
            unsigned v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a, b;
                if (mode == 0) { a = w[((size_t)tap * Cin + k0 + 2 * e) * Cout + o]; b = w[((size_t)tap * Cin + k0 + 2 * e + 1) * Cout + o]; }
                else           { a = w[((size_t)(8 - tap) * Cin + o) * Cout + k0 + 2 * e]; b = w[((size_t)(8 - tap) * Cin + o) * Cout + k0 + 2 * e + 1]; }
                v[e] = cb_pack2(a, b);
            }
            (mode ? dst1 : dst0)[i] = make_uint4(v[0], v[1], v[2], v[3]);
        }
    } else {
        const long n = (long)4 * Cin * Cout / 8;
        if (i >= n) return;
#pragma unroll
        for (int mode = 0; mode < 2; ++mode) {
            const int K = mode ? 4 * Cout : Cin, Ncol = mode ? Cin : 4 * Cout;
            const int col = (int)(i % Ncol); const int k0 = (int)(i / Ncol) * 8;
            unsigned v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = k0 + 2 * e;
                const float a = mode == 0 ? w[(size_t)col * K + k] : w[(size_t)k * Ncol + col];
                const float b = mode == 0 ? w[(size_t)col * K + k + 1] : w[(size_t)(k + 1) * Ncol + col];
                v[e] = cb_pack2(a, b);
            }
            (mode ? dst1 : dst0)[i] = make_uint4(v[0], v[1], v[2], v[3]);
        }
    }
}

}  // namespace

extern "C" int unet_bf16_pack_weights_batch(const void* jobs, int njobs, int total_blocks, void* stream) {
    UNET_CHECK_ARG(jobs && njobs > 0 && total_blocks > 0);
    bf16_pack_batch_kernel<<<(unsigned)total_blocks, 256, 0, (hipStream_t)stream>>>((const long long*)jobs, njobs);
    return UNET_LAUNCH_STATUS();
}
