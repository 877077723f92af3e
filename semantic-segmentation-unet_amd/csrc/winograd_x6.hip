// Fused Winograd F(2x2,3x3) forward / data gradient with the 16 point-products on the BF16 matrix pipe at fp32 grade ("BF16x6").
// Reference layer: UNet._conv_layer, UNet/model.py:28-35 (fp32 in the reference; this is a second ROUTE to the same fp32-grade result,
// beside the v_mfma_f32_32x32x2_f32 kernels of winograd.hip, which stay selectable).
//
// Arithmetic.  An fp32 value is EXACTLY the sum of three bf16 pieces h + m + l (8 + 8 + 8 significant bits, taken by truncation: every
// piece has the sign of the value and the low piece ends where the fp32 mantissa ends).  With both operands of a product split this way,
//     a b  =  ah bh + ah bm + am bh + ah bl + al bh + am bm   +  (am bl + al bm + al bl),
// the six kept piece products are exact in an fp32 accumulator and the three dropped ones are below 2^-24 |a b| each -- the size of
// ONE fp32 multiply's rounding.  Accumulation is fp32 (v_mfma_f32_32x32x16_bf16).  So the result carries the rounding of an fp32
// dot product (measured against fp64: profiles/r03_bf16x6_micro.txt, tests/test_gpu_x6.py), at 6 x 32 matrix-pipe cycles per
// 32 x 32 x 16 block instead of 8 x 64.  The Winograd transforms themselves (B^T d B on the data, G g G^T on the weights, A^T m A on the
// result) stay fp32 vector arithmetic, identical to winograd.hip.
// Not covered: Inf / NaN inputs give NaN (Inf - Inf in the split); values below ~1e-33 lose their low pieces to bf16 underflow.
//
// Tiling.  As winograd.hip: workgroup = 8x8 Winograd tiles x 64 output channels, wave (mi, ni) = [32 channels x 32 tiles] x 16 points in
// 256 accumulator registers, same element order, so the epilogue (wino_epilogue.h) is shared.  The reduction runs in chunks of 16 input
// channels = one MFMA K; a chunk is four UNITS of four Winograd points (one row of the 4x4 point grid), and per unit a wave runs
// 4 points x 6 piece products = 24 MFMAs between two barriers.  LDS (144 KB):
//   D  raw patch 18x18 px x 16 ch fp32, two chunk buffers x 24 KB; pixel slots permuted so that the transform's ds_read_b128 is conflict-free
//   V  [point 4][piece 3][tile 64][16 ch bf16] of one unit, two buffers x 24 KB     <- in-kernel B^T d B + split, every lane one (tile, channel quad)
//   U  [point 4][piece 3][co 64][16 ci bf16]  of one unit, two buffers x 24 KB     <- LDS-DMA from the pre-split weights (L2-resident)
// Pipeline, everything one unit ahead of its use: during unit g the wave issues the DMA of U(g+1) and its share of a later D chunk, turns
// the row-stage registers of unit g+1 into V(g+1) (column stage + split + 12 LDS writes), and reads the raw rows of unit g+2 from D and
// row-stages them in place.  Unlike the fp32 MFMA, the bf16 MFMA leaves the vector ALU free: ~5 vector instructions are slotted behind
// every MFMA (sched_barrier-pinned), the stream is vector-issue-bound rather than matrix-bound.
#include "common.h"
#include "wino_epilogue.h"
#include <cstdlib>

#ifndef UNET_X6_ABLATE
#define UNET_X6_ABLATE 0        /* diagnostic builds (scripts/build_variant.sh), bits: 8 = s_memtime stamps around the phases of a unit, 16 = no split, 32 = no MFMAs, 64 = no DMA, 128 = no operand reads, 256 = no V writes, 512 = no row reads (results wrong) */
#endif

namespace {

#if (UNET_X6_ABLATE & 8)
__device__ long long g_x6_timeline[16];
__device__ __forceinline__ void x6_tl_add(int i, long long v) { if (blockIdx.x == 0 && (threadIdx.x & 255) == 0) g_x6_timeline[i] += v; }
#define X6_STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory")
#else
#define X6_STAMP(t)
#endif

typedef int x6_i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned x6_u32x2 __attribute__((ext_vector_type(2)));

constexpr int kX6DB = 21 * 1024;                  // one D chunk buffer: 324 pixel slots x 64 B in 21 1-KB DMA pieces (the last one a quarter used)
constexpr int kX6IB = 24 * 1024;                  // one V or U unit image
constexpr int kX6Blk = 2048;                      // one (point, piece) block: 64 rows x 32 B
// LDS map: D0 D1 (21 KB each) | V0 U0 | V1 U1 | 16 KB spare.  The epilogue's exchange region X = V1 U1 spare (64 KB): idle at a tile's end (the last unit's parity is 1).
constexpr int kX6V = 2 * kX6DB, kX6U = kX6V + kX6IB, kX6Par = 2 * kX6IB, kX6X = kX6V + kX6Par, kX6Smem = kX6X + 64 * 1024;      // 154 KB
constexpr int kX6RowB = 18 * 64;                  // bytes between patch rows in D

#define X6_RD128(dst, base, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst) : "v"(base), "n"(off))
#define X6_WR2(base, o0, o1, v0, v1) asm volatile("ds_write2st64_b64 %0, %1, %2 offset0:%3 offset1:%4" : : "v"(base), "v"(v0), "v"(v1), "n"(o0), "n"(o1) : "memory")
#define X6_WR64(base, off, val) asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(base), "v"(val), "n"(off) : "memory")
#define X6_MFMA(accv, av, bv) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(accv) : "v"(av), "v"(bv) : "memory")
// LDS-DMA of one 1-KB piece (16 B per lane) to LDS byte address `ldsaddr` (wave-uniform): M0 carries the LDS address.  Written as asm so
// that the U pieces take the SGPR-base + 32-bit lane offset form (no per-piece 64-bit vector add); every DMA of this file goes through
// these two macros, so the compiler never manages M0 itself here.
// (ldsw = the wave's LDS base in ONE scalar register, ldsoff an immediate: the sum is formed in M0 itself)
#define X6_DMA_S(voff, sbase, ldsw, ldsoff) asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(ldsw), "n"(ldsoff) : "memory", "scc")
#define X6_DMA_V(vptr, ldsw, ldsoff) asm volatile("s_add_u32 m0, %1, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(vptr), "s"(ldsw), "n"(ldsoff) : "memory", "scc")
#define X6_MFMA0(accv, av, bv) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(accv) : "v"(av), "v"(bv) : "memory")

// slot of patch column x (0..17) inside a patch row: pixels two apart (the stride between neighbouring tiles) must land on different
// 64-byte bank quarters for the hardware's ds_read_b128 lane groups {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31}: slot mod 4 = (x/2 + x) mod 4
__host__ __device__ constexpr int x6_slot_of(int x) { return x >= 16 ? x : (x & ~7) + ((x & 7) == 0 ? 0 : (x & 7) == 1 ? 1 : (x & 7) == 2 ? 5 : (x & 7) == 3 ? 2 : (x & 7) == 4 ? 6 : (x & 7) == 5 ? 3 : (x & 7) == 6 ? 7 : 4); }
__host__ __device__ constexpr int x6_col_of(int s) { return s >= 16 ? s : (s & ~7) + ((s & 7) == 0 ? 0 : (s & 7) == 1 ? 1 : (s & 7) == 2 ? 3 : (s & 7) == 3 ? 5 : (s & 7) == 4 ? 7 : (s & 7) == 5 ? 2 : (s & 7) == 6 ? 4 : 6); }

struct X6Frag { x6_i32x4 u[3], v[3]; };           // MFMA operands of one point: weight pieces (h, m, l), data pieces (h, m, l)
struct X6Split { float v[4], a[4], b[4]; unsigned h[2], m[2], l[2]; };

// operand reads of point PT of the unit in buffers PAR: 6 ds_read_b128
template <int PAR, int PT> __device__ __forceinline__ void x6_read_ops(X6Frag& f, unsigned a_base, unsigned b_base) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        X6_RD128(f.u[k], b_base, PAR * kX6IB + (PT * 3 + k) * kX6Blk);
        X6_RD128(f.v[k], a_base, PAR * kX6IB + (PT * 3 + k) * kX6Blk);
    }
}
#define X6_TIE_FRAG(f) "+v"(f.u[0]), "+v"(f.u[1]), "+v"(f.u[2]), "+v"(f.v[0]), "+v"(f.v[1]), "+v"(f.v[2])

__device__ __forceinline__ unsigned x6_hi2(float lo, float hi) {       // { bf16 bits of lo (truncated) , of hi } packed, lo in the low half
    return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, hi), __builtin_bit_cast(unsigned, lo), 0x07060302u);
}
__device__ __forceinline__ float x6_trunc(float v) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & 0xffff0000u); }

// Column stage + split of one point in five steps of 5-6 vector instructions; tt[c] = the row-stage result of the wave's patch column c
// (4 channels).  The four points of a unit's row: V[.][0] = t0 - t2, [1] = t1 + t2, [2] = t2 - t1, [3] = t1 - t3.
// X6_PIN: an empty volatile asm over a step's inputs / results.  Instruction selection orders pure arithmetic freely between the volatile
// MFMAs (sched_barrier only binds the machine scheduler); tied to a volatile statement on both sides a step stays in its gap.
#define X6_PIN(...) asm volatile("" : __VA_ARGS__)
template <int K, int TA, int TB, bool ADD> __device__ __forceinline__ void x6_split_step(X6Split& s, f32x4 (&tt)[6]) {
#if (UNET_X6_ABLATE & 16)
    if (K) return;
#endif
    if constexpr (K == 0) {
        X6_PIN("+v"(tt[TA]), "+v"(tt[TB]));
        const f32x4 vv = ADD ? tt[TA] + tt[TB] : tt[TA] - tt[TB];
        s.v[0] = vv[0]; s.v[1] = vv[1]; s.v[2] = vv[2]; s.v[3] = vv[3];
        s.h[0] = x6_hi2(s.v[0], s.v[1]);
        X6_PIN("+v"(s.v[0]), "+v"(s.v[1]), "+v"(s.v[2]), "+v"(s.v[3]), "+v"(s.h[0]));
    } else if constexpr (K == 1) {
        s.a[0] = s.v[0] - x6_trunc(s.v[0]); s.a[1] = s.v[1] - x6_trunc(s.v[1]);
        s.h[1] = x6_hi2(s.v[2], s.v[3]);
        X6_PIN("+v"(s.a[0]), "+v"(s.a[1]), "+v"(s.h[1]));
    } else if constexpr (K == 2) {
        s.a[2] = s.v[2] - x6_trunc(s.v[2]); s.a[3] = s.v[3] - x6_trunc(s.v[3]);
        s.m[0] = x6_hi2(s.a[0], s.a[1]);
        X6_PIN("+v"(s.a[2]), "+v"(s.a[3]), "+v"(s.m[0]));
    } else if constexpr (K == 3) {
        s.b[0] = s.a[0] - x6_trunc(s.a[0]); s.b[1] = s.a[1] - x6_trunc(s.a[1]);
        s.m[1] = x6_hi2(s.a[2], s.a[3]);
        X6_PIN("+v"(s.b[0]), "+v"(s.b[1]), "+v"(s.m[1]));
    } else {
        s.b[2] = s.a[2] - x6_trunc(s.a[2]); s.b[3] = s.a[3] - x6_trunc(s.a[3]);
        s.l[0] = x6_hi2(s.b[0], s.b[1]); s.l[1] = x6_hi2(s.b[2], s.b[3]);
        X6_PIN("+v"(s.l[0]), "+v"(s.l[1]));
    }
}
// the wave's two points: J0 = 2 PH, J0 + 1; local patch columns 0..2 = global PH .. PH + 2
template <int K, int Q, int PH> __device__ __forceinline__ void x6_point_step(X6Split& s, f32x4 (&tt)[6]) {
    constexpr int J = 2 * PH + Q;
    constexpr int GA = J == 0 ? 0 : J == 2 ? 2 : 1, GB = J == 0 ? 2 : J == 1 ? 2 : J == 2 ? 1 : 3;      // global columns: tA -/+ tB
    x6_split_step<K, GA - PH, GB - PH, J == 1>(s, tt);
}

// raw rows of the unit with point row R2: rows ra, rb of the patch; tt = ra -/+ rb:   R2 = 0: d0 - d2;  1: d1 + d2;  2: d2 - d1;  3: d1 - d3
template <int R2> struct X6Rows {
    static constexpr int RA = R2 == 0 ? 0 : R2 == 2 ? 2 : 1, RB = R2 == 0 ? 2 : R2 == 1 ? 2 : R2 == 2 ? 1 : 3;
    static constexpr bool ADD = R2 == 1;
};
// the wave's three patch columns of both rows from D buffer DPR: dd[c] (row ra), dd[3 + c] (row rb); six ds_read_b128
template <int R2, int DPR> __device__ __forceinline__ void x6_read_rows(f32x4 (&dd)[6], const unsigned (&d_base)[3]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        X6_RD128(dd[c], d_base[c], DPR * kX6DB + X6Rows<R2>::RA * kX6RowB);
        X6_RD128(dd[3 + c], d_base[c], DPR * kX6DB + X6Rows<R2>::RB * kX6RowB);
    }
}
#define X6_TIE_DD(d) "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5])
template <int R2> __device__ __forceinline__ void x6_row_stage(f32x4 (&dd)[6]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) dd[c] = X6Rows<R2>::ADD ? dd[c] + dd[3 + c] : dd[c] - dd[3 + c];
    X6_PIN("+v"(dd[0]), "+v"(dd[1]), "+v"(dd[2]));
}

// ---- one unit of one wave ------------------------------------------------------------------------------------------------------------------
// Eight waves; SIMD partners (wq, PH = 0 / 1) split every unit's four points (a row of the 4x4 point grid): the wave multiplies points
// J0 = 2 PH, J0 + 1 into accumulators 2 R + q and transforms those two points' V of the NEXT unit for its 16 tiles x channel quad.  Both
// partners run the same kind of stream, so one's LDS-write / DMA-issue / wait stalls are the other's issue slots, and the vector work
// issues from two waves (2 cycles per instruction instead of one wave's 4).
//   12 MFMAs:  point q:  (h,h) (h,m) (m,h) (m,m) (h,l) (l,h)   [weights piece, data piece]; operand pieces read one by one, >= 3 MFMAs ahead;
//   transform: raw rows of unit g+1 (6 reads at the top) -> row stage behind the 3rd MFMA -> five split steps per point, one per MFMA ->
//              three ds_write2st64_b64;   DMA: three pieces of U(g+1), one D piece (R != 2).
// LDS instructions retire in order: each lgkmcnt = the LDS instructions issued behind the piece waited for.
#define X6_RDU(dst, J, PC) X6_RD128(dst, b_base, ((J) * 3 + PC) * kX6Blk)
#define X6_RDV(dst, J, PC) X6_RD128(dst, a_base, ((J) * 3 + PC) * kX6Blk)
template <int R, int DP, bool FIRST, int PH>
__device__ __forceinline__ void x6_unit(f32x16 (&acc)[8], unsigned a_base0, unsigned b_base0, const unsigned (&d_base)[3], unsigned v_base0,
                                        const char* us, size_t ublk, unsigned u_lane, const float* dsrc, bool has_d, unsigned lds_w) {
    constexpr int P = R & 1, PN = P ^ 1, J0 = 2 * PH;
    constexpr int R1 = (R + 1) & 3, DPR = DP ^ (R == 3 ? 1 : 0);      // rows of unit g+1
    // D piece of this unit: R = 3: chunk c+2, piece w8; R = 0: chunk c+1, piece 8 + w8; R = 1: chunk c+1, piece 16 + w8 (w8 < 5); R = 2: none
    constexpr int DPC = R == 3 ? 0 : R == 0 ? 8 : 16;
    constexpr int DPW = R == 3 ? DP : (DP ^ 1);
    const unsigned a_base = a_base0 + P * kX6Par, b_base = b_base0 + P * kX6Par, v_base = v_base0 + PN * kX6Par;
    x6_i32x4 uh[2], vh[2], vm[2], um[2], vl[2], ul[2];               // [point]
    f32x4 dd[6];
    X6Split sp;
    unsigned lA[2];
    X6_RDU(uh[0], J0, 0); X6_RDV(vh[0], J0, 0); X6_RDV(vm[0], J0, 1); X6_RDU(um[0], J0, 1);
    x6_read_rows<R1, DPR>(dd, d_base);
    f32x16& A0 = acc[2 * R], &A1 = acc[2 * R + 1];
#define X6_M(A, u, v) X6_MFMA(A, u, v)
    // ---- point 0
    asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(uh[0]), "+v"(vh[0]));
    if (FIRST) X6_MFMA0(A0, uh[0], vh[0]); else X6_M(A0, uh[0], vh[0]);
    X6_RDV(vl[0], J0, 2);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(vm[0]));
    X6_M(A0, uh[0], vm[0]);
    X6_RDU(ul[0], J0, 2);
    if (!(UNET_X6_ABLATE & 64)) X6_DMA_S(u_lane, us, lds_w, kX6U + PN * kX6Par);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(um[0]));
    X6_M(A0, um[0], vh[0]);
    X6_RDU(uh[1], J0 + 1, 0); X6_RDV(vh[1], J0 + 1, 0);
    asm volatile("s_waitcnt lgkmcnt(4)" : X6_TIE_DD(dd));                 // the six row reads (behind them: v_l, u_l, u_h', v_h')
    x6_row_stage<R1>(dd);
    __builtin_amdgcn_sched_barrier(0);
    X6_M(A0, um[0], vm[0]);
    X6_RDV(vm[1], J0 + 1, 1);
    x6_point_step<0, 0, PH>(sp, dd);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(vl[0]));
    X6_M(A0, uh[0], vl[0]);
    X6_RDU(um[1], J0 + 1, 1);
    if (!(UNET_X6_ABLATE & 64)) X6_DMA_S(u_lane, us + ublk, lds_w, kX6U + PN * kX6Par + 8192);
    x6_point_step<1, 0, PH>(sp, dd);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(ul[0]));
    X6_M(A0, ul[0], vh[0]);
    x6_point_step<2, 0, PH>(sp, dd);
    __builtin_amdgcn_sched_barrier(0);
    // ---- point 1
    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(uh[1]), "+v"(vh[1]));
    if (FIRST) X6_MFMA0(A1, uh[1], vh[1]); else X6_M(A1, uh[1], vh[1]);
    X6_RDV(vl[1], J0 + 1, 2);
    x6_point_step<3, 0, PH>(sp, dd);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(vm[1]));
    X6_M(A1, uh[1], vm[1]);
    X6_RDU(ul[1], J0 + 1, 2);
    x6_point_step<4, 0, PH>(sp, dd);
    if (!(UNET_X6_ABLATE & 256)) X6_WR2(v_base, (J0 * 3 + 0) * 4, (J0 * 3 + 1) * 4, (x6_u32x2{sp.h[0], sp.h[1]}), (x6_u32x2{sp.m[0], sp.m[1]}));
    lA[0] = sp.l[0]; lA[1] = sp.l[1];
    if (!(UNET_X6_ABLATE & 64)) X6_DMA_S(u_lane, us + 2 * ublk, lds_w, kX6U + PN * kX6Par + 2 * 8192);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(um[1]));
    X6_M(A1, um[1], vh[1]);
    x6_point_step<0, 1, PH>(sp, dd);
    __builtin_amdgcn_sched_barrier(0);
    X6_M(A1, um[1], vm[1]);
    x6_point_step<1, 1, PH>(sp, dd);
    if (R != 2 && has_d && !(UNET_X6_ABLATE & 64)) X6_DMA_V(dsrc, lds_w, DPW * kX6DB + DPC * 1024);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(vl[1]));
    X6_M(A1, uh[1], vl[1]);
    x6_point_step<2, 1, PH>(sp, dd);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(ul[1]));
    X6_M(A1, ul[1], vh[1]);
    x6_point_step<3, 1, PH>(sp, dd);
    x6_point_step<4, 1, PH>(sp, dd);
    if (!(UNET_X6_ABLATE & 256)) {
        X6_WR2(v_base, ((J0 + 1) * 3 + 0) * 4, ((J0 + 1) * 3 + 1) * 4, (x6_u32x2{sp.h[0], sp.h[1]}), (x6_u32x2{sp.m[0], sp.m[1]}));
        X6_WR2(v_base, (J0 * 3 + 2) * 4, ((J0 + 1) * 3 + 2) * 4, (x6_u32x2{lA[0], lA[1]}), (x6_u32x2{sp.l[0], sp.l[1]}));
    }
#undef X6_M
    // this unit's D piece stays in flight (needed two units later at the earliest)
#if (UNET_X6_ABLATE & 2048)      /* diagnostics (results wrong): no waits at the end of a unit, barrier only */
    asm volatile("s_barrier" ::: "memory");
#elif (UNET_X6_ABLATE & 4096)    /* diagnostics (results wrong): no barrier either */
    asm volatile("" ::: "memory");
#else
    if (R != 2 && has_d) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

struct X6Args {
    WinoFusedArgs f;             // x, bias, out, geometry, stats, pad; f.Uc unused
    const uint16_t* U6;          // [K/16][unit 4][point 4][piece 3][Nout][16] bf16
};

// ---- epilogue of the wave pair ------------------------------------------------------------------------------------------------------------
// SIMD partners (mi, ni, PH = 0 / 1) hold point COLUMNS {0, 1} / {2, 3} of the same [32 channels x 32 tiles] block, all four rows.  The
// output transform's row stage is lane-local; its column stage is linear in the columns, so each partner forms its partial 2x2 output
//     PH = 0:  y[.][0] = rr0 + rr1,  y[.][1] = rr1          PH = 1:  y[.][0] = rr2,  y[.][1] = -rr2 - rr3
// hands the two channel quads it does not finish to the other one through LDS (8 KB per wave, 64 KB in the buffers that are idle at a
// tile's end), adds what it receives and finishes its own two quads: bias, ReLU, BatchNorm sums, stores.  The PH = 1 wave reads its
// weight rows rotated by 16 (x6_group_body), so in BOTH waves accumulator elements 0..7 (quads 0, 1) are the channels the wave finishes
// itself -- 32 ni + 16 PH + 8 q + 4 lh + {0..3} -- and elements 8..15 the ones it hands over.
template <int G, int PH> __device__ __forceinline__ void x6_partial_quad(const f32x16 (&acc)[8], f32x4 (&y)[4]) {
    f32x2 yy[2][2][2];                            // [out row][out col][channel pair]
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        f32x2 rr[2][2];                           // [out row][the wave's column q]
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            f32x2 m[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // explicit accumulator reads (element extraction left to the compiler round-trips whole accumulators through VGPRs)
                float e0, e1;
                asm("v_accvgpr_read_b32 %0, %1" : "=v"(e0) : "a"(acc[2 * i + q][4 * G + 2 * h]));
                asm("v_accvgpr_read_b32 %0, %1" : "=v"(e1) : "a"(acc[2 * i + q][4 * G + 2 * h + 1]));
                m[i] = f32x2{e0, e1};
            }
            rr[0][q] = m[0] + (m[1] + m[2]);
            rr[1][q] = (m[1] - m[2]) - m[3];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (PH == 0) { yy[i][0][h] = rr[i][0] + rr[i][1]; yy[i][1][h] = rr[i][1]; }
            else { yy[i][0][h] = rr[i][0]; yy[i][1][h] = -rr[i][0] - rr[i][1]; }
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) y[2 * i + j] = f32x4{yy[i][j][0].x, yy[i][j][0].y, yy[i][j][1].x, yy[i][j][1].y};
}
#define X6_WR128_RT(base, off, val) asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(base), "v"(val), "n"(off) : "memory")

template <int STATS, int PH>
__device__ __forceinline__ void x6_pair_epilogue(const f32x16 (&acc)[8], const WinoFusedArgs& p, int img, int by, int bx, int n0, int mi, int ni,
                                                 int wq, int li, int lh, unsigned xbase, f32x2 (&s1)[4], f32x2 (&s2)[4]) {
    constexpr int kq0 = 2 * PH;                    // kept channel quads (of the wave pair's 32 channels): kq0, kq0 + 1
    const unsigned xw = xbase + (unsigned)((wq * 2 + PH) * 8192), xr = xbase + (unsigned)((wq * 2 + (PH ^ 1)) * 8192);
    const int lt = 32 * mi + li;
    const int ty = 8 * by + (lt >> 3), tx = 8 * bx + (lt & 7);
    const bool ok = ty < (p.H >> 1) && tx < (p.W >> 1);
    float* o = p.out + ((size_t)(img * p.H + 2 * ty) * p.W + 2 * tx) * p.ldo + n0 + 32 * ni + 4 * lh + 8 * kq0;
    const size_t rowstride = (size_t)p.W * p.ldo;
    {
        f32x4 y[4];
        x6_partial_quad<2, PH>(acc, y);
#pragma unroll
        for (int px = 0; px < 4; ++px) X6_WR128_RT(xw, px * 1024, y[px]);
        x6_partial_quad<3, PH>(acc, y);
#pragma unroll
        for (int px = 0; px < 4; ++px) X6_WR128_RT(xw, (4 + px) * 1024, y[px]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const float lo = p.relu ? 0.f : -__builtin_inff();
    const bool rok = STATS == 2 && ok && n0 >= p.bn_c0 && n0 < p.bn_c1;
    const float* r = STATS == 2 ? p.bn_r + ((size_t)(img * p.H + 2 * ty) * p.W + 2 * tx) * p.bn_ldr + (n0 - p.bn_c0) + 32 * ni + 4 * lh + 8 * kq0 : nullptr;
#pragma unroll
    for (int q = 0; q < 2; ++q) {                   // one kept quad at a time: partner's half from LDS, own half from the accumulators
        f32x4 yp[4], rv[4];
#pragma unroll
        for (int px = 0; px < 4; ++px) X6_RD128(yp[px], xr, (4 * q + px) * 1024);
        f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias) bias4 = *reinterpret_cast<const f32x4*>(p.bias + n0 + 32 * ni + 4 * lh + 8 * (kq0 + q));
        if (STATS == 2) {
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                rv[px] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (rok) rv[px] = *reinterpret_cast<const f32x4*>(r + ((size_t)(px >> 1) * p.W + (px & 1)) * p.bn_ldr + 8 * q);
            }
        }
        f32x4 yk[4];
        if (q == 0) x6_partial_quad<0, PH>(acc, yk); else x6_partial_quad<1, PH>(acc, yk);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(yp[0]), "+v"(yp[1]), "+v"(yp[2]), "+v"(yp[3]));
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            f32x4 v = (yk[px] + yp[px]) + bias4;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], lo);
            if (ok) {
                *reinterpret_cast<f32x4*>(o + (size_t)(px >> 1) * rowstride + (size_t)(px & 1) * p.ldo + 8 * q) = v;
                if (STATS == 1) {
                    s1[2 * q] += f32x2{v[0], v[1]}; s1[2 * q + 1] += f32x2{v[2], v[3]};
                    s2[2 * q] += f32x2{v[0] * v[0], v[1] * v[1]}; s2[2 * q + 1] += f32x2{v[2] * v[2], v[3] * v[3]};
                }
                if (STATS == 2) {
                    s1[2 * q] += f32x2{v[0], v[1]}; s1[2 * q + 1] += f32x2{v[2], v[3]};
                    s2[2 * q] += f32x2{v[0] * rv[px][0], v[1] * rv[px][1]}; s2[2 * q + 1] += f32x2{v[2] * rv[px][2], v[3] * rv[px][3]};
                }
            }
        }
    }
    asm volatile("s_barrier" ::: "memory");        // every partner read is done: the exchange buffers are the next unit's V / U images
}

// per-lane running sums -> the wave's 16 channels of its row of partials; layout as wf_write_stats: stat_part[tn][row][64 channels][2]
__device__ __forceinline__ void x6_write_stats(const WinoFusedArgs& p, int t0, int rows_per_tn, int mi, int ni, int ph, int li, int lh,
                                               f32x2 (&s1)[4], f32x2 (&s2)[4]) {
    float v[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[4 * i] = s1[i].x; v[4 * i + 1] = s1[i].y; v[4 * i + 2] = s2[i].x; v[4 * i + 3] = s2[i].y; }
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int m = 1; m < 32; m <<= 1) v[i] += __shfl_xor(v[i], m, 32);
    if (li != 0) return;
    const int tn = t0 % p.nt, row = 2 * (t0 / p.nt) + mi;
    float* o = p.stat_part + ((size_t)tn * rows_per_tn + row) * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i) {                       // pair i = 2 q + h -> channels 32 ni + 8 (2 ph + q) + 4 lh + 2 h + {0, 1}
        const int ch = 32 * ni + 8 * (2 * ph + (i >> 1)) + 4 * lh + 2 * (i & 1);
        o[2 * ch] = v[4 * i]; o[2 * ch + 1] = v[4 * i + 2]; o[2 * ch + 2] = v[4 * i + 1]; o[2 * ch + 3] = v[4 * i + 3];
    }
}

// The body of one wave group.  PH (waves 4 PH .. 4 PH + 3) is a template parameter: the two groups run different straight-line code.
template <int STATS, int PH>
__device__ __forceinline__ void x6_group_body(const X6Args& q, int ntiles, char* smem) {
    const WinoFusedArgs& p = q.f;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wq = w8 & 3;                                      // SIMD partners are waves wq and wq + 4
    const int mi = wq & 1, ni = wq >> 1;
    const int li = lane & 31, lh = lane >> 5;
    const int nchunks = p.K / 16;

    // ---- DMA duty.  U: piece w8 + 8 j (j = 0..2) of a unit image = block (wq >> 1) + 2 PH + 4 j, rows 32 (wq & 1) + lane / 2, 16-byte slot
    //      lane & 1 (source-side swizzle: slot ^ bit 3 of the row).  D: piece = pixel slots 16 piece + lane / 4, channel quad lane & 3; the
    //      wave owns pieces w8 (issued in units R = 3), 8 + w8 (R = 0) and 16 + w8 (R = 1; exists for w8 < 5).
    const int urow = 32 * (wq & 1) + (lane >> 1);
    const unsigned u_lane = (unsigned)(urow * 32 + 16 * ((lane & 1) ^ ((urow >> 3) & 1)));
    const size_t ublk = (size_t)4 * p.Nout * 32;                                      // blocks b and b + 4 are 4 N rows apart
    const size_t ustep = (size_t)12 * p.Nout * 32;                                    // bytes between units
    auto piece_geom = [&](int piece, int& py_, int& px_, int& off_) {
        const int s_ = 16 * piece + (lane >> 2);
        const int py = s_ / 18, px = x6_col_of(s_ % 18);
        py_ = s_ < 324 ? py : (1 << 20);                                              // past the patch: never inside the image
        px_ = px;
        off_ = (py * p.W + px) * p.ldx + 4 * (lane & 3);
    };
    struct TileCoord { int tn, bx, by, img; };
    auto decode = [&](int t) { TileCoord c; c.tn = t % p.nt; t /= p.nt; c.bx = t % p.tbx; t /= p.tbx; c.by = t % p.tby; c.img = t / p.tby; return c; };
    const TileCoord dstep = decode((int)gridDim.x);
    auto advance = [&](TileCoord c) {
        c.tn += dstep.tn; int cy = c.tn >= p.nt; c.tn -= cy ? p.nt : 0;
        c.bx += dstep.bx + cy; cy = c.bx >= p.tbx; c.bx -= cy ? p.tbx : 0;
        c.by += dstep.by + cy; cy = c.by >= p.tby; c.by -= cy ? p.tby : 0;
        c.img += dstep.img + cy;
        return c;
    };
    const float* const padsrc = p.pad ? p.pad : g_zero_page_f;
    auto pixel_src = [&](const TileCoord& c, int py, int px, int off) {
        const int gy0 = 16 * c.by - 1, gx0 = 16 * c.bx - 1;
        const float* xb = p.x + ((long long)(c.img * p.H + gy0) * p.W + gx0) * p.ldx;
        const bool ok = (unsigned)(gy0 + py) < (unsigned)p.H && (unsigned)(gx0 + px) < (unsigned)p.W;
        return ok ? xb + off : padsrc + 4 * (lane & 3);
    };
    auto slot_src = [&](const TileCoord& c, int j) { int py, px, off; piece_geom(8 * j + w8, py, px, off); return pixel_src(c, py, px, off); };
    auto u_source = [&](const TileCoord& c) { return reinterpret_cast<const char*>(q.U6) + ((size_t)((wq >> 1) + 2 * PH) * p.Nout + (size_t)c.tn * 64) * 32; };
    const bool has2 = w8 < 5;                                                         // piece 16 + w8 exists

    // ---- LDS byte addresses
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_f*)smem;
    const int arow = 32 * mi + li;
    const int t_lt = 16 * wq + (lane >> 2), t_q = lane & 3;                      // transform duty: (tile, channel quad), points 2 PH, 2 PH + 1
    const unsigned a_base = lds0 + kX6V + (unsigned)(arow * 32 + 16 * (lh ^ ((arow >> 3) & 1)));          // parity 0; parity 1 is kX6Par further
    // (the PH = 1 partner takes its 32 weight rows rotated by 16: see x6_pair_epilogue)
    const int brow_r = 32 * ni + ((li + 16 * PH) & 31);
    const unsigned b_base = lds0 + kX6U + (unsigned)(brow_r * 32 + 16 * (lh ^ ((brow_r >> 3) & 1)));
    const unsigned v_base = lds0 + kX6V + (unsigned)(t_lt * 32 + 16 * ((t_q >> 1) ^ ((t_lt >> 3) & 1)) + 8 * (t_q & 1));
    unsigned d_base[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
        d_base[c] = lds0 + (unsigned)(((2 * (t_lt >> 3)) * 18 + x6_slot_of(2 * (t_lt & 7) + PH + c)) * 64 + 16 * t_q);
    const unsigned lds_w = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + w8 * 1024));       // this wave's first piece, as an M0 value

    f32x16 acc[8];
    const float* dptr[3]; const char* ucur; const char* unxt;
    int t = blockIdx.x;
    if ((gridDim.x & 7) == 0 && (p.nt & 7) != 0) t = (t & 7) * (int)(gridDim.x >> 3) + (t >> 3);       // XCD-aware renumbering, as winograd.hip
    const int t_first = t;
    f32x2 s1[4], s2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { s1[i] = f32x2{0.f, 0.f}; s2[i] = f32x2{0.f, 0.f}; }
    TileCoord tc = decode(t);
#pragma unroll
    for (int j = 0; j < 3; ++j) dptr[j] = slot_src(tc, j);
    ucur = u_source(tc);

    // ---- prologue of the workgroup's first tile: D(0) (every piece), D(1) pieces 0..7, U(unit 0) -> LDS; then V(unit 0)
    X6_DMA_V(dptr[0], lds_w, 0); X6_DMA_V(dptr[1], lds_w, 8192);
    if (has2) X6_DMA_V(dptr[2], lds_w, 16384);
    X6_DMA_V(dptr[0] + 16, lds_w, kX6DB);
    dptr[0] += 32; dptr[1] += 16; dptr[2] += 16;                              // next: chunk 2 (piece w8), chunk 1 (pieces 8 + w8, 16 + w8)
#pragma unroll
    for (int j = 0; j < 3; ++j) X6_DMA_S(u_lane, ucur + (size_t)j * ublk, lds_w, kX6U + j * 8192);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    {
        f32x4 dd[6]; X6Split sp; unsigned lA[2];
        x6_read_rows<0, 0>(dd, d_base);
        asm volatile("s_waitcnt lgkmcnt(0)" : X6_TIE_DD(dd));
        x6_row_stage<0>(dd);
        x6_point_step<0, 0, PH>(sp, dd); x6_point_step<1, 0, PH>(sp, dd); x6_point_step<2, 0, PH>(sp, dd); x6_point_step<3, 0, PH>(sp, dd); x6_point_step<4, 0, PH>(sp, dd);
        X6_WR2(v_base, (2 * PH * 3 + 0) * 4, (2 * PH * 3 + 1) * 4, (x6_u32x2{sp.h[0], sp.h[1]}), (x6_u32x2{sp.m[0], sp.m[1]}));
        lA[0] = sp.l[0]; lA[1] = sp.l[1];
        x6_point_step<0, 1, PH>(sp, dd); x6_point_step<1, 1, PH>(sp, dd); x6_point_step<2, 1, PH>(sp, dd); x6_point_step<3, 1, PH>(sp, dd); x6_point_step<4, 1, PH>(sp, dd);
        X6_WR2(v_base, ((2 * PH + 1) * 3 + 0) * 4, ((2 * PH + 1) * 3 + 1) * 4, (x6_u32x2{sp.h[0], sp.h[1]}), (x6_u32x2{sp.m[0], sp.m[1]}));
        X6_WR2(v_base, (2 * PH * 3 + 2) * 4, ((2 * PH + 1) * 3 + 2) * 4, (x6_u32x2{lA[0], lA[1]}), (x6_u32x2{sp.l[0], sp.l[1]}));
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    for (; t < ntiles; t += gridDim.x) {
        const TileCoord tcn = t + (int)gridDim.x < ntiles ? advance(tc) : tc;            // the last tile prefetches itself again
        unxt = u_source(tcn);
        // the D pointer of the slot a unit issued moves on behind it: one chunk further, or to the next tile's patch behind the tile's last chunk
        auto advance_slot = [&](int s0, bool last) { if (last) dptr[s0] = slot_src(tcn, s0); else dptr[s0] += 16; };
        // U(g + 1) of unit g = 4 c + R, continuing into the next tile.  D pieces: the unit R = 3 issues chunk c + 2 (the tile's last one when
        // c = nchunks - 3), R = 0, 1 issue chunk c + 1 (the last one when c = nchunks - 2); behind the last chunk the pointer jumps to the next tile
#if (UNET_X6_ABLATE & 1024)      /* diagnostics: every U DMA re-reads the first unit's weights (always cache-hot) */
#define X6_US(c, R) (ucur)
#else
#define X6_US(c, R) ((4 * (c) + (R) + 1 < 4 * nchunks) ? ucur + (size_t)(4 * (c) + (R) + 1) * ustep : unxt)
#endif
#define X6_UNIT(R, DP, FIRST, c) do { \
        x6_unit<R, DP, FIRST, PH>(acc, a_base, b_base, d_base, v_base, X6_US(c, R), ublk, u_lane, dptr[(R) == 3 ? 0 : (R) == 0 ? 1 : 2], (R) != 1 || has2, lds_w); \
        if ((R) == 3) advance_slot(0, (c) == nchunks - 3); \
        if ((R) == 0) advance_slot(1, (c) == nchunks - 2); \
        if ((R) == 1) advance_slot(2, (c) == nchunks - 2); } while (0)
        X6_UNIT(0, 0, true, 0); X6_UNIT(1, 0, true, 0); X6_UNIT(2, 0, true, 0); X6_UNIT(3, 0, true, 0);
        X6_UNIT(0, 1, false, 1); X6_UNIT(1, 1, false, 1); X6_UNIT(2, 1, false, 1); X6_UNIT(3, 1, false, 1);
        for (int c = 2; c < nchunks; c += 2) {
            X6_UNIT(0, 0, false, c); X6_UNIT(1, 0, false, c); X6_UNIT(2, 0, false, c); X6_UNIT(3, 0, false, c);
            X6_UNIT(0, 1, false, c + 1); X6_UNIT(1, 1, false, c + 1); X6_UNIT(2, 1, false, c + 1); X6_UNIT(3, 1, false, c + 1);
        }
#undef X6_UNIT
#undef X6_US
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // inline-asm MFMAs are invisible to the compiler's hazard recogniser
        x6_pair_epilogue<STATS, PH>(acc, p, tc.img, tc.by, tc.bx, tc.tn * 64, mi, ni, wq, li, lh, lds0 + kX6X + (unsigned)lane * 16, s1, s2);
        ucur = unxt; tc = tcn;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          // retire the prefetches of the tile that never runs
    if (STATS) x6_write_stats(p, t_first, 2 * ((int)gridDim.x / p.nt), mi, ni, PH, li, lh, s1, s2);
}
template <int STATS>
__device__ __forceinline__ void x6_stream_body(const X6Args& q, int ntiles) {
    __shared__ __attribute__((aligned(1024))) char smem[kX6Smem];
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 8) == 0) x6_group_body<STATS, 0>(q, ntiles, smem);
    else x6_group_body<STATS, 1>(q, ntiles, smem);
}
__global__ __launch_bounds__(512, 2) void wino_x6_stream_kernel(X6Args q, int ntiles) { x6_stream_body<0>(q, ntiles); }
__global__ __launch_bounds__(512, 2) void wino_x6_stream_stats_kernel(X6Args q, int ntiles) { x6_stream_body<1>(q, ntiles); }
__global__ __launch_bounds__(512, 2) void wino_x6_stream_bnbwd_kernel(X6Args q, int ntiles) { x6_stream_body<2>(q, ntiles); }

// ---- weight operands: G g G^T in fp32 (as winograd.hip), then the exact three-piece split, in the kernel's unit layout -----------------
//   U6[((((k/16) * 4 + r) * 4 + j) * 3 + piece) * N + n) * 16 + k % 16],   point xi = 4 r + j
// mode 0 (forward): k = ci, n = co;  thread = 8 consecutive ci x one co (adjacent threads = adjacent co: coalesced reads of w, 16-byte stores).
// mode 1 (data gradient): k = co, n = ci, filter rotated by 180 degrees = the forward transform with points 0 and 3 swapped in both directions;
//         thread = 8 consecutive co x one ci (32 contiguous bytes of w per tap).
// scale / shift (forward only, nullable): BatchNorm-apply on load -- g is scaled by the PRODUCER's BatchNorm scale per input channel
// (floored as in winograd.hip's fold).
__device__ __forceinline__ void x6_pieces8(const float (&t)[8], x6_i32x4& h, x6_i32x4& m, x6_i32x4& l) {
    float a[8], b[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = t[e] - x6_trunc(t[e]); b[e] = a[e] - x6_trunc(a[e]); }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h[e] = (int)x6_hi2(t[2 * e], t[2 * e + 1]); m[e] = (int)x6_hi2(a[2 * e], a[2 * e + 1]); l[e] = (int)x6_hi2(b[2 * e], b[2 * e + 1]);
    }
}
__device__ __forceinline__ void x6_transform_g(const float (&g)[3][3], float (&t)[16]) {
    float s[4][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        s[0][b] = g[0][b];
        s[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
        s[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
        s[3][b] = g[2][b];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        t[4 * r + 0] = s[r][0]; t[4 * r + 1] = 0.5f * (s[r][0] + s[r][1] + s[r][2]);
        t[4 * r + 2] = 0.5f * (s[r][0] - s[r][1] + s[r][2]); t[4 * r + 3] = s[r][2];
    }
}
constexpr float kX6FoldScaleFloor = 1e-30f;
__device__ __forceinline__ void x6_weight_item(const float* __restrict__ w, uint16_t* __restrict__ U6, int Ci, int Co, int mode, long it,
                                               const float* __restrict__ scale, const float* __restrict__ shift) {
    const int N = mode ? Ci : Co;
    const int k8 = (int)(it / N), n = (int)(it % N);            // 8 consecutive k, one n
    float t[16][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = 8 * k8 + e;
        const int ci = mode ? n : k, co = mode ? k : n;
        float sc = 1.f;
        if (scale) { const float sh = shift[ci]; const float smin = kX6FoldScaleFloor * fmaxf(1.f, fabsf(sh)); sc = scale[ci]; sc = fabsf(sc) < smin ? copysignf(smin, sc) : sc; }
        float g[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) { const float v = w[((size_t)(a * 3 + b) * Ci + ci) * Co + co]; g[a][b] = scale ? sc * v : v; }
        float tt[16];
        x6_transform_g(g, tt);
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) t[xi][e] = tt[xi];
    }
    const int c16 = k8 >> 1, half = k8 & 1;
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) {
        int r = xi >> 2, j = xi & 3;
        if (mode) { r = r == 0 ? 3 : (r == 3 ? 0 : r); j = j == 0 ? 3 : (j == 3 ? 0 : j); }
        x6_i32x4 h, m, l;
        x6_pieces8(t[xi], h, m, l);
        uint16_t* o = U6 + (((size_t)((c16 * 4 + r) * 4 + j) * 3) * N + n) * 16 + 8 * half;
        *reinterpret_cast<x6_i32x4*>(o) = h;
        *reinterpret_cast<x6_i32x4*>(o + (size_t)N * 16) = m;
        *reinterpret_cast<x6_i32x4*>(o + (size_t)2 * N * 16) = l;
    }
}
__global__ __launch_bounds__(256) void wino_x6_weight_kernel(const float* __restrict__ w, uint16_t* __restrict__ U6, int Ci, int Co, int mode) {
    const long items = (long)Ci * Co / 8;
    const long it = (long)blockIdx.x * 256 + threadIdx.x;
    if (it < items) x6_weight_item(w, U6, Ci, Co, mode, it, nullptr, nullptr);
}
// jobs[j] = { w, U6, Ci | Co << 32, first block, mode, 0 }: every direction of every layer in ONE launch
__global__ __launch_bounds__(256) void wino_x6_weight_batch_kernel(const long long* __restrict__ jobs, int njobs) {
    int j = 0;
    while (j + 1 < njobs && (int)jobs[(j + 1) * 6 + 3] <= (int)blockIdx.x) ++j;
    const float* w = reinterpret_cast<const float*>(jobs[j * 6 + 0]);
    uint16_t* U6 = reinterpret_cast<uint16_t*>(jobs[j * 6 + 1]);
    const int Ci = (int)(jobs[j * 6 + 2] & 0xffffffffll), Co = (int)(jobs[j * 6 + 2] >> 32);
    const long it = ((long)blockIdx.x - (int)jobs[j * 6 + 3]) * 256 + threadIdx.x;
    if (it < (long)Ci * Co / 8) x6_weight_item(w, U6, Ci, Co, (int)jobs[j * 6 + 4], it, nullptr, nullptr);
}

// BatchNorm-apply on load (see winograd.hip, wino_weight_fold_kernel): U6 = pieces of scale . transform(w), bias_out = bias + sum_taps shift . w,
// pad = -shift / scale.  A workgroup owns kX6FoldCo output channels and all input-channel groups, so the folded bias is finished inside it.
constexpr int kX6FoldCo = 4, kX6FoldLanes = 256 / kX6FoldCo;
__global__ __launch_bounds__(256) void wino_x6_weight_fold_kernel(const float* __restrict__ w, const float* __restrict__ scale,
        const float* __restrict__ shift, const float* __restrict__ bias, uint16_t* __restrict__ U6, float* __restrict__ bias_out,
        float* __restrict__ pad, int Ci, int Co) {
    __shared__ double sPart[kX6FoldLanes][kX6FoldCo];
    const int col = threadIdx.x % kX6FoldCo, gl = threadIdx.x / kX6FoldCo;
    const int co = blockIdx.x * kX6FoldCo + col;
    const bool live = co < Co;
    double bsum = 0.0;
    for (int c8 = gl; c8 < (Ci >> 3) && live; c8 += kX6FoldLanes) {
        float tsum = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ci = 8 * c8 + e;
            const float sh = shift[ci];
            float gs = 0.f;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) gs += w[((size_t)tap * Ci + ci) * Co + co];
            tsum = fmaf(sh, gs, tsum);
            if (co == 0) {
                const float smin = kX6FoldScaleFloor * fmaxf(1.f, fabsf(sh));
                float sc = scale[ci]; sc = fabsf(sc) < smin ? copysignf(smin, sc) : sc;
                pad[ci] = -sh / sc;
            }
        }
        x6_weight_item(w, U6, Ci, Co, 0, (long)c8 * Co + co, scale, shift);
        bsum += (double)tsum;
    }
    sPart[gl][col] = bsum;
    __syncthreads();
    if (gl == 0 && live) {
        double sacc = 0.0;
#pragma unroll
        for (int l = 0; l < kX6FoldLanes; ++l) sacc += sPart[l][col];
        bias_out[co] = (float)((double)(bias ? bias[co] : 0.f) + sacc);
    }
    if (blockIdx.x == 0 && threadIdx.x < 8) pad[Ci + threadIdx.x] = 0.f;
}

bool x6_shape_ok(int N, int H, int W, int K, int Nout) {
    return N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && K % 32 == 0 && K >= 64 && K <= kWinoFusedMaxK && Nout % 64 == 0;
}

int run_wino_x6(const float* x, int ldx, const uint16_t* U6, const float* bias, float* out, int ldo, int N, int H, int W,
                int K, int Nout, int relu, float* stat_part, hipStream_t st, const WinoBnBwd* bb, const float* pad) {
    X6Args q{};
    WinoFusedArgs& a = q.f;
    q.U6 = U6;
    a.pad = pad;
    a.x = x; a.Uc = nullptr; a.bias = bias; a.out = out; a.ldx = ldx; a.ldo = ldo; a.N = N; a.H = H; a.W = W; a.K = K; a.Nout = Nout; a.relu = relu;
    a.tby = (H / 2 + 7) / 8; a.tbx = (W / 2 + 7) / 8; a.nt = Nout / 64; a.stat_part = stat_part;
    const long blocks = (long)N * a.tby * a.tbx * a.nt;
    if (blocks <= 0 || blocks > 0x7fffffffL) return UNET_EINVAL;
    const int cus = wino_stream_cus();
    const dim3 grid((unsigned)(blocks < cus ? blocks : cus));
    if (bb) {
        a.bn_r = bb->r; a.bn_ldr = bb->ldr; a.bn_c0 = bb->c0; a.bn_c1 = bb->c1;
        wino_x6_stream_bnbwd_kernel<<<grid, 512, 0, st>>>(q, (int)blocks);
    }
    else if (stat_part) wino_x6_stream_stats_kernel<<<grid, 512, 0, st>>>(q, (int)blocks);
    else                wino_x6_stream_kernel<<<grid, 512, 0, st>>>(q, (int)blocks);
    return UNET_LAUNCH_STATUS();
}

}  // namespace

#if (UNET_X6_ABLATE & 8)
extern "C" int unet_debug_x6_timeline(long long* out16) { return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_x6_timeline), 128); }
#endif

// 1 when the BF16x6 kernels take the layer: H, W even, reduce channels K a multiple of 32 (>= 64), output channels a multiple of 64.
extern "C" int unet_winograd_x6_supported(int N, int H, int W, int K, int Nout) { return x6_shape_ok(N, H, W, K, Nout) ? 1 : 0; }

// bytes of one direction's weight operand
extern "C" size_t unet_winograd_x6_weight_bytes(int Cin, int Cout) { return (size_t)16 * 3 * Cin * Cout * sizeof(uint16_t); }

// mode 0: forward operand (k = Cin, n = Cout), mode 1: data-gradient operand (k = Cout, n = Cin, rotated filter)
extern "C" int unet_winograd_weight_transform_x6(const float* w, void* U6, int Cin, int Cout, int mode, void* stream) {
    UNET_CHECK_ARG(w && U6 && Cin > 0 && Cout > 0 && Cin % 16 == 0 && Cout % 16 == 0 && (mode == 0 || mode == 1) && unet_aligned16(U6));
    const long items = (long)Cin * Cout / 8;
    wino_x6_weight_kernel<<<(unsigned)((items + 255) / 256), 256, 0, (hipStream_t)stream>>>(w, (uint16_t*)U6, Cin, Cout, mode);
    return UNET_LAUNCH_STATUS();
}

// jobs: device array of njobs x 6 int64 = { w, U6, Cin | Cout << 32, first_block, mode, 0 }, first_block = running sum of ceil(Cin*Cout/8 / 256)
extern "C" int unet_winograd_weight_transform_x6_batch(const void* jobs, int njobs, int total_blocks, void* stream) {
    UNET_CHECK_ARG(jobs && njobs > 0 && total_blocks > 0);
    wino_x6_weight_batch_kernel<<<dim3((unsigned)total_blocks), 256, 0, (hipStream_t)stream>>>((const long long*)jobs, njobs);
    return UNET_LAUNCH_STATUS();
}

// BatchNorm-apply on load for the BF16x6 forward kernel: the counterpart of unet_winograd_weight_fold
extern "C" int unet_winograd_weight_fold_x6(const float* w, const float* bias, const float* scale, const float* shift, void* U6, float* bias_out,
                                            float* pad, int Cin, int Cout, void* stream) {
    UNET_CHECK_ARG(w && scale && shift && U6 && bias_out && pad && Cin > 0 && Cin % 16 == 0 && Cout > 0 && Cout % 16 == 0 && unet_aligned16(U6));
    wino_x6_weight_fold_kernel<<<(unsigned)((Cout + kX6FoldCo - 1) / kX6FoldCo), 256, 0, (hipStream_t)stream>>>(w, scale, shift, bias, (uint16_t*)U6, bias_out, pad, Cin, Cout);
    return UNET_LAUNCH_STATUS();
}

// Forward: the arguments of unet_conv3x3_fwd_winograd_fused with U6 (unet_winograd_weight_transform_x6 mode 0 / _fold_x6) in place of Uc.
// stat_part rows = unet_conv3x3_fwd_winograd_fused_stats_rows (the persistent grid is the same).
extern "C" int unet_conv3x3_fwd_winograd_x6(const float* x, int ldx, const float* pad, const void* U6, const float* bias, float* out, int ldo,
        int N, int H, int W, int Cin, int Cout, int relu, float* stat_part, size_t stat_bytes, void* stream) {
    UNET_CHECK_ARG(x && U6 && out && x6_shape_ok(N, H, W, Cin, Cout));
    UNET_CHECK_ARG(ldx >= Cin && ldo >= Cout && ldx % 4 == 0 && ldo % 4 == 0 && unet_aligned16(x) && unet_aligned16(U6) && unet_aligned16(out));
    UNET_CHECK_ARG((!bias || unet_aligned16(bias)) && (!pad || unet_aligned16(pad)));
    if (stat_part) {
        const int rows = wino_stats_rows(N, H, W, Cin, Cout);
        UNET_CHECK_ARG(rows > 0);
        if (stat_bytes < (size_t)(Cout / 64) * rows * 128 * sizeof(float)) return UNET_ENOSPC;
    }
    return run_wino_x6(x, ldx, (const uint16_t*)U6, bias, out, ldo, N, H, W, Cin, Cout, relu, stat_part, (hipStream_t)stream, nullptr, pad);
}

// Data gradient: the arguments of unet_conv3x3_dgrad_winograd_fused with U6d (mode 1) in place of Ucd.
extern "C" int unet_conv3x3_dgrad_winograd_x6(const float* dz, int lddz, const void* U6d, float* dx, int lddx,
        int N, int H, int W, int Cin, int Cout, const float* r_prev, int ldr, int c0, int c1,
        float* stat_part, size_t stat_bytes, void* stream) {
    UNET_CHECK_ARG(dz && U6d && dx && x6_shape_ok(N, H, W, Cout, Cin) && (r_prev == nullptr) == (stat_part == nullptr));
    UNET_CHECK_ARG(lddz >= Cout && lddx >= Cin && lddz % 4 == 0 && lddx % 4 == 0 && unet_aligned16(dz) && unet_aligned16(U6d) && unet_aligned16(dx));
    if (!r_prev) return run_wino_x6(dz, lddz, (const uint16_t*)U6d, nullptr, dx, lddx, N, H, W, Cout, Cin, 0, nullptr, (hipStream_t)stream, nullptr, nullptr);
    UNET_CHECK_ARG(c0 >= 0 && c1 > c0 && c1 <= Cin && c0 % 64 == 0 && c1 % 64 == 0 && ldr >= c1 - c0 && ldr % 4 == 0 && unet_aligned16(r_prev));
    const int rows = wino_stats_rows(N, H, W, Cout, Cin);
    UNET_CHECK_ARG(rows > 0);
    if (stat_bytes < (size_t)(Cin / 64) * rows * 128 * sizeof(float)) return UNET_ENOSPC;
    const WinoBnBwd bb{r_prev, ldr, c0, c1};
    return run_wino_x6(dz, lddz, (const uint16_t*)U6d, nullptr, dx, lddx, N, H, W, Cout, Cin, 0, stat_part, (hipStream_t)stream, &bb, nullptr);
}
