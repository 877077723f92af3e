// BF16x6 fused Winograd F(2x2,3x3) forward / data gradient, the round-4 tiling -- kept for layers with fewer than 256 reduce channels.
// Reference layer: UNet._conv_layer, UNet/model.py:28-35.  Arithmetic: winograd_x6.hip (an fp32 value as three bf16 pieces, six exact
// products per fp32-grade product on v_mfma_f32_32x32x16_bf16, fp32 accumulation, fp32 transforms).
//
// Why two tilings.  winograd_x6.hip (round 5) gives each wave one ROW of Winograd points for all 64 channels x 64 tiles and feeds both MFMA
// operands from registers; its end-of-tile exchange and transposing store cost ~8,500 cycles per output tile against ~4,600 per 16-channel
// chunk, so it wins where a tile has many chunks: 1.1-1.25x over this kernel from 256 reduce channels up, a tie at 128, 0.8-0.9x at 64
// (profiles/r05_x6_layer_comparison.txt).  This kernel keeps all 16 points of a [32 channels x 32 tiles] block in one wave -- the output
// transform is lane-local, no exchange -- and passes the split data operand V and the weights U through LDS:
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
// row-stages them in place.  ~5 vector instructions are slotted behind every MFMA (sched_barrier-pinned).
// Weight operand layout (written by winograd_x6.hip's transform kernels for layers that come here):
//   U6[((((k/16) * 4 + r) * 4 + j) * 3 + piece) * N + n) * 16 + k % 16],   point xi = 4 r + j
#include "common.h"
#include "wino_epilogue.h"
#include <cstdlib>

#undef UNET_X6_ABLATE
#ifndef UNET_X6_ABLATE
#define UNET_X6_ABLATE 0        /* this file honours none of the diagnostic bits of winograd_x6.hip */
#endif

namespace {

#if (UNET_X6_ABLATE & 8)
__device__ long long g_x6s_timeline[8];
#define X6_STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory")
#else
#define X6_STAMP(t)
#endif

typedef int x6_i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned x6_u32x2 __attribute__((ext_vector_type(2)));

constexpr int kX6DB = 24 * 1024;                  // one D chunk buffer: 324 pixel slots x 64 B in 24 1-KB DMA pieces (21 used + 3 dummies)
constexpr int kX6IB = 24 * 1024;                  // one V or U unit image
constexpr int kX6Blk = 2048;                      // one (point, piece) block: 64 rows x 32 B
constexpr int kX6V = 2 * kX6DB, kX6U = kX6V + 2 * kX6IB, kX6Smem = kX6U + 2 * kX6IB;      // 147456 B
constexpr int kX6RowB = 18 * 64;                  // bytes between patch rows in D

#define X6_RD128(dst, base, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst) : "v"(base), "n"(off))
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

// Column stage + split of point J (column J of the unit's row of points) in five steps of 5-6 vector instructions; tt[c] = the row-stage
// result of patch column c (4 channels).  V[.][0] = t0 - t2, [1] = t1 + t2, [2] = t2 - t1, [3] = t1 - t3.
// X6_PIN: an empty volatile asm over a step's inputs / results.  Instruction selection orders pure arithmetic freely between the volatile
// MFMAs (sched_barrier only binds the machine scheduler); tied to a volatile statement on both sides a step stays in its gap.
#define X6_PIN(...) asm volatile("" : __VA_ARGS__)
template <int K, int J> __device__ __forceinline__ void x6_split_step(X6Split& s, f32x4 (&tt)[8]) {
#if (UNET_X6_ABLATE & 16)        /* diagnostics: no column stage / split (results wrong) */
    return;
#endif
    if constexpr (K == 0) {
        constexpr int TA = J == 0 ? 0 : J == 2 ? 2 : 1, TB = J == 0 ? 2 : J == 1 ? 2 : J == 2 ? 1 : 3;
        X6_PIN("+v"(tt[TA]), "+v"(tt[TB]));
        const f32x4 vv = J == 1 ? tt[TA] + tt[TB] : tt[TA] - tt[TB];
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
    }
}
// the three pieces of point J -> V image PAR (3 ds_write_b64)
template <int PAR, int J> __device__ __forceinline__ void x6_write_v(const X6Split& s, unsigned v_base) {
    X6_WR64(v_base, PAR * kX6IB + (J * 3 + 0) * kX6Blk, (x6_u32x2{s.h[0], s.h[1]}));
    X6_WR64(v_base, PAR * kX6IB + (J * 3 + 1) * kX6Blk, (x6_u32x2{s.m[0], s.m[1]}));
    X6_WR64(v_base, PAR * kX6IB + (J * 3 + 2) * kX6Blk, (x6_u32x2{s.l[0], s.l[1]}));
}

// raw rows of the unit with point row R2 from D buffer DPR -> dd[0..3] (row ra), dd[4..7] (row rb); tt = ra -/+ rb:
//   R2 = 0: d0 - d2;  1: d1 + d2;  2: d2 - d1;  3: d1 - d3
template <int R2> struct X6Rows {
    static constexpr int RA = R2 == 0 ? 0 : R2 == 2 ? 2 : 1, RB = R2 == 0 ? 2 : R2 == 1 ? 2 : R2 == 2 ? 1 : 3;
    static constexpr bool ADD = R2 == 1;
};
template <int R2, int DPR, int C> __device__ __forceinline__ void x6_read_rows(f32x4 (&dd)[8], const unsigned (&d_base)[4]) {
    X6_RD128(dd[C], d_base[C], DPR * kX6DB + X6Rows<R2>::RA * kX6RowB);
    X6_RD128(dd[4 + C], d_base[C], DPR * kX6DB + X6Rows<R2>::RB * kX6RowB);
}
template <int R2, int C> __device__ __forceinline__ void x6_row_stage(f32x4 (&dd)[8]) {
    X6_PIN("+v"(dd[C]), "+v"(dd[4 + C]));
    dd[C] = X6Rows<R2>::ADD ? dd[C] + dd[4 + C] : dd[C] - dd[4 + C];
    X6_PIN("+v"(dd[C]));
}
#define X6_TIE_DD(d) "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7])

// One unit: the 24 MFMAs of point row R of a chunk with D parity DP (V / U buffers R & 1), and everything that runs in their shadow.
//   S0 / S1: the two sets of 8 row registers; set (R & 1) holds the row stage of unit g+1 (consumed here), the other one takes the raw
//            rows of unit g+2 and ends as its row stage;
//   us: source of U(g+1) for this wave's first block (uniform), uoff[j]: the lane's byte offset of its piece j from there;
//   dptr: this lane's running sources of its six D pieces -- the two issued in this unit (R != 1) advance by one chunk, or jump to the next
//         tile's patch (dnxt) when `dswitch` says this was the tile's last chunk (uniform).
// LDS instructions retire in order; the lgkmcnt immediates count the LDS instructions issued behind the one waited for.
template <int R, int DP, bool FIRST>
__device__ __forceinline__ void x6_unit(f32x16 (&acc)[16], f32x4 (&S0)[8], f32x4 (&S1)[8], X6Frag (&fr)[2], X6Split& sp,
                                        unsigned a_base, unsigned b_base, const unsigned (&d_base)[4], unsigned v_base,
                                        const char* us, const unsigned (&uoff)[6], const float* (&dptr)[6], const float* (&dnxt)[6], bool dswitch,
                                        unsigned lds_w, long long (&tl)[6]) {
    constexpr int P = R & 1, PN = P ^ 1;
#if (UNET_X6_ABLATE & 8)
    long long q0, q1, q2, q3;
    X6_STAMP(q0);
#endif
    constexpr int R2 = (R + 2) & 3, DPR = DP ^ (R >= 2 ? 1 : 0);
    // D pieces issued here: R = 2: chunk c+2 pieces 0,1 (wave's j = 0,1); R = 3: pieces j = 2,3; R = 0: chunk c+1, j = 4,5; R = 1: none
    constexpr int ND = R == 1 ? 0 : 2;
    constexpr int DJ = R == 2 ? 0 : R == 3 ? 2 : 4;
    constexpr int DPW = R == 0 ? (DP ^ 1) : DP;                       // buffer of that chunk
    f32x4 (&tt)[8] = P ? S1 : S0;
    f32x4 (&dd)[8] = P ? S0 : S1;
    asm volatile("s_waitcnt lgkmcnt(0)" : X6_TIE_FRAG(fr[0]));        // point 0's operands (issued by the caller side of the barrier)
    X6_STAMP(q1);
#pragma unroll
    for (int n = 0; n < 24; ++n) {
        const int p = n / 6, k = n % 6;
        X6Frag& f = fr[p & 1];
        if (k == 0 && p > 0) asm volatile("s_waitcnt lgkmcnt(3)" : X6_TIE_FRAG(f));       // behind its reads: the previous point's 3 V writes
        // piece products, small to large: (m,m) (l,h) (h,l) (m,h) (h,m) (h,h);  u = weights (rows = channels), v = data (columns = tiles)
        const int ui = k == 0 ? 1 : k == 1 ? 2 : k == 2 ? 0 : k == 3 ? 1 : 0;
        const int vi = k == 0 ? 1 : k == 1 ? 0 : k == 2 ? 2 : k == 3 ? 0 : k == 4 ? 1 : 0;
#if !(UNET_X6_ABLATE & 32)       /* diagnostics: 32 = no MFMAs (results wrong) */
        if (FIRST && k == 0) X6_MFMA0(acc[4 * R + p], f.u[ui], f.v[vi]);
        else X6_MFMA(acc[4 * R + p], f.u[ui], f.v[vi]);
#endif
        // ---- in the shadow of MFMA n
        if (k == 0 && p < 3 && !(UNET_X6_ABLATE & 128)) {        // operands of the next point
            if (p == 0) x6_read_ops<P, 1>(fr[1], a_base, b_base);
            if (p == 1) x6_read_ops<P, 2>(fr[0], a_base, b_base);
            if (p == 2) x6_read_ops<P, 3>(fr[1], a_base, b_base);
        }
        if (n >= 1 && n <= 4 && !(UNET_X6_ABLATE & 512)) {       // raw rows of unit g+2, two reads per gap
            if (n == 1) x6_read_rows<R2, DPR, 0>(dd, d_base);
            if (n == 2) x6_read_rows<R2, DPR, 1>(dd, d_base);
            if (n == 3) x6_read_rows<R2, DPR, 2>(dd, d_base);
            if (n == 4) x6_read_rows<R2, DPR, 3>(dd, d_base);
        }
        if ((n & 1) == 0 && n < 12 && !(UNET_X6_ABLATE & 64))    // U(g+1): one DMA every second MFMA (the vector-memory issue path is busy ~64 cycles per DMA)
            X6_DMA_S(uoff[n >> 1], us, lds_w, kX6U + PN * kX6IB + (n >> 1) * 4096);
        if (ND && n == 12 && !(UNET_X6_ABLATE & 64)) X6_DMA_V(dptr[DJ], lds_w, DPW * kX6DB + DJ * 4096);
        if (ND && n == 14 && !(UNET_X6_ABLATE & 64)) X6_DMA_V(dptr[DJ + 1], lds_w, DPW * kX6DB + (DJ + 1) * 4096);
        if (ND && n == 17) {                                     // (a light gap) the two pointers move on
            if (dswitch) { dptr[DJ] = dnxt[DJ]; dptr[DJ + 1] = dnxt[DJ + 1]; }
            else { dptr[DJ] += 16; dptr[DJ + 1] += 16; }
            X6_PIN("+v"(dptr[DJ]), "+v"(dptr[DJ + 1]));
        }
        if (k < 5) {                                             // column stage + split of point p of unit g+1
            if (p == 0) { if (k == 0) x6_split_step<0, 0>(sp, tt); if (k == 1) x6_split_step<1, 0>(sp, tt); if (k == 2) x6_split_step<2, 0>(sp, tt); if (k == 3) x6_split_step<3, 0>(sp, tt); if (k == 4) x6_split_step<4, 0>(sp, tt); }
            if (p == 1) { if (k == 0) x6_split_step<0, 1>(sp, tt); if (k == 1) x6_split_step<1, 1>(sp, tt); if (k == 2) x6_split_step<2, 1>(sp, tt); if (k == 3) x6_split_step<3, 1>(sp, tt); if (k == 4) x6_split_step<4, 1>(sp, tt); }
            if (p == 2) { if (k == 0) x6_split_step<0, 2>(sp, tt); if (k == 1) x6_split_step<1, 2>(sp, tt); if (k == 2) x6_split_step<2, 2>(sp, tt); if (k == 3) x6_split_step<3, 2>(sp, tt); if (k == 4) x6_split_step<4, 2>(sp, tt); }
            if (p == 3) { if (k == 0) x6_split_step<0, 3>(sp, tt); if (k == 1) x6_split_step<1, 3>(sp, tt); if (k == 2) x6_split_step<2, 3>(sp, tt); if (k == 3) x6_split_step<3, 3>(sp, tt); if (k == 4) x6_split_step<4, 3>(sp, tt); }
            if (k == 4 && !(UNET_X6_ABLATE & 256)) {
                if (p == 0) x6_write_v<PN, 0>(sp, v_base);
                if (p == 1) x6_write_v<PN, 1>(sp, v_base);
                if (p == 2) x6_write_v<PN, 2>(sp, v_base);
                if (p == 3) x6_write_v<PN, 3>(sp, v_base);
            }
        } else {                                                 // row stage of unit g+2, one patch column per point
            if (p == 0) { asm volatile("s_waitcnt lgkmcnt(3)" : X6_TIE_DD(dd)); x6_row_stage<R2, 0>(dd); }      // behind the row reads: point 0's V writes
            if (p == 1) x6_row_stage<R2, 1>(dd);
            if (p == 2) x6_row_stage<R2, 2>(dd);
            if (p == 3) x6_row_stage<R2, 3>(dd);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    X6_STAMP(q2);
#if (UNET_X6_ABLATE & 8)
    if (ND) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    X6_STAMP(q3);
    asm volatile("s_barrier" ::: "memory");
    { long long q4; X6_STAMP(q4); tl[0] += q1 - q0; tl[1] += q2 - q1; tl[2] += q3 - q2; tl[3] += q4 - q3; tl[4] += 1; }
#else
    if (ND) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
    x6_read_ops<PN, 0>(fr[0], a_base, b_base);                   // point 0 of the next unit
}

struct X6Args {
    WinoFusedArgs f;             // x, bias, out, geometry, stats, pad; f.Uc unused
    const uint16_t* U6;          // [K/16][unit 4][point 4][piece 3][Nout][16] bf16
};

template <int STATS>
__device__ __forceinline__ void x6_stream_body(const X6Args& q, int ntiles) {
    const WinoFusedArgs& p = q.f;
    __shared__ __attribute__((aligned(1024))) char smem[kX6Smem];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mi = wv & 1, ni = wv >> 1;
    const int li = lane & 31, lh = lane >> 5;
    const int nchunks = p.K / 16;

    // ---- DMA duty.  U: piece wv + 4 j of a unit image = block (wv >> 1) + 2 j, rows 32 (wv & 1) + lane / 2, 16-byte slot lane & 1
    //      (source-side swizzle: slot ^ bit 3 of the row).  D: piece wv + 4 j = pixel slots 16 (wv + 4 j) + lane / 4, channel quad lane & 3.
    const int urow = 32 * (wv & 1) + (lane >> 1);
    const unsigned u_lane = (unsigned)(urow * 32 + 16 * ((lane & 1) ^ ((urow >> 3) & 1)));
    unsigned uoff[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) uoff[j] = u_lane + (unsigned)j * 2u * (unsigned)p.Nout * 32u;      // blocks b and b + 2 are 2 N rows apart
    const size_t ustep = (size_t)12 * p.Nout * 32;                                    // bytes between units
    int ppy[6], ppx[6], poff[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int s = 16 * (wv + 4 * j) + (lane >> 2);
        const int py = s / 18, px = x6_col_of(s % 18);
        ppy[j] = s < 324 ? py : (1 << 20);                                            // past the patch: never inside the image
        ppx[j] = px;
        poff[j] = (py * p.W + px) * p.ldx + 4 * (lane & 3);
    }
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
    auto tile_sources = [&](const TileCoord& c, const float* (&dp)[6], const char*& ub0) {
        const int gy0 = 16 * c.by - 1, gx0 = 16 * c.bx - 1;
        const float* xb = p.x + ((long long)(c.img * p.H + gy0) * p.W + gx0) * p.ldx;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const bool ok = (unsigned)(gy0 + ppy[j]) < (unsigned)p.H && (unsigned)(gx0 + ppx[j]) < (unsigned)p.W;
            dp[j] = ok ? xb + poff[j] : padsrc + 4 * (lane & 3);
        }
        ub0 = reinterpret_cast<const char*>(q.U6) + ((size_t)(wv >> 1) * p.Nout + (size_t)c.tn * 64) * 32;
    };

    // ---- LDS byte addresses
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_f*)smem;
    const int arow = 32 * mi + li, brow = 32 * ni + li;
    const unsigned a_base = lds0 + kX6V + (unsigned)(arow * 32 + 16 * (lh ^ ((arow >> 3) & 1)));
    const unsigned b_base = lds0 + kX6U + (unsigned)(brow * 32 + 16 * (lh ^ ((brow >> 3) & 1)));
    const int t_lt = 16 * wv + (lane >> 2), t_q = lane & 3;                      // transform duty: (tile, channel quad)
    unsigned d_base[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
        d_base[c] = lds0 + (unsigned)(((2 * (t_lt >> 3)) * 18 + x6_slot_of(2 * (t_lt & 7) + c)) * 64 + 16 * t_q);
    const unsigned v_base = lds0 + kX6V + (unsigned)(t_lt * 32 + 16 * ((t_q >> 1) ^ ((t_lt >> 3) & 1)) + 8 * (t_q & 1));
    const unsigned lds_w = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + wv * 1024));       // this wave's first piece, as an M0 value

    f32x16 acc[16];
    f32x4 S0[8], S1[8];
    X6Frag fr[2];
    X6Split sp;
    const float* dptr[6]; const float* dnxt[6]; const char* ucur; const char* unxt;
    int t = blockIdx.x;
    if ((gridDim.x & 7) == 0 && (p.nt & 7) != 0) t = (t & 7) * (int)(gridDim.x >> 3) + (t >> 3);       // XCD-aware renumbering, as winograd.hip
    const int t_first = t;
    f32x2 s1[8], s2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { s1[i] = f32x2{0.f, 0.f}; s2[i] = f32x2{0.f, 0.f}; }
    TileCoord tc = decode(t);
    tile_sources(tc, dptr, ucur);

    // ---- prologue of the workgroup's first tile: D(0), D(1) pieces 0..3, U(unit 0) -> LDS; V(unit 0) by a full transform; row stage of unit 1
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        X6_DMA_V(dptr[j], lds_w, j * 4096);
        if (j < 4) X6_DMA_V(dptr[j] + 16, lds_w, kX6DB + j * 4096);
        X6_DMA_S(uoff[j], ucur, lds_w, kX6U + j * 4096);
        dptr[j] += j < 4 ? 32 : 16;                              // next issue: chunk 2 (pieces 0..3), chunk 1 (pieces 4, 5)
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    x6_read_rows<0, 0, 0>(S1, d_base); x6_read_rows<0, 0, 1>(S1, d_base); x6_read_rows<0, 0, 2>(S1, d_base); x6_read_rows<0, 0, 3>(S1, d_base);
    x6_read_rows<1, 0, 0>(S0, d_base); x6_read_rows<1, 0, 1>(S0, d_base); x6_read_rows<1, 0, 2>(S0, d_base); x6_read_rows<1, 0, 3>(S0, d_base);
    asm volatile("s_waitcnt lgkmcnt(0)" : X6_TIE_DD(S1));
    asm volatile("" : X6_TIE_DD(S0));
    x6_row_stage<0, 0>(S1); x6_row_stage<0, 1>(S1); x6_row_stage<0, 2>(S1); x6_row_stage<0, 3>(S1);
    x6_row_stage<1, 0>(S0); x6_row_stage<1, 1>(S0); x6_row_stage<1, 2>(S0); x6_row_stage<1, 3>(S0);
    x6_split_step<0, 0>(sp, S1); x6_split_step<1, 0>(sp, S1); x6_split_step<2, 0>(sp, S1); x6_split_step<3, 0>(sp, S1); x6_split_step<4, 0>(sp, S1); x6_write_v<0, 0>(sp, v_base);
    x6_split_step<0, 1>(sp, S1); x6_split_step<1, 1>(sp, S1); x6_split_step<2, 1>(sp, S1); x6_split_step<3, 1>(sp, S1); x6_split_step<4, 1>(sp, S1); x6_write_v<0, 1>(sp, v_base);
    x6_split_step<0, 2>(sp, S1); x6_split_step<1, 2>(sp, S1); x6_split_step<2, 2>(sp, S1); x6_split_step<3, 2>(sp, S1); x6_split_step<4, 2>(sp, S1); x6_write_v<0, 2>(sp, v_base);
    x6_split_step<0, 3>(sp, S1); x6_split_step<1, 3>(sp, S1); x6_split_step<2, 3>(sp, S1); x6_split_step<3, 3>(sp, S1); x6_split_step<4, 3>(sp, S1); x6_write_v<0, 3>(sp, v_base);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    x6_read_ops<0, 0>(fr[0], a_base, b_base);

    long long tl[6] = {0, 0, 0, 0, 0, 0};
    for (; t < ntiles; t += gridDim.x) {
#if (UNET_X6_ABLATE & 8)
        long long e0; X6_STAMP(e0);
#endif
        const TileCoord tcn = t + (int)gridDim.x < ntiles ? advance(tc) : tc;            // the last tile prefetches itself again
        tile_sources(tcn, dnxt, unxt);
        f32x4 bias4[4];
        wf_load_bias(p, tc.tn * 64, ni, lh, bias4);
        // U(g + 1) of unit g = 4 c + R, continuing into the next tile.  D pieces: unit R = 0 issues chunk c + 1 (the tile's last one when
        // c = nchunks - 2), R = 2, 3 issue chunk c + 2 (the last one when c = nchunks - 3); behind the last chunk the pointers jump to the next tile
#define X6_US(c, R) ((4 * (c) + (R) + 1 < 4 * nchunks) ? ucur + (size_t)(4 * (c) + (R) + 1) * ustep : unxt)
#define X6_UNIT(R, DP, FIRST, c) \
        x6_unit<R, DP, FIRST>(acc, S0, S1, fr, sp, a_base, b_base, d_base, v_base, X6_US(c, R), uoff, dptr, dnxt, \
                              (R) == 0 ? (c) == nchunks - 2 : (c) == nchunks - 3, lds_w, tl)
        X6_UNIT(0, 0, true, 0); X6_UNIT(1, 0, true, 0); X6_UNIT(2, 0, true, 0); X6_UNIT(3, 0, true, 0);
        X6_UNIT(0, 1, false, 1); X6_UNIT(1, 1, false, 1); X6_UNIT(2, 1, false, 1); X6_UNIT(3, 1, false, 1);
        for (int c = 2; c < nchunks; c += 2) {
            X6_UNIT(0, 0, false, c); X6_UNIT(1, 0, false, c); X6_UNIT(2, 0, false, c); X6_UNIT(3, 0, false, c);
            X6_UNIT(0, 1, false, c + 1); X6_UNIT(1, 1, false, c + 1); X6_UNIT(2, 1, false, c + 1); X6_UNIT(3, 1, false, c + 1);
        }
#undef X6_UNIT
#undef X6_US
        asm volatile("s_waitcnt lgkmcnt(0)" : X6_TIE_FRAG(fr[0]));           // the next unit's first operands have landed before anything below may move them
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // inline-asm MFMAs are invisible to the compiler's hazard recogniser
        f32x4 rv[4][4];
        if (STATS == 2) wf_load_r(p, tc.img, tc.by, tc.bx, tc.tn * 64, mi, ni, li, lh, rv);
        wf_epilogue<STATS>(acc, p, tc.img, tc.by, tc.bx, tc.tn * 64, mi, ni, li, lh, bias4, s1, s2, rv);
        ucur = unxt; tc = tcn;
#if (UNET_X6_ABLATE & 8)
        { long long e1; X6_STAMP(e1); tl[5] += e1 - e0; }
#endif
    }
#if (UNET_X6_ABLATE & 8)
    if (blockIdx.x == 0 && tid == 0) for (int i = 0; i < 6; ++i) g_x6s_timeline[i] = tl[i];
#endif
    // retire the prefetches of the tile that never runs (LDS reads into fr[0], DMAs) before the wave ends
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : X6_TIE_FRAG(fr[0]) :: "memory");
    if (STATS) wf_write_stats(p, t_first, 2 * ((int)gridDim.x / p.nt), mi, ni, li, lh, s1, s2);
}
__global__ __launch_bounds__(256, 1) void wino_x6s_stream_kernel(X6Args q, int ntiles) { x6_stream_body<0>(q, ntiles); }
__global__ __launch_bounds__(256, 1) void wino_x6s_stream_stats_kernel(X6Args q, int ntiles) { x6_stream_body<1>(q, ntiles); }
__global__ __launch_bounds__(256, 1) void wino_x6s_stream_bnbwd_kernel(X6Args q, int ntiles) { x6_stream_body<2>(q, ntiles); }

}  // namespace

// called by winograd_x6.hip's entry points for K < 256 (arguments as its run_wino_x6; bn_r null = no BatchNorm-backward sums)
int unet_run_wino_x6_small_k(const float* x, int ldx, const void* U6, const float* bias, float* out, int ldo, int N, int H, int W,
                             int K, int Nout, int relu, float* stat_part, hipStream_t st, const float* bn_r, int bn_ldr, int bn_c0, int bn_c1,
                             const float* pad, int max_workgroups) {
    X6Args q{};
    WinoFusedArgs& a = q.f;
    q.U6 = (const uint16_t*)U6;
    a.pad = pad;
    a.x = x; a.Uc = nullptr; a.bias = bias; a.out = out; a.ldx = ldx; a.ldo = ldo; a.N = N; a.H = H; a.W = W; a.K = K; a.Nout = Nout; a.relu = relu;
    a.tby = (H / 2 + 7) / 8; a.tbx = (W / 2 + 7) / 8; a.nt = Nout / 64; a.stat_part = stat_part;
    const long blocks = (long)N * a.tby * a.tbx * a.nt;
    if (blocks <= 0 || blocks > 0x7fffffffL) return UNET_EINVAL;
    const int cus = unet_grid_slots(wino_stream_cus(), max_workgroups);
    const dim3 grid((unsigned)(blocks < cus ? blocks : cus));
    if (bn_r) {
        a.bn_r = bn_r; a.bn_ldr = bn_ldr; a.bn_c0 = bn_c0; a.bn_c1 = bn_c1;
        wino_x6s_stream_bnbwd_kernel<<<grid, 256, 0, st>>>(q, (int)blocks);
    }
    else if (stat_part) wino_x6s_stream_stats_kernel<<<grid, 256, 0, st>>>(q, (int)blocks);
    else                wino_x6s_stream_kernel<<<grid, 256, 0, st>>>(q, (int)blocks);
    return UNET_LAUNCH_STATUS();
}
