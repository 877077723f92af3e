// Fused Winograd F(2x2,3x3) forward / data gradient with the 16 point-products on the BF16 matrix pipe at fp32 grade ("BF16x6").
// Reference layer: UNet._conv_layer, UNet/model.py:28-35 (fp32 in the reference; this is a second ROUTE to the same fp32-grade result,
// beside the v_mfma_f32_32x32x2_f32 kernels of winograd.hip, which stay selectable).
//
// Arithmetic.  An fp32 value is EXACTLY the sum of three bf16 pieces h + m + l (8 + 8 + 8 significant bits; h = rn(v), m = rn(v - h),
// l = v - h - m, round-to-nearest-even since round 6 -- rounds 4-5 truncated).  With both operands of a product split this way,
//     a b  =  ah bh + ah bm + am bh + ah bl + al bh + am bm   +  (am bl + al bm + al bl),
// the six kept piece products are exact in an fp32 accumulator and the three dropped ones are zero-mean and sum to at most 2^-23.4 |a b|
// (rms 2^-26) -- below ONE fp32 multiply's rounding (tests/test_bf16x6_arithmetic.py restates and checks this on the CPU).  Accumulation is fp32 (v_mfma_f32_32x32x16_bf16).  So the result
// carries the rounding of an fp32 dot product (measured against fp64: tests/test_gpu_fp32_errors.py, tests/test_gpu_x6.py), at 6 x 32
// matrix-pipe cycles per 32 x 32 x 16 block instead of 8 x 64.  The Winograd transforms themselves (B^T d B on the data, G g G^T on the
// weights, A^T m A on the result) stay fp32 vector arithmetic.
// Not covered: Inf / NaN inputs give NaN (Inf - Inf in the split); values below ~1e-33 lose their low pieces to bf16 underflow.
//
// Round 5: ONE POINT ROW PER WAVE, both MFMA operands straight from registers.  (Round 4's kernel kept a wave's [32 channels x 32 tiles]
// x 16 points and passed the split data operand V and the weights U through LDS: 1030 LDS cycles per 768 matrix cycles, profiles/
// r04_x6_ablation.txt -- every schedule of that shape landed at the same time.)  Workgroup = 8x8 Winograd tiles x 64 output channels as
// before; wave r (0..3) owns the four Winograd points of point ROW r for ALL 64 channels x 64 tiles: 4 points x 2 channel blocks x 2 tile
// blocks of 32 x 32 = 256 accumulator registers.  Then
//   * the data operand of a lane IS what that lane transforms: MFMA B-operand lane (li, lh) holds tile li, reduce channels 8 lh .. 8 lh + 7,
//     so the lane reads its tile's raw rows ra, rb of the patch (row stage r: d[ra] -/+ d[rb]), forms the column stage of point j and splits
//     the 8 values into pieces in its own registers -- V never touches LDS, no barrier sits between transform and product;
//   * the weight operand is loaded from global memory (L2-resident, pre-split, stored in MFMA A-operand order: one contiguous 1-KB
//     global_load_dwordx4 per fragment) straight into registers, one point ahead -- no LDS, no LDS-DMA issue cost for U;
//   * LDS carries only the raw patch D (18 x 18 px x 16 ch fp32 per 16-channel chunk, by LDS-DMA, two buffers) in four channel-quad PLANES
//     [quad][row][20 slots] with even columns first: the 16 lanes of a ds_read_b128 group then hit 16 different 16-byte bank groups
//     (row pitch 20 slots: tile rows two apart land 8 slots apart mod 16), and the end-of-tile exchange below.
// Per 16-channel chunk a wave issues 96 MFMAs, 480 vector instructions (128 transform adds + 352 for the split: 5 per MFMA gap, the
// figure one wave per SIMD can hide), 32 ds_read_b128, 24 global_load_dwordx4, 6 LDS-DMA pieces and ONE barrier -- against 96 MFMAs, the
// same 480 vector instructions, 128 ds_read_b128, 48 ds_write_b64, 30 LDS-DMA pieces and four barriers in round 4.
// The price: the output transform needs all four point rows of a (tile, channel): the waves reduce their rows' column stage
// (m0 + m1 + m2, m1 - m2 - m3) locally and exchange the halves through LDS once per tile (96 KB), each wave finishing one
// [32 channels x 32 tiles] quarter with the epilogue element order of the other Winograd kernels (bias, ReLU, BatchNorm sums).
#include <type_traits>
#include "common.h"
#include "wino_epilogue.h"
#include <cstdlib>

#ifndef UNET_X6_ABLATE
#define UNET_X6_ABLATE 0        /* diagnostic builds (scripts/build_variant.sh), bits: 8 = s_memtime stamps, 16 = no split, 32 = no MFMAs, 64 = no patch DMA, 128 = no weight loads, 512 = no row reads, 1024 = no output stores (results wrong) */
#endif

namespace {

typedef int x6_i32x4 __attribute__((ext_vector_type(4)));

#if (UNET_X6_ABLATE & 8)        /* diagnostics: s_memtime stamps around the phases of a period (workgroup 0, wave 0) */
__device__ long long g_x6_timeline[16];
#define X6_STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory")
#else
#define X6_STAMP(t)
#endif

constexpr int kX6RowB = 20 * 16;                  // bytes between patch rows inside a plane: 18 pixel slots + 2 pad slots of 16 B
constexpr int kX6PlaneB = 18 * kX6RowB;           // one channel-quad plane of a chunk: 5760 B
constexpr int kX6DB = 24 * 1024;                  // one D chunk buffer: 4 planes = 23040 B in 24 1-KB DMA pieces (the last 1.5 are dummies)
constexpr int kX6X = 2 * kX6DB;                   // end-of-tile exchange: per owner wave 3 sources x 8 KB
constexpr int kX6XW = 24 * 1024;
constexpr int kX6Smem = kX6X + 4 * kX6XW;         // 147456 B
constexpr int kX6UPoint = 6 * 1024;               // bytes of one point's weight fragments: [piece 3][channel block 2][lane 64][16 B]
constexpr int kX6UChunkWave = 4 * kX6UPoint;      // one point row of a chunk
constexpr int kX6UChunk = 4 * kX6UChunkWave;      // all four point rows
constexpr int kX6TileB = 528;                     // bytes per tile in a wave's transposed output area [tile 32][pixel 4][32 channels] (512 + 16: conflict-free both ways)

#define X6_RD128(dst, base, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst) : "v"(base), "n"(off))
#define X6_WR128(base, off, val) asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(base), "v"(val), "n"(off) : "memory")
#define X6_MFMA(accv, av, bv) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(accv) : "v"(av), "v"(bv) : "memory")
#define X6_MFMA0(accv, av, bv) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(accv) : "v"(av), "v"(bv) : "memory")
// LDS-DMA of one 1-KB piece (16 B per lane) to LDS byte address ldsw + ldsoff (wave-uniform): M0 carries the LDS address.
#define X6_DMA_V(vptr, ldsw, ldsoff) asm volatile("s_add_u32 m0, %1, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(vptr), "s"(ldsw), "n"(ldsoff) : "memory", "scc")
// the same in two statements, for a stream that has an instruction of its own to put between them (nothing else in these kernels uses M0)
#define X6_DMA_M0(ldsw, ldsoff) asm volatile("s_add_u32 m0, %0, %1" :: "s"(ldsw), "n"(ldsoff) : "memory", "scc")
#define X6_DMA_GO(vptr) asm volatile("global_load_lds_dwordx4 %0, off" :: "v"(vptr) : "memory")
// one weight fragment: 16 B per lane at sbase + voff + imm (sbase wave-uniform)
#define X6_LDU(dst, voff, sbase, imm) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=&v"(dst) : "v"(voff), "s"(sbase), "n"(imm) : "memory")

// slot of patch column x (0..17) inside a plane row: even columns first (0, 2, .., 16 -> slots 0..8), then the odd ones (slots 9..17)
__host__ __device__ constexpr int x6_pos_of(int x) { return (x & 1) * 9 + (x >> 1); }
__host__ __device__ constexpr int x6_col_of(int pos) { return pos < 9 ? 2 * pos : 2 * (pos - 9) + 1; }

struct X6Split { float v[4], a[4], b[4]; unsigned h[2], m[2], l[2]; };

// Round 6: the pieces are taken by ROUND-TO-NEAREST-EVEN (v_cvt_pk_bf16_f32, two values per instruction) instead of by truncation: h = rn(v),
// m = rn(v - h), l = v - h - m are still an exact split (|v - h| <= half a bf16 ulp of v and a multiple of v's fp32 ulp: 16 significant bits;
// the second remainder has at most 8), at the same instruction count (pack 1 + unpack 2 + subtract 2 per value pair and level, where truncation
// took mask 2 + subtract 2 + pack 1), but the pieces m, l no longer carry the sign of v: the three dropped products am bl + al bm + al bl are
// zero-mean and bounded by 2^-23.4 |a b| instead of a bias of the product's sign up to 2^-21 (tests/test_bf16x6_arithmetic.py).
typedef __bf16 x6_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned x6_rn2(float lo, float hi) {       // { bf16(lo), bf16(hi) } rounded to nearest even, lo in the low half
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, x6_bf16x2));
}
__device__ __forceinline__ float x6_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }                 // the two bf16 values of a pair, as fp32
__device__ __forceinline__ float x6_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// Column stage + split of four channels of point J (column J of the wave's row of points) in five steps of 5-6 vector instructions;
// T[c] = the row-stage result of patch column c.  V[.][0] = t0 - t2, [1] = t1 + t2, [2] = t2 - t1, [3] = t1 - t3.
// X6_PIN: an empty volatile asm over a step's inputs / results.  Instruction selection orders pure arithmetic freely between the volatile
// MFMAs (sched_barrier only binds the machine scheduler); tied to a volatile statement on both sides a step stays in its gap.
#define X6_PIN(...) asm volatile("" : __VA_ARGS__)
#define X6_USE(...) asm volatile("" :: __VA_ARGS__)
template <int K, int J, int G> __device__ __forceinline__ void x6_split_step(X6Split& s, f32x4 (&T)[4][4]) {
#if (UNET_X6_ABLATE & 16)        /* diagnostics: no column stage / split (results wrong) */
    return;
#endif
    if constexpr (K == 0) {
        constexpr int TA = J == 0 ? 0 : J == 2 ? 2 : 1, TB = J == 0 ? 2 : J == 1 ? 2 : J == 2 ? 1 : 3;
        // (no X6_PIN over the two quads and no asm here: a vector instruction that reads a register an asm statement has just defined -- an empty
        //  "+v" pin included -- is padded with an s_nop by the compiler unless an instruction of its own lies in between, and with ~7 instructions
        //  in every MFMA gap each s_nop is ~4 cycles of matrix time (scripts/micro/valu_dep_cost.hip); X6_USE below keeps the step in its gap)
        // (this group's two quads were pinned in the middle of the previous group's last step)
        if constexpr (J == 1) { const f32x4 vv = T[TA][G] + T[TB][G]; s.v[0] = vv[0]; s.v[1] = vv[1]; s.v[2] = vv[2]; s.v[3] = vv[3]; }
        else { const f32x4 vv = T[TA][G] - T[TB][G]; s.v[0] = vv[0]; s.v[1] = vv[1]; s.v[2] = vv[2]; s.v[3] = vv[3]; }
        s.h[0] = x6_rn2(s.v[0], s.v[1]);
        X6_USE("v"(s.v[0]), "v"(s.v[1]), "v"(s.v[2]), "v"(s.v[3]), "v"(s.h[0]));
    } else if constexpr (K == 1) {
        s.a[0] = s.v[0] - x6_lo(s.h[0]); s.a[1] = s.v[1] - x6_hi(s.h[0]);
        s.h[1] = x6_rn2(s.v[2], s.v[3]);
        X6_USE("v"(s.a[0]), "v"(s.a[1]), "v"(s.h[1]));
    } else if constexpr (K == 2) {
        s.a[2] = s.v[2] - x6_lo(s.h[1]); s.a[3] = s.v[3] - x6_hi(s.h[1]);
        s.m[0] = x6_rn2(s.a[0], s.a[1]);
        X6_USE("v"(s.a[2]), "v"(s.a[3]), "v"(s.m[0]));
    } else if constexpr (K == 3) {
        s.b[0] = s.a[0] - x6_lo(s.m[0]); s.b[1] = s.a[1] - x6_hi(s.m[0]);
        s.m[1] = x6_rn2(s.a[2], s.a[3]);
        X6_USE("v"(s.b[0]), "v"(s.b[1]), "v"(s.m[1]));
    } else {
        s.b[2] = s.a[2] - x6_lo(s.m[1]); s.b[3] = s.a[3] - x6_hi(s.m[1]);
        {                                // the next group's column stage may not start before this point; the two packs below separate the pin from it
            constexpr int JX = G < 3 ? J : ((J + 1) & 3), GX = G < 3 ? G + 1 : 0;      // (behind group 3: group 0 of the next point, built in the next period)
            constexpr int TA = JX == 0 ? 0 : JX == 2 ? 2 : 1, TB = JX == 0 ? 2 : JX == 1 ? 2 : JX == 2 ? 1 : 3;
            X6_PIN("+v"(T[TA][GX]), "+v"(T[TB][GX]), "+v"(s.b[2]), "+v"(s.b[3]));
        }
        s.l[0] = x6_rn2(s.b[0], s.b[1]); s.l[1] = x6_rn2(s.b[2], s.b[3]);
    }
}
// the pieces of channel quad QQ of tile block TB -> the MFMA data operands of buffer vb (dwords 2 QQ, 2 QQ + 1 of pieces h, m, l)
template <int TB, int QQ> __device__ __forceinline__ void x6_take_pieces(const X6Split& s, x6_i32x4 (&vb)[6]) {
    vb[3 * TB + 0][2 * QQ] = (int)s.h[0]; vb[3 * TB + 0][2 * QQ + 1] = (int)s.h[1];
    vb[3 * TB + 1][2 * QQ] = (int)s.m[0]; vb[3 * TB + 1][2 * QQ + 1] = (int)s.m[1];
    vb[3 * TB + 2][2 * QQ] = (int)s.l[0]; vb[3 * TB + 2][2 * QQ + 1] = (int)s.l[1];
}
// one (tile block, channel quad) group G = 2 TB + QQ of point J, step K
template <int K, int J, int G> __device__ __forceinline__ void x6_build_step(X6Split& s, f32x4 (&T)[4][4], x6_i32x4 (&vb)[6]) {
    x6_split_step<K, J, G>(s, T);
    if constexpr (K == 4) x6_take_pieces<(G >> 1), (G & 1)>(s, vb);
}

// raw rows ra, rb of patch column C, group G (tile block G >> 1: tiles 32 (G >> 1) + li, channel quad 2 lh + (G & 1)) from D buffer DPR
template <int DPR, int C, int G> __device__ __forceinline__ void x6_read_rows(f32x4 (&dd)[2], unsigned d_a, unsigned d_b) {
    constexpr int OFF = DPR * kX6DB + (G & 1) * kX6PlaneB + (G >> 1) * 8 * kX6RowB + x6_pos_of(C) * 16;
    X6_RD128(dd[0], d_a, OFF);
    X6_RD128(dd[1], d_b, OFF);
}
// row stage of the wave's point row: d[ra] + sgn d[rb]  (r = 0: d0 - d2;  1: d1 + d2;  2: d2 - d1;  3: d1 - d3; an fma by +-1 is exact).
// In two parts: the counted wait for the pair goes in FRONT of the MFMA whose shadow holds the four fmas -- the wait statement carries the two
// 128-bit quads as operands, and a vector instruction that writes into such an operand directly behind the statement gets an s_nop from the
// compiler; with the MFMA in between none is owed.
template <int PENDING> __device__ __forceinline__ void x6_row_wait(f32x4 (&dd)[2]) {
    if constexpr (PENDING == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dd[0]), "+v"(dd[1]));
    else if constexpr (PENDING == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(dd[0]), "+v"(dd[1]));
    else if constexpr (PENDING == 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(dd[0]), "+v"(dd[1]));
    else asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(dd[0]), "+v"(dd[1]));
}
__device__ __forceinline__ void x6_row_fma(f32x4& t, f32x4 (&dd)[2], float sgn) {
    // (scalar fmas written out: left to the compiler they become v_pk_fma_f32, which costs an MFMA-paced stream more than two plain ones;
    //  volatile: they stay in this gap without an X6_PIN over the 128-bit result)
#pragma unroll
    for (int e = 0; e < 4; ++e) { float r; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(dd[1][e]), "v"(sgn), "v"(dd[0][e])); t[e] = r; }
}
template <int PENDING> __device__ __forceinline__ void x6_row_stage(f32x4& t, f32x4 (&dd)[2], float sgn) {
    x6_row_wait<PENDING>(dd);
    x6_row_fma(t, dd, sgn);
}
#define X6_TIE6(f) "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5])

// End of period J: the weight fragments the NEXT period multiplies with have landed.  Vector-memory operations return in order, so
// vmcnt(n) = "all but the n youngest".  NS = deferred output stores this period issued behind its loads (x6_period); they may stay in flight.
//   period 0 (loads u(1), then u(3), then the store; no DMA): u(1) -> 6 + NS younger.  The barrier behind it needs every DMA piece of the wave:
//            they are older than the loads, so they have landed too.
//   period 1 (u(2), 2 pieces, store): 2 + NS; u(3) is older and has landed with it.
//   period 2 (2 pieces): nothing to wait for -- period 3 multiplies with u(3).
//   period 3 (the next chunk's u(0), 2 pieces, store): 2 + NS.
// A store issued in period X is older than the loads of period X + 1, so it has until the end of that period to leave the CU: one or two stores
// per wave and period are well inside the ~75 cycles per KB a CU's store path sustains.
template <int J, int NS> __device__ __forceinline__ void x6_period_wait(x6_i32x4 (&uf)[2][6], x6_i32x4 (&ue)[6]) {
    static_assert(NS >= 0 && NS <= 2, "");
#define X6_VMW(N, F) asm volatile("s_waitcnt vmcnt(" #N ")" : X6_TIE6(F) :: "memory")
    if constexpr (J == 0) { if constexpr (NS == 0) X6_VMW(6, uf[1]); else if constexpr (NS == 1) X6_VMW(7, uf[1]); else X6_VMW(8, uf[1]); }
    else if constexpr (J == 1) {
        if constexpr (NS == 0) asm volatile("s_waitcnt vmcnt(2)" : X6_TIE6(uf[0]), X6_TIE6(ue) :: "memory");
        else if constexpr (NS == 1) asm volatile("s_waitcnt vmcnt(3)" : X6_TIE6(uf[0]), X6_TIE6(ue) :: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" : X6_TIE6(uf[0]), X6_TIE6(ue) :: "memory");
    }
    else if constexpr (J == 3) { if constexpr (NS == 0) X6_VMW(2, uf[0]); else if constexpr (NS == 1) X6_VMW(3, uf[0]); else X6_VMW(4, uf[0]); }
#undef X6_VMW
}

// The output stores of a tile are DEFERRED into the first four chunks of the next one (every tile has at least four): a CU's store path takes
// ~75 cycles per 1-KB store instruction, so the 64 stores of a tile issued back to back held the four waves for ~4,700 cycles with the
// matrix pipe idle (profiles/r05_x6_timeline.txt); one store per wave and period (two in period 2, which loads nothing) disappears behind the
// MFMAs.  Round 5, late: TWO per wave and period, all sixteen in chunks 0 and 1 (two chunk copies with stores instead of four: 20 KB less code, the
// same speed); every read of the transposed area is over long before another wave can reach the next end-of-tile exchange, which writes there.
// The finished values wait in
// the wave's transposed area (x6_finish); state of the wave's pending tile:
struct X6Pending {
    float* base;             // wave-uniform: the tile block's first pixel, the output tile's first channel -- or the sink when nothing is pending
    unsigned off;            // per lane, bytes: the lane's pixel of pass 0 and its channel quad
    unsigned rowstep, colstep;      // wave-uniform, bytes: two image rows / four pixels (0 when nothing is pending: all 16 stores hit the lane's sink slot)
};
__device__ float g_x6_sink[256];
// (s_nop 2: a store of more than 64 bits reads its data registers for a few cycles after issue; the compiler, which keeps that distance for
// its own stores, does not see into the asm and may hand the data registers to the very next vector instruction -- seen: the first dword
// of one pass of the BatchNorm-sum kernels overwritten by the next store's address arithmetic)
#define X6_STORE(voff, data, sbase) asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 2" :: "v"(voff), "v"(data), "s"(sbase) : "memory")

// One PERIOD: the 24 MFMAs of point J of a chunk with D parity DP (operand buffers J & 1), and everything that runs in their shadow:
//   * the data operand of the NEXT point JN = J + 1 (of the next chunk for J = 3): column stage + split of 4 groups x 5 steps (gaps n % 6 != 5);
//   * the row stage of ONE patch column (PC: 3, 0, 2, 1 for J = 0..3; column 3 belongs to this chunk, the others to the next one) in the
//     gaps n % 6 == 5, its two ds_read_b128 issued four gaps earlier -- so that every column register set is written right after its last
//     reader: no double buffering of the 64 row-stage registers;
//   * weight fragments (global loads from gap 0 on: point 1 and point 3 in period 0, point 2 in period 1, the next chunk's point 0 in period 3)
//     and, for J >= 1, two LDS-DMA pieces of the patch of chunk c + 2 into the buffer chunk c left behind at the barrier after period 0 (gaps
//     6, 8; 12, 14 in period 1: right behind the weight loads -- a piece that is the first touch of its cache lines holds up every later
//     vector-memory instruction of the wave until it returns).
// At the end the wave waits for the fragments it loaded (vector-memory operations return in order: everything older has landed too,
// in particular the wave's DMA pieces of earlier periods), and after period 0 all waves meet: D(c) is dead, D(c + 1) complete.
// SI >= 0: this period also issues NS deferred stores SI, SI + 1 (passes of the previous tile's quarter): ds_read_b128 from the transposed
// area at gaps 2, 3, global_store_dwordx4 at gap 10 (16 in periods 0 and 1, behind their loads and pieces) and, for the second one, gap 16;
// the lgkmcnt immediates below count them.
template <int J, int DP, bool FIRST, int SI, int NS>
__device__ __forceinline__ void x6_period(f32x16 (&acc)[16], f32x4 (&T)[4][4], f32x4 (&dd)[2][2], x6_i32x4 (&uf)[2][6], x6_i32x4 (&ue)[6], x6_i32x4 (&vf)[2][6], X6Split& sp,
                                          unsigned d_a, unsigned d_b, float sgn, const char* us, unsigned voff0, unsigned voff1,
                                          const float* (&dptr)[6], unsigned lds_w, const X6Pending& pend, unsigned tr_lane, long long (&tl)[16]) {
    static_assert(NS == 0 || ((NS == 1 || NS == 2) && SI >= 0), "up to two deferred stores per period");
    constexpr int GS = J <= 1 ? 16 : 10;                         // the gap of the (first) deferred store: behind the period's loads and pieces
    f32x4 sv, sv2;
    constexpr int CB = J & 1, NB = CB ^ 1;
    // weight fragments: u(0), u(2) live in uf[0], u(1) in uf[1], u(3) in ue.  Period 0 loads u(1) AND u(3) (twelve loads, no DMA piece); period 1
    // loads u(2); period 2 loads nothing; period 3 loads the next chunk's u(0).  So no weight load is issued in the period behind the chunk's
    // first-touch DMA pieces (period 1's: they return after ~1900 cycles from HBM and hold up every vector-memory instruction issued meanwhile --
    // profiles/r05_x6_timeline.txt), and the heaviest period for the CU's 64 B/clk vector-memory path carries 48 KB instead of 56 (u(3) used to ride
    // in period 1 with u(2) and two pieces: moving it is worth 1-3.5 % per layer, profiles/r05_x6_load_plan.txt).
    x6_i32x4 (&ucur)[6] = J == 3 ? ue : uf[CB];
    x6_i32x4 (&unew)[6] = J == 0 ? uf[1] : uf[0];
    constexpr int NLOAD = J == 0 ? 12 : J == 2 ? 0 : 6;
#if (UNET_X6_ABLATE & 8)
    long long q0, q1, q2;
    X6_STAMP(q0);
#endif
    constexpr int JN = (J + 1) & 3;
    constexpr int PC = J == 0 ? 3 : J == 1 ? 0 : J == 2 ? 2 : 1;
    constexpr int DPR = J == 0 ? DP : (DP ^ 1);
    constexpr int ND = J == 0 ? 0 : 2;
    constexpr int DJ = J == 0 ? 0 : 2 * (J - 1);
#pragma unroll
    for (int n = 0; n < 24; ++n) {
        const int q = n >> 2, cb = (n >> 1) & 1, tb = n & 1;
        // piece products, small to large: (m,m) (l,h) (h,l) (m,h) (h,m) (h,h);  u = weights (rows = channels), v = data (columns = tiles)
        const int ui = q == 0 ? 1 : q == 1 ? 2 : q == 2 ? 0 : q == 3 ? 1 : 0;
        const int vi = q == 0 ? 1 : q == 1 ? 0 : q == 2 ? 2 : q == 3 ? 0 : q == 4 ? 1 : 0;
#if !(UNET_X6_ABLATE & 32)
        if (FIRST && q == 0) X6_MFMA0(acc[4 * J + 2 * cb + tb], ucur[2 * ui + cb], vf[CB][3 * tb + vi]);
        else X6_MFMA(acc[4 * J + 2 * cb + tb], ucur[2 * ui + cb], vf[CB][3 * tb + vi]);
#endif
        // ---- in the shadow of MFMA n
        constexpr int L0 = 0;                                            // first gap with a weight load (period 3, measured: starting at gap 8 instead -- further behind period 1's first-touch pieces -- turns a 250-cycle issue stall into a 300-cycle wait)
        if (n >= L0 && n < L0 + NLOAD && !(UNET_X6_ABLATE & 128)) {      // weight fragments [piece][channel block], 1 KB each
            if (n - L0 < 4) X6_LDU(unew[n - L0], voff0, us, (n - L0) * 1024);
            else if (n - L0 < 6) X6_LDU(unew[n - L0], voff1, us, (n - L0 - 4) * 1024);
            else if (n < 10) X6_LDU(ue[n - 6], voff0, us + 2 * kX6UPoint, (n - 6) * 1024);     // (period 0: point 3 lies two points behind point 1; a scalar add,
            else X6_LDU(ue[n - 6], voff1, us + 2 * kX6UPoint, (n - 10) * 1024);                //  not two more per-lane offsets kept live across the loop)
        }
        if (!(UNET_X6_ABLATE & 512)) {                                   // raw rows of the four groups of column PC: 5 / 10 gaps ahead of their row stage
            if (n == 0) x6_read_rows<DPR, PC, 0>(dd[0], d_a, d_b);
            if (n == 1) x6_read_rows<DPR, PC, 1>(dd[1], d_a, d_b);
            if (n == 7) x6_read_rows<DPR, PC, 2>(dd[0], d_a, d_b);
            if (n == 13) x6_read_rows<DPR, PC, 3>(dd[1], d_a, d_b);
        }
        // (M0 is written one gap ahead of the piece that uses it: the MFMA in between is the wait state the hardware asks for, instead of an s_nop)
        if (ND && n == (J == 1 ? 11 : 5) && !(UNET_X6_ABLATE & 64)) X6_DMA_M0(lds_w, DP * kX6DB + DJ * 4096);
        if (ND && n == (J == 1 ? 12 : 6) && !(UNET_X6_ABLATE & 64)) X6_DMA_GO(dptr[DJ]);
        if (ND && n == (J == 1 ? 13 : 7) && !(UNET_X6_ABLATE & 64)) X6_DMA_M0(lds_w, DP * kX6DB + (DJ + 1) * 4096);
        if (ND && n == (J == 1 ? 14 : 8) && !(UNET_X6_ABLATE & 64)) X6_DMA_GO(dptr[DJ + 1]);
        if (NS >= 1 && n == 2) X6_RD128(sv, tr_lane, (SI < 0 ? 0 : SI) * 2 * kX6TileB);
        if (NS == 2 && n == 3) X6_RD128(sv2, tr_lane, (SI < 0 ? 0 : SI + 1) * 2 * kX6TileB);
        // LDS operations retire in order; per period: row pairs at gaps 0, 1, the deferred reads at 2 (, 3), row pairs at 7, 13.  Every wait
        // names what may still be in flight BEHIND the read it needs.
        if (NS >= 1 && n == GS) {
            if (GS == 16) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(sv));                     // behind it: pairs 2, 3
            else if (NS == 2) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(sv));                 // the second deferred read, pair 2
            else asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(sv));                              // pair 2
            const unsigned so = pend.off + (unsigned)(SI >> 2) * pend.rowstep + (unsigned)(SI & 3) * pend.colstep;
            if (!(UNET_X6_ABLATE & 1024)) X6_STORE(so, sv, pend.base);
        }
        if (NS == 2 && n == 16) {
            asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(sv2));                                  // pairs 2, 3
            const unsigned so = pend.off + (unsigned)((SI + 1) >> 2) * pend.rowstep + (unsigned)((SI + 1) & 3) * pend.colstep;
            if (!(UNET_X6_ABLATE & 1024)) X6_STORE(so, sv2, pend.base);
        }
        // the waits of the row stages in the NEXT gap (x6_row_wait), at the head of this one: the split step's instructions separate the statement
        // from the fmas; no LDS operation is issued in between, so the counts are the next gap's
        if (n == 4) x6_row_wait<2 + NS>(dd[0]);                                            // behind pair 0: pair 1 and the deferred reads
        if (n == 10) x6_row_wait<((NS >= 1 && GS == 16) ? 3 : NS == 2 ? 3 : 2)>(dd[1]);      // behind pair 1: pair 2 (+ a deferred read not yet waited for)
        if (n == 16) x6_row_wait<2>(dd[0]);
        if (n == 22) x6_row_wait<0>(dd[1]);
        if (n % 6 == 5) {
            if (n == 5) x6_row_fma(T[PC][0], dd[0], sgn);
            if (n == 11) x6_row_fma(T[PC][1], dd[1], sgn);
            if (n == 17) x6_row_fma(T[PC][2], dd[0], sgn);
            if (n == 23) x6_row_fma(T[PC][3], dd[1], sgn);
            if (ND && n == 17) { dptr[DJ] += 16; dptr[DJ + 1] += 16; X6_PIN("+v"(dptr[DJ]), "+v"(dptr[DJ + 1])); }     // the two pointers move on by one chunk
        } else {
            const int sidx = n - n / 6;                                  // 0..19: group sidx / 5, step sidx % 5
#define X6_BS(S) if (sidx == S) x6_build_step<S % 5, JN, S / 5>(sp, T, vf[NB]);
            X6_BS(0) X6_BS(1) X6_BS(2) X6_BS(3) X6_BS(4) X6_BS(5) X6_BS(6) X6_BS(7) X6_BS(8) X6_BS(9)
            X6_BS(10) X6_BS(11) X6_BS(12) X6_BS(13) X6_BS(14) X6_BS(15) X6_BS(16) X6_BS(17) X6_BS(18) X6_BS(19)
#undef X6_BS
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#if (UNET_X6_ABLATE & 8)
    X6_STAMP(q1);
    x6_period_wait<J, NS>(uf, ue);
    X6_STAMP(q2);
    tl[2 * J] += q1 - q0; tl[2 * J + 1] += q2 - q1;
    if (J == 0) { asm volatile("s_barrier" ::: "memory"); long long q3; X6_STAMP(q3); tl[8] += q3 - q2; tl[9] += 1; }
#else
    x6_period_wait<J, NS>(uf, ue);
    if (J == 0) asm volatile("s_barrier" ::: "memory");
#endif
    asm volatile("" : X6_TIE6(vf[NB]));
}

struct X6Args {
    WinoFusedArgs f;             // x, bias, out, geometry, stats, pad; f.Uc unused
    const uint16_t* U6;          // [Nout / 64][K / 16][point row 4][point 4][piece 3][channel block 2][lane 64][8] bf16
};

// The finished quarter of owner wave (tile block tb, channel block cb): lane (li, lh) holds tile li, channels 16 lh + e of the block in
// element e of y[out row][out col] (the weight operand's rows are ordered for this: row 8 g + 4 lh + i of a fragment = channel 16 lh + 4 g + i).
// Stored that way every lane would write 16-byte pieces of 64 different pixels per instruction.  The quarter goes through the wave's own
// (now free) exchange area instead -- [tile 32][pixel 4][32 channels] fp32, tile stride 528 B: conflict-free both ways -- and comes back with
// lane = (pixel L / 8, channel quad L % 8): a store instruction then covers 8 pixels x 128 contiguous bytes, the lane's channel quad is
// the same in all 16 passes (bias: 4 values; BatchNorm sums: 2 x 4 registers instead of 2 x 16), and the STATS 2 reads of the producer's
// activation are coalesced the same way.  STATS 1: sum, sum of squares of the stored values; STATS 2: sum dy, sum dy r.
// -> the pending-store state of this tile (x6_period issues the stores); a tile block that sticks out of the image is stored here instead
// (per-lane masks) and leaves nothing pending.  bias16: the lane's 16 channels of the accumulator layout.
// STATS 2 reads the producer's saved activation at the finished quarter's pixels (x6_finish), lines written a whole forward pass ago: loaded where
// they are used, every tile waited ~2.3 us for them (the data gradient ran 4-16 % slower than the forward kernel on the same shape, most on the
// full-resolution layers with their many short tiles), and the 64 registers they fill are not free before the exchange is over.  So the same 16
// addresses are TOUCHED at the start of the end-of-tile work -- 16 loads into one scratch register, never read -- and the round trip to HBM passes behind
// the column stage and the exchange; x6_finish's own loads then hit the L2.  `scratch` must stay allocated until those loads have landed: the
// caller ties it to a statement behind x6_finish (vector-memory operations return in order: x6_finish's loads are younger).
// Both go through ONE buffer descriptor over the saved activation with the tile block's first pixel in the scalar offset and a single per-lane offset:
// a load is a scalar add and the instruction (an address with bounds checks and a select per load was 250 instructions for the sixteen touches and as
// many again in x6_finish -- kernel size is performance here, see the chunk macros in x6_stream_body).  Pixels outside the image but inside the tensor
// read their neighbours' values (x6_finish skips those passes), offsets past the tensor's end are rejected by the range check; no channel of
// [c0, c1) -> zero records, every load returns zeros.
struct X6Saved { __amdgpu_buffer_rsrc_t srd; unsigned soff, vlane, rowstep, colstep; };
__device__ __forceinline__ X6Saved x6_saved(const WinoFusedArgs& p, int img, int by, int bx, int n0, int tb, int cb, int lane_in) {
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const int cq = lane & 7, pl = lane >> 3;
    const bool rok = n0 >= p.bn_c0 && n0 < p.bn_c1;
    const int oa = (pl >> 1) & 1, ob = pl & 1;
    X6Saved r;
    const unsigned bytes = (unsigned)((((size_t)p.N * p.H * p.W - 1) * p.bn_ldr + (p.bn_c1 - p.bn_c0)) * 4);      // (< 4 GB: checked on the host)
    r.srd = __builtin_amdgcn_make_buffer_rsrc((void*)p.bn_r, 0, rok ? (int)bytes : 0, 0x00020000);
    r.soff = (unsigned)((((size_t)(img * p.H + 16 * by) * p.W + 16 * bx) * p.bn_ldr + (rok ? n0 - p.bn_c0 : 0)) * 4);
    r.vlane = (unsigned)((((8 * tb + oa) * p.W + 2 * (pl >> 2) + ob) * p.bn_ldr + 32 * cb + 4 * cq) * 4);
    r.rowstep = (unsigned)(2 * p.W * p.bn_ldr) * 4u; r.colstep = (unsigned)(4 * p.bn_ldr) * 4u;
    return r;
}
// (one dword per lane: eight lanes still cover each 128-byte line, and a quarter of the bytes crosses the CU's 64 B/clk vector-memory path --
//  sixteen 16-byte touches held the wave's instruction stream for ~1 300 cycles)
__device__ __forceinline__ void x6_touch_one(float& scratch, const X6Saved& r, int q) {
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "+v"(scratch) : "v"(r.vlane), "s"(r.srd), "s"(r.soff + (unsigned)(q >> 2) * r.rowstep + (unsigned)(q & 3) * r.colstep) : "memory");
}

template <int STATS>
__device__ __forceinline__ X6Pending x6_finish(float (&y)[2][2][16], const WinoFusedArgs& p, int img, int by, int bx, int n0, int tb, int cb, int lane_in,
                                               unsigned t_area, const f32x4 (&bias16)[4], f32x4& s1, f32x4& s2) {
    int lane = lane_in;
    asm volatile("" : "+v"(lane));                                // (everything per-lane below is re-derived here: kept live across the chunk loop it is spilled)
    const int li = lane & 31, lh = lane >> 5;
    const unsigned tw = t_area + (unsigned)(li * kX6TileB + lh * 64);
    int relu = p.relu;
    asm volatile("" : "+s"(relu));                                // (a scalar select here; hoisted out of the tile loop the floor became a spilled vector register)
    const float lo = relu ? 0.f : -__builtin_inff();
    // STATS 2: the producer's saved activation at the lane's 16 pixels (reader layout below), four loads behind each quarter of the transposed
    // writes -- that quarter's 16 registers of y are free by then, and the L2 round trip of the loads passes behind the remaining writes
    f32x4 rall[16];
    X6Saved rs;
    if constexpr (STATS == 2) rs = x6_saved(p, img, by, bx, n0, tb, cb, lane_in);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(y[a][b][4 * g + e] + bias16[g][e], lo);
                X6_WR128(tw, (2 * a + b) * 128 + g * 16, v);
            }
            if constexpr (STATS == 2) {
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int q = 4 * (2 * a + b) + q4;
                    rall[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs.srd, (int)rs.vlane, (int)(rs.soff + (unsigned)(q >> 2) * rs.rowstep + (unsigned)(q & 3) * rs.colstep), 0));
                }
            }
        }
    const int cq = lane & 7, pl = lane >> 3;                     // reader: channel quad, pixel of a pass (tile pl >> 2 of the pass, pixel pl & 3)
    const unsigned tr = t_area + (unsigned)((pl >> 2) * kX6TileB + (pl & 3) * 128 + cq * 16);
    const int oa = (pl >> 1) & 1, ob = pl & 1;
    // addresses: a wave-uniform 64-bit base (the tile block's first pixel, the output tile's first channel) + a 32-bit lane offset
    // (16 rows x W x ld of the tensor: far below 2^31 elements for every shape the entry points admit)
    const int ch = 32 * cb + 4 * cq;
    const int gy = 16 * by + 8 * tb + oa, gx = 16 * bx + 2 * (pl >> 2) + ob;       // the lane's pixel in pass (hp = 0, k = 0)
    float* const ob_ = p.out + ((size_t)(img * p.H + 16 * by) * p.W + 16 * bx) * p.ldo + n0;
    const int pix0 = (8 * tb + oa) * p.W + 2 * (pl >> 2) + ob;                     // pixel offset of that pixel from the block's first pixel
    const bool edge = 16 * by + 16 > p.H || 16 * bx + 16 > p.W;                    // wave-uniform
    // two passes at a time (half of tile row h2 / 2 of the block), the next pair's reads in flight while this pair is summed: one LDS round
    // trip per tile instead of eight (no MFMA runs here: every cycle of this loop is matrix-pipe time).  EDGE = the tile block sticks out of the
    // image (wave-uniform): only then are the per-lane in-image tests and the direct stores compiled in -- as one loop they cost every tile sixteen
    // exec-mask branches.
    auto sums = [&](auto edge_c) {
        constexpr bool EDGE = decltype(edge_c)::value;
        f32x4 v[2][2];
        X6_RD128(v[0][0], tr, 0); X6_RD128(v[0][1], tr, 2 * kX6TileB);
#pragma unroll
        for (int h2 = 0; h2 < 8; ++h2) {
            const int hp = h2 >> 1, k0 = 2 * (h2 & 1), cur = h2 & 1;
            if (h2 < 7) {
                const int hn = (h2 + 1) >> 1, kn = 2 * ((h2 + 1) & 1);
                X6_RD128(v[cur ^ 1][0], tr, (4 * hn + kn) * 2 * kX6TileB); X6_RD128(v[cur ^ 1][1], tr, (4 * hn + kn + 1) * 2 * kX6TileB);
                asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(v[cur][0]), "+v"(v[cur][1]));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[cur][0]), "+v"(v[cur][1]));
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if constexpr (EDGE) {
                    if (!(gy + 2 * hp < p.H && gx + 4 * (k0 + k) < p.W)) continue;
                    if (!(UNET_X6_ABLATE & 1024)) *reinterpret_cast<f32x4*>(ob_ + (unsigned)((pix0 + 2 * hp * p.W + 4 * (k0 + k)) * p.ldo + ch)) = v[cur][k];
                }
                if (STATS == 1) { s1 += v[cur][k]; s2 += v[cur][k] * v[cur][k]; }
                if (STATS == 2) { s1 += v[cur][k]; s2 += v[cur][k] * rall[4 * hp + k0 + k]; }
            }
        }
    };
    if (edge) sums(std::true_type{});
    else if (STATS != 0) sums(std::false_type{});
    X6Pending pd;
    pd.base = edge ? g_x6_sink : ob_;
    pd.off = edge ? (unsigned)lane * 16u : (unsigned)(pix0 * p.ldo + ch) * 4u;
    pd.rowstep = edge ? 0u : (unsigned)(2 * p.W * p.ldo) * 4u;
    pd.colstep = edge ? 0u : (unsigned)(4 * p.ldo) * 4u;
    return pd;
}
// Per-lane running sums -> one row of partials per wave (the layout of wf_write_stats: stat_part[tn][row][64 channels][2],
// row = 2 * (first tile / nt) + tile block): the lanes with one channel quad (lane % 8) are reduced, lanes 0..7 write.
__device__ __forceinline__ void x6_write_stats(const WinoFusedArgs& p, int t0, int rows_per_tn, int tb, int cb, int lane, f32x4 s1, f32x4 s2) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int m = 8; m < 64; m <<= 1) { s1[e] += __shfl_xor(s1[e], m, 64); s2[e] += __shfl_xor(s2[e], m, 64); }
    if (lane >= 8) return;
    const int tn = t0 % p.nt, row = 2 * (t0 / p.nt) + tb;
    float* o = p.stat_part + ((size_t)tn * rows_per_tn + row) * 128 + 2 * (32 * cb + 4 * lane);
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[2 * e] = s1[e]; o[2 * e + 1] = s2[e]; }
}

// Column stage of wave WV's point row at the end of a tile: z0 = m0 + m1 + m2, z1 = m1 - m2 - m3 over the row's four points, per quarter (channel
// block qd >> 1, tile block qd & 1) of the 64 x 64 tile; the quarter this wave owns stays in registers (zown), the others go to their owners' exchange areas.
// TOUCH: two of the sixteen saved-activation touches go behind each of the first eight groups of sixteen accumulator reads -- issued back to back in front
// of the stage they held the wave's instruction stream for 1 000 - 1 600 cycles.
template <int WV, bool TOUCH>
__device__ __forceinline__ void x6_column_stage(f32x16 (&acc)[16], unsigned x_lane, float (&zown)[2][16], float& scratch, const X6Saved& sv) {
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {                            // owner = wave qd
        const unsigned xw = x_lane + (unsigned)(qd * kX6XW) + (unsigned)((WV - (WV > qd ? 1 : 0)) * 8192);
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
            f32x4 z0, z1;
#pragma unroll
            for (int i = 0; i < 4; i += 2) {                     // two elements at a time: packed adds on aligned register pairs (no MFMA runs here)
                f32x2 m0, m1, m2, m3;
                asm("v_accvgpr_read_b32 %0, %1" : "=v"(m0.x) : "a"(acc[0 + qd][4 * e4 + i])); asm("v_accvgpr_read_b32 %0, %1" : "=v"(m0.y) : "a"(acc[0 + qd][4 * e4 + i + 1]));
                asm("v_accvgpr_read_b32 %0, %1" : "=v"(m1.x) : "a"(acc[4 + qd][4 * e4 + i])); asm("v_accvgpr_read_b32 %0, %1" : "=v"(m1.y) : "a"(acc[4 + qd][4 * e4 + i + 1]));
                asm("v_accvgpr_read_b32 %0, %1" : "=v"(m2.x) : "a"(acc[8 + qd][4 * e4 + i])); asm("v_accvgpr_read_b32 %0, %1" : "=v"(m2.y) : "a"(acc[8 + qd][4 * e4 + i + 1]));
                asm("v_accvgpr_read_b32 %0, %1" : "=v"(m3.x) : "a"(acc[12 + qd][4 * e4 + i])); asm("v_accvgpr_read_b32 %0, %1" : "=v"(m3.y) : "a"(acc[12 + qd][4 * e4 + i + 1]));
                const f32x2 r0 = (m0 + m1) + m2, r1 = wf_pk_sub(wf_pk_sub(m1, m2), m3);
                z0[i] = r0.x; z0[i + 1] = r0.y; z1[i] = r1.x; z1[i + 1] = r1.y;
            }
            if (qd == WV) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { zown[0][4 * e4 + i] = z0[i]; zown[1][4 * e4 + i] = z1[i]; }
            } else {
                X6_WR128(xw, (2 * e4) * 1024, z0);
                X6_WR128(xw, (2 * e4 + 1) * 1024, z1);
            }
            if constexpr (TOUCH) { if (qd < 2) { x6_touch_one(scratch, sv, 2 * (4 * qd + e4)); x6_touch_one(scratch, sv, 2 * (4 * qd + e4) + 1); } }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int STATS>
__device__ __forceinline__ void x6_stream_body(const X6Args& q, int ntiles) {
    const WinoFusedArgs& p = q.f;
    __shared__ __attribute__((aligned(1024))) char smem[kX6Smem];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);              // = the wave's point row r
    const int li = lane & 31, lh = lane >> 5;
    const int nchunks = p.K / 16;

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
    // DMA duty: piece wv + 4 j of a chunk = 16-byte slots 64 (wv + 4 j) + lane of [quad 4][row 18][pos 20].  The per-lane slot geometry is
    // re-derived at every tile switch (a few dozen integer instructions): kept live across the chunk loop it was spilled, and the reloads --
    // serialised vector-memory round trips behind the DMA pieces in flight -- cost microseconds per tile.
    auto tile_sources = [&](const TileCoord& c, const float* (&dp)[6]) {
        const int gy0 = 16 * c.by - 1, gx0 = 16 * c.bx - 1;
        const float* xb = p.x + ((long long)(c.img * p.H + gy0) * p.W + gx0) * p.ldx;
        int ln = lane;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int s = 64 * (wv + 4 * j) + ln;
            const int qd = s / 360, rem = s - 360 * qd, py = rem / 20, pos = rem - 20 * py;
            const bool real = s < 1440 && pos < 18;
            const int px = x6_col_of(pos < 18 ? pos : 0);
            const bool ok = real && (unsigned)(gy0 + py) < (unsigned)p.H && (unsigned)(gx0 + px) < (unsigned)p.W;
            dp[j] = ok ? xb + ((py * p.W + px) * p.ldx + 4 * qd) : padsrc + 4 * (qd & 3);
        }
    };
    auto u_source = [&](const TileCoord& c) {
        return reinterpret_cast<const char*>(q.U6) + ((size_t)c.tn * nchunks * 4 + wv) * kX6UChunkWave;
    };

    // ---- LDS byte addresses
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_f*)smem;
    // the lane's patch corner: tile (li / 8, li % 8) of a tile block, channel quads 2 lh, 2 lh + 1; raw rows ra / rb of the wave's point row
    const int ra = wv == 0 ? 0 : wv == 2 ? 2 : 1, rb = wv == 0 ? 2 : wv == 1 ? 2 : wv == 2 ? 1 : 3;
    const float sgn = wv == 1 ? 1.f : -1.f;
    const unsigned d_lane = lds0 + (unsigned)(2 * lh * kX6PlaneB + (li >> 3) * 2 * kX6RowB + (li & 7) * 16);
    const unsigned d_a = d_lane + (unsigned)(ra * kX6RowB), d_b = d_lane + (unsigned)(rb * kX6RowB);
    const unsigned lds_w = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + wv * 1024));       // this wave's first DMA piece, as an M0 value
    const unsigned voff0 = (unsigned)lane * 16u, voff1 = voff0 + 4096u;

    f32x16 acc[16];
    f32x4 T[4][4];
    f32x4 dd[2][2];
    x6_i32x4 uf[2][6], ue[6], vf[2][6];
    X6Split sp;
    const float* dptr[6]; const char* ucur; const char* unxt;
    int t = blockIdx.x;
    if ((gridDim.x & 7) == 0 && (p.nt & 7) != 0) t = (t & 7) * (int)(gridDim.x >> 3) + (t >> 3);       // XCD-aware renumbering, as winograd.hip
    const int t_first = t;
    f32x4 s1 = f32x4{0.f, 0.f, 0.f, 0.f}, s2 = f32x4{0.f, 0.f, 0.f, 0.f};
    X6Pending pend;                                                // nothing pending yet: the sixteen store slots of the first tile write the lane's sink slot
    pend.base = g_x6_sink; pend.off = (unsigned)lane * 16u; pend.rowstep = 0u; pend.colstep = 0u;
    const unsigned t_area = lds0 + kX6X + (unsigned)(wv * kX6XW);
    const unsigned tr_lane = t_area + (unsigned)(((lane >> 3) >> 2) * kX6TileB + ((lane >> 3) & 3) * 128 + (lane & 7) * 16);
    TileCoord tc = decode(t);
    tile_sources(tc, dptr);
    ucur = u_source(tc);

#ifdef UNET_X6_STAGGER
    // workgroups start UNET_X6_STAGGER x 64 cycles apart in 8 phases: identical tiles otherwise keep every CU's end-of-tile store burst in step
    for (int i = 0; i < (int)(blockIdx.x & 7) * UNET_X6_STAGGER; ++i) __builtin_amdgcn_s_sleep(1);
#endif
    // ---- prologue of the workgroup's first tile: D(0), D(1) -> LDS; u(point 0); row stage of columns 1, 0, 2 of chunk 0; V(point 0)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        X6_DMA_V(dptr[j], lds_w, j * 4096);
        X6_DMA_V(dptr[j] + 16, lds_w, kX6DB + j * 4096);
        dptr[j] += 32;                                           // next issue: chunk 2
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) { if (k < 4) X6_LDU(uf[0][k], voff0, ucur, k * 1024); else X6_LDU(uf[0][k], voff1, ucur, (k - 4) * 1024); }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" : X6_TIE6(uf[0]) :: "memory");
#define X6_COL(C) \
    x6_read_rows<0, C, 0>(dd[0], d_a, d_b); x6_row_stage<0>(T[C][0], dd[0], sgn); x6_read_rows<0, C, 1>(dd[0], d_a, d_b); x6_row_stage<0>(T[C][1], dd[0], sgn); \
    x6_read_rows<0, C, 2>(dd[0], d_a, d_b); x6_row_stage<0>(T[C][2], dd[0], sgn); x6_read_rows<0, C, 3>(dd[0], d_a, d_b); x6_row_stage<0>(T[C][3], dd[0], sgn);
    X6_COL(1) X6_COL(0) X6_COL(2)
#undef X6_COL
#define X6_G(G) x6_build_step<0, 0, G>(sp, T, vf[0]); x6_build_step<1, 0, G>(sp, T, vf[0]); x6_build_step<2, 0, G>(sp, T, vf[0]); x6_build_step<3, 0, G>(sp, T, vf[0]); x6_build_step<4, 0, G>(sp, T, vf[0]);
    X6_G(0) X6_G(1) X6_G(2) X6_G(3)
#undef X6_G
    asm volatile("" : X6_TIE6(vf[0]));

    long long tl[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (; t < ntiles; t += gridDim.x) {
#if (UNET_X6_ABLATE & 8)
        long long e0; X6_STAMP(e0);
#endif
        const TileCoord tcn = t + (int)gridDim.x < ntiles ? advance(tc) : tc;            // the last tile prefetches itself again
        unxt = u_source(tcn);
        // weight fragments of point J + 1 (period J of chunk c): this chunk's next point, the next chunk's point 0, or the next tile's
#define X6_US(c, J) ((J) < 3 ? ucur + (size_t)(c) * kX6UChunk + ((J) + 1) * kX6UPoint : ((c) + 1 < nchunks ? ucur + (size_t)((c) + 1) * kX6UChunk : unxt))
#define X6_PERIOD(J, DP, FIRST, c, SI, NS) x6_period<J, DP, FIRST, SI, NS>(acc, T, dd, uf, ue, vf, sp, d_a, d_b, sgn, X6_US(c, J), voff0, voff1, dptr, lds_w, pend, tr_lane, tl)
        // (the patch pieces issued from chunk nchunks - 2 on belong to the next tile: its pointers are formed in the one period without DMA)
        // X6_CHUNK_S: the chunk issues the previous tile's deferred stores S0 .. S0 + 7, two per period
        // The switch sits in front of chunk nchunks - 2, an EVEN chunk (K % 32 == 0: x6_shape_ok): only the loop's even chunk carries its ~150
        // instructions.  A chunk is ~7 KB of code; with six chunk copies (three with five stores, one with the sixteenth, two generic) and the switch
        // inlined in all of them the kernels were 68-72 KB.  In the INSTRUMENTED build (s_memtime stamps, ~3 KB more) that showed as a column stage of
        // 6 500 cycles in the kernels with BatchNorm sums against 4 000 in the plain kernel for the same instructions -- code that runs once per tile
        // re-fetched every tile -- and went away below ~67 KB; the product kernels measure the same at 69 KB and at 49 KB (same-box bench.py runs,
        // profiles/r05_x6_timeline.txt section 5), so the smaller form is kept for what it is: two store chunks, two generic ones, 48-51 KB.
#define X6_SWITCH(c) if ((c) == nchunks - 2) tile_sources(tcn, dptr);
#define X6_CHUNK_S(DP, FIRST, c, S0) \
        X6_PERIOD(0, DP, FIRST, c, S0, 2); X6_PERIOD(1, DP, FIRST, c, S0 + 2, 2); X6_PERIOD(2, DP, FIRST, c, S0 + 4, 2); X6_PERIOD(3, DP, FIRST, c, S0 + 6, 2);
#define X6_CHUNK(DP, c) \
        X6_PERIOD(0, DP, false, c, -1, 0); X6_PERIOD(1, DP, false, c, -1, 0); X6_PERIOD(2, DP, false, c, -1, 0); X6_PERIOD(3, DP, false, c, -1, 0);
        X6_CHUNK_S(0, true, 0, 0)
        X6_CHUNK_S(1, false, 1, 8)
        for (int c = 2; c < nchunks; c += 2) {
            X6_SWITCH(c) X6_CHUNK(0, c)
            X6_CHUNK(1, c + 1)
        }
#undef X6_CHUNK_S
#undef X6_SWITCH
#undef X6_CHUNK
#undef X6_PERIOD
#undef X6_US
        // ---- end of tile.  Column stage of the wave's point row (lane-local), halves to their owners through LDS, row stage at the owner
#if (UNET_X6_ABLATE & 8)
        long long e1; X6_STAMP(e1);
#endif
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // inline-asm MFMAs are invisible to the compiler's hazard recogniser
        f32x4 bias16[4];                                         // the bias of the lane's 16 channels (accumulator layout); its latency passes behind the column stage
        int lnb = lane;
        asm volatile("" : "+v"(lnb));                            // (the lane's part of the address is formed here: hoisted out of the tile loop it was spilled)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bias16[g] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.bias) bias16[g] = *reinterpret_cast<const f32x4*>(p.bias + tc.tn * 64 + 32 * (wv >> 1) + 16 * (lnb >> 5) + 4 * g);
        }
        int lnx = lane;
        asm volatile("" : "+v"(lnx));                            // (the exchange addresses are formed here: hoisted out of the tile loop they were spilled)
        const unsigned x_lane = lds0 + kX6X + (unsigned)lnx * 16u;
        float touched = 0.f;
        X6Saved saved{};
        if constexpr (STATS == 2) saved = x6_saved(p, tc.img, tc.by, tc.bx, tc.tn * 64, wv & 1, wv >> 1, lane);
        float zown[2][16];
        // (one straight-line copy per wave: with the owner test inside the loop every quarter cost two taken branches and the own quarter 32 register
        //  copies -- the accumulators are only READ here, so the four copies do not disturb their allocation)
        switch (wv) {
            case 0: x6_column_stage<0, STATS == 2>(acc, x_lane, zown, touched, saved); break;
            case 1: x6_column_stage<1, STATS == 2>(acc, x_lane, zown, touched, saved); break;
            case 2: x6_column_stage<2, STATS == 2>(acc, x_lane, zown, touched, saved); break;
            default: x6_column_stage<3, STATS == 2>(acc, x_lane, zown, touched, saved); break;
        }
#if (UNET_X6_ABLATE & 8)
        long long e3, e4, e5; X6_STAMP(e3);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        X6_STAMP(e4);
#else
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
        // owner: y[0][b] = z_b(row 0) + z_b(row 1) + z_b(row 2), y[1][b] = z_b(row 1) - z_b(row 2) - z_b(row 3)
        float y[2][2][16];
        {
            const float a0 = wv == 3 ? 0.f : 1.f, a1 = wv == 0 ? 0.f : wv == 1 ? 1.f : -1.f;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) { y[0][b][e] = a0 * zown[b][e]; y[1][b][e] = a1 * zown[b][e]; }
        }
        const unsigned xr = x_lane + (unsigned)(wv * kX6XW);
        {
            // value index vi = 2 e4 + b (1 KB each) of the three other rows, one value quad ahead of its use
            f32x4 zr[2][3];
            X6_RD128(zr[0][0], xr, 0 * 8192); X6_RD128(zr[0][1], xr, 1 * 8192); X6_RD128(zr[0][2], xr, 2 * 8192);
#pragma unroll
            for (int vi = 0; vi < 8; ++vi) {
                const int e4 = vi >> 1, b = vi & 1;
                if (vi < 7) { X6_RD128(zr[(vi + 1) & 1][0], xr, 0 * 8192 + (vi + 1) * 1024); X6_RD128(zr[(vi + 1) & 1][1], xr, 1 * 8192 + (vi + 1) * 1024); X6_RD128(zr[(vi + 1) & 1][2], xr, 2 * 8192 + (vi + 1) * 1024); }
                if (vi < 7) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(zr[vi & 1][0]), "+v"(zr[vi & 1][1]), "+v"(zr[vi & 1][2]));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(zr[vi & 1][0]), "+v"(zr[vi & 1][1]), "+v"(zr[vi & 1][2]));
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const int src = s < wv ? s : s + 1;
                    const float a0 = src == 3 ? 0.f : 1.f, a1 = src == 0 ? 0.f : src == 1 ? 1.f : -1.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        y[0][b][4 * e4 + i] = __builtin_fmaf(a0, zr[vi & 1][s][i], y[0][b][4 * e4 + i]);
                        y[1][b][4 * e4 + i] = __builtin_fmaf(a1, zr[vi & 1][s][i], y[1][b][4 * e4 + i]);
                    }
                }
            }
        }
#if (UNET_X6_ABLATE & 8)
        X6_STAMP(e5);
#endif
        pend = x6_finish<STATS>(y, p, tc.img, tc.by, tc.bx, tc.tn * 64, wv & 1, wv >> 1, lane, t_area, bias16, s1, s2);
        if constexpr (STATS == 2) asm volatile("" : "+v"(touched));          // (x6_touch_saved: the scratch register lives until here)
        ucur = unxt; tc = tcn;
#if (UNET_X6_ABLATE & 8)
        { long long e2; X6_STAMP(e2); tl[10] += e1 - e0; tl[11] += e2 - e1; tl[12] += 1; tl[13] += e3 - e1; tl[14] += e4 - e3; tl[15] += e5 - e4; }
#endif
    }
#if (UNET_X6_ABLATE & 8)
    if (blockIdx.x == 0 && tid == 0) for (int i = 0; i < 16; ++i) g_x6_timeline[i] = tl[i];
#endif
    // retire the prefetches of the tile that never runs (weight fragments, DMA pieces), then the last tile's deferred stores
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : X6_TIE6(uf[0]) :: "memory");
#pragma unroll
    for (int si = 0; si < 16; ++si) {
        f32x4 sv;
        X6_RD128(sv, tr_lane, si * 2 * kX6TileB);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sv));
        const unsigned so = pend.off + (unsigned)(si >> 2) * pend.rowstep + (unsigned)(si & 3) * pend.colstep;
        if (!(UNET_X6_ABLATE & 1024)) X6_STORE(so, sv, pend.base);
    }
    if (STATS) x6_write_stats(p, t_first, 2 * ((int)gridDim.x / p.nt), wv & 1, wv >> 1, lane, s1, s2);
}
__global__ __launch_bounds__(256, 1) void wino_x6_stream_kernel(X6Args q, int ntiles) { x6_stream_body<0>(q, ntiles); }
__global__ __launch_bounds__(256, 1) void wino_x6_stream_stats_kernel(X6Args q, int ntiles) { x6_stream_body<1>(q, ntiles); }
__global__ __launch_bounds__(256, 1) void wino_x6_stream_bnbwd_kernel(X6Args q, int ntiles) { x6_stream_body<2>(q, ntiles); }

// ---- weight operands: G g G^T in fp32 (as winograd.hip), then the exact three-piece split, in MFMA A-operand order ----------------------
//   U6[(((((n/64) * (K/16) + k/16) * 4 + r) * 4 + j) * 3 + piece) * 2 + (n%64)/32) * 512 + (row(n%32) + 32 * ((k%16)/8)) * 8 + k%8],   point xi = 4 r + j,
//   row(16 a + 4 g + i) = 8 g + 4 a + i
// : per 64-channel output tile, 16-channel reduce chunk and point ROW r one contiguous 24-KB block (a wave's stream), in it per point and
// piece the two 1-KB fragments a wave loads with one global_load_dwordx4 each (lane (li, lh) = fragment row li, reduce channels 8 lh + 0..7).
// mode 0 (forward): k = ci, n = co;  thread = 8 consecutive ci x one co (adjacent threads = adjacent co: coalesced reads of w, 16-byte stores).
// mode 1 (data gradient): k = co, n = ci, filter rotated by 180 degrees = the forward transform with points 0 and 3 swapped in both directions;
//         thread = 8 consecutive co x one ci (32 contiguous bytes of w per tap).
// scale / shift (forward only, nullable): BatchNorm-apply on load -- g is scaled by the PRODUCER's BatchNorm scale per input channel
// (floored as in winograd.hip's fold).
__device__ __forceinline__ void x6_pieces8(const float (&t)[8], x6_i32x4& h, x6_i32x4& m, x6_i32x4& l) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned hp = x6_rn2(t[2 * e], t[2 * e + 1]);
        const float a0 = t[2 * e] - x6_lo(hp), a1 = t[2 * e + 1] - x6_hi(hp);
        const unsigned mp = x6_rn2(a0, a1);
        const float b0 = a0 - x6_lo(mp), b1 = a1 - x6_hi(mp);
        h[e] = (int)hp; m[e] = (int)mp; l[e] = (int)x6_rn2(b0, b1);
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
// One item = 8 consecutive k (k8) x one n: ONE lane's 16 bytes of the fragments of all 16 points x 3 pieces.  Callers map threads so that
// consecutive n are neighbours: a wave's store instruction then covers two 512-byte runs.
__device__ __forceinline__ void x6_weight_item(const float* __restrict__ w, uint16_t* __restrict__ U6, int Ci, int Co, int mode, int k8, int n,
                                               const float* __restrict__ scale, const float* __restrict__ shift) {
    const int N = mode ? Ci : Co;
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
    const int nchunks = (mode ? Co : Ci) >> 4;
    (void)N;
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) {
        int r = xi >> 2, j = xi & 3;
        if (mode) { r = r == 0 ? 3 : (r == 3 ? 0 : r); j = j == 0 ? 3 : (j == 3 ? 0 : j); }
        x6_i32x4 h, m, l;
        x6_pieces8(t[xi], h, m, l);
        // fragment row of channel n % 32 = 16 a + 4 g + i:  8 g + 4 a + i  (a lane of the accumulator then holds 16 consecutive channels)
        const int nl = n & 31, frow = 8 * ((nl >> 2) & 3) + 4 * (nl >> 4) + (nl & 3);
        uint16_t* o = U6 + ((((size_t)(((n >> 6) * nchunks + c16) * 4 + r) * 4 + j) * 3) * 2 + ((n >> 5) & 1)) * 512 + (frow + 32 * half) * 8;
        *reinterpret_cast<x6_i32x4*>(o) = h;
        *reinterpret_cast<x6_i32x4*>(o + 2 * 512) = m;
        *reinterpret_cast<x6_i32x4*>(o + 4 * 512) = l;
    }
}
__global__ __launch_bounds__(256) void wino_x6_weight_kernel(const float* __restrict__ w, uint16_t* __restrict__ U6, int Ci, int Co, int mode) {
    const long items = (long)Ci * Co / 8;
    const long it = (long)blockIdx.x * 256 + threadIdx.x;
    const int N = mode ? Ci : Co;
    if (it < items) x6_weight_item(w, U6, Ci, Co, mode, 2 * (int)((it >> 1) / N) + (int)(it & 1), (int)((it >> 1) % N), nullptr, nullptr);
}
// jobs[j] = { w, U6, Ci | Co << 32, first block, mode, 0 }: every direction of every layer in ONE launch
__global__ __launch_bounds__(256) void wino_x6_weight_batch_kernel(const long long* __restrict__ jobs, int njobs) {
    int j = 0;
    while (j + 1 < njobs && (int)jobs[(j + 1) * 6 + 3] <= (int)blockIdx.x) ++j;
    const float* w = reinterpret_cast<const float*>(jobs[j * 6 + 0]);
    uint16_t* U6 = reinterpret_cast<uint16_t*>(jobs[j * 6 + 1]);
    const int Ci = (int)(jobs[j * 6 + 2] & 0xffffffffll), Co = (int)(jobs[j * 6 + 2] >> 32);
    const long it = ((long)blockIdx.x - (int)jobs[j * 6 + 3]) * 256 + threadIdx.x;
    const int mode = (int)jobs[j * 6 + 4], N = mode ? Ci : Co;
    if (it < (long)Ci * Co / 8) x6_weight_item(w, U6, Ci, Co, mode, 2 * (int)((it >> 1) / N) + (int)(it & 1), (int)((it >> 1) % N), nullptr, nullptr);
}

// BatchNorm-apply on load (see winograd.hip, wino_weight_fold_kernel): U6 = pieces of scale . transform(w), bias_out = bias + sum_taps shift . w,
// pad = -shift / scale.  ONE launch, two kinds of workgroup: the first `tblocks` transform 256 items each (the batch transform's item, with the
// producer's scale on the weights: as many workgroups as the plain transform has -- round 4 ran the items inside the Cout / 8 bias
// workgroups, 8 .. 128 of them on 256 CUs, 12-45 us per layer); the others own kX6FoldCo output channels and all input channels each and finish
// the folded bias in a fixed order (and the padding values).
constexpr int kX6FoldCo = 8, kX6FoldLanes = 256 / kX6FoldCo;
__global__ __launch_bounds__(256) void wino_x6_weight_fold_kernel(const float* __restrict__ w, const float* __restrict__ scale,
        const float* __restrict__ shift, const float* __restrict__ bias, uint16_t* __restrict__ U6, float* __restrict__ bias_out,
        float* __restrict__ pad, int Ci, int Co, int tblocks) {
    if ((int)blockIdx.x < tblocks) {
        const long it = (long)blockIdx.x * 256 + threadIdx.x;
        if (it < (long)Ci * Co / 8) x6_weight_item(w, U6, Ci, Co, 0, 2 * (int)((it >> 1) / Co) + (int)(it & 1), (int)((it >> 1) % Co), scale, shift);
        return;
    }
    __shared__ double sPart[kX6FoldLanes][kX6FoldCo];
    const int bb = (int)blockIdx.x - tblocks;
    const int col = threadIdx.x % kX6FoldCo, gl = threadIdx.x / kX6FoldCo;      // output channel of the workgroup, input-channel octet lane
    const int co = bb * kX6FoldCo + col;
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
    if (bb == 0 && threadIdx.x < 8) pad[Ci + threadIdx.x] = 0.f;
}

bool x6_shape_ok(int N, int H, int W, int K, int Nout) {
    return N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && K % 32 == 0 && K >= 64 && K <= kWinoFusedMaxK && Nout % 64 == 0;
}

int run_wino_x6(const float* x, int ldx, const uint16_t* U6, const float* bias, float* out, int ldo, int N, int H, int W,
                int K, int Nout, int relu, float* stat_part, hipStream_t st, const WinoBnBwd* bb, const float* pad, int max_workgroups) {
    X6Args q{};
    WinoFusedArgs& a = q.f;
    q.U6 = U6;
    a.pad = pad;
    a.x = x; a.Uc = nullptr; a.bias = bias; a.out = out; a.ldx = ldx; a.ldo = ldo; a.N = N; a.H = H; a.W = W; a.K = K; a.Nout = Nout; a.relu = relu;
    a.tby = (H / 2 + 7) / 8; a.tbx = (W / 2 + 7) / 8; a.nt = Nout / 64; a.stat_part = stat_part;
    const long blocks = (long)N * a.tby * a.tbx * a.nt;
    if (blocks <= 0 || blocks > 0x7fffffffL) return UNET_EINVAL;
    const int cus = unet_grid_slots(wino_stream_cus(), max_workgroups);
    const dim3 grid((unsigned)(blocks < cus ? blocks : cus));
    if (bb) {
        a.bn_r = bb->r; a.bn_ldr = bb->ldr; a.bn_c0 = bb->c0; a.bn_c1 = bb->c1;
        wino_x6_stream_bnbwd_kernel<<<grid, 256, 0, st>>>(q, (int)blocks);
    }
    else if (stat_part) wino_x6_stream_stats_kernel<<<grid, 256, 0, st>>>(q, (int)blocks);
    else                wino_x6_stream_kernel<<<grid, 256, 0, st>>>(q, (int)blocks);
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
    const int tblocks = (int)(((long)Cin * Cout / 8 + 255) / 256);
    wino_x6_weight_fold_kernel<<<(unsigned)(tblocks + (Cout + kX6FoldCo - 1) / kX6FoldCo), 256, 0, (hipStream_t)stream>>>(w, scale, shift, bias, (uint16_t*)U6, bias_out, pad, Cin, Cout, tblocks);
    return UNET_LAUNCH_STATUS();
}

// Forward: the arguments of unet_conv3x3_fwd_winograd_fused with U6 (unet_winograd_weight_transform_x6 mode 0 / _fold_x6) in place of Uc.
// stat_part rows = unet_conv3x3_fwd_winograd_fused_stats_rows[_wg] (the persistent grid is the same).  max_workgroups as there.
extern "C" int unet_conv3x3_fwd_winograd_x6_wg(const float* x, int ldx, const float* pad, const void* U6, const float* bias, float* out, int ldo,
        int N, int H, int W, int Cin, int Cout, int relu, float* stat_part, size_t stat_bytes, int max_workgroups, void* stream) {
    UNET_CHECK_ARG(x && U6 && out && x6_shape_ok(N, H, W, Cin, Cout));
    UNET_CHECK_ARG(ldx >= Cin && ldo >= Cout && ldx % 4 == 0 && ldo % 4 == 0 && unet_aligned16(x) && unet_aligned16(U6) && unet_aligned16(out));
    UNET_CHECK_ARG((!bias || unet_aligned16(bias)) && (!pad || unet_aligned16(pad)));
    if (stat_part) {
        const int rows = wino_stats_rows(N, H, W, Cin, Cout, max_workgroups);
        UNET_CHECK_ARG(rows > 0);
        if (stat_bytes < (size_t)(Cout / 64) * rows * 128 * sizeof(float)) return UNET_ENOSPC;
    }
    return run_wino_x6(x, ldx, (const uint16_t*)U6, bias, out, ldo, N, H, W, Cin, Cout, relu, stat_part, (hipStream_t)stream, nullptr, pad, max_workgroups);
}
extern "C" int unet_conv3x3_fwd_winograd_x6(const float* x, int ldx, const float* pad, const void* U6, const float* bias, float* out, int ldo,
        int N, int H, int W, int Cin, int Cout, int relu, float* stat_part, size_t stat_bytes, void* stream) {
    return unet_conv3x3_fwd_winograd_x6_wg(x, ldx, pad, U6, bias, out, ldo, N, H, W, Cin, Cout, relu, stat_part, stat_bytes, 0, stream);
}

// Data gradient: the arguments of unet_conv3x3_dgrad_winograd_fused with U6d (mode 1) in place of Ucd.
extern "C" int unet_conv3x3_dgrad_winograd_x6_wg(const float* dz, int lddz, const void* U6d, float* dx, int lddx,
        int N, int H, int W, int Cin, int Cout, const float* r_prev, int ldr, int c0, int c1,
        float* stat_part, size_t stat_bytes, int max_workgroups, void* stream) {
    UNET_CHECK_ARG(dz && U6d && dx && x6_shape_ok(N, H, W, Cout, Cin) && (r_prev == nullptr) == (stat_part == nullptr));
    UNET_CHECK_ARG(lddz >= Cout && lddx >= Cin && lddz % 4 == 0 && lddx % 4 == 0 && unet_aligned16(dz) && unet_aligned16(U6d) && unet_aligned16(dx));
    if (!r_prev) return run_wino_x6(dz, lddz, (const uint16_t*)U6d, nullptr, dx, lddx, N, H, W, Cout, Cin, 0, nullptr, (hipStream_t)stream, nullptr, nullptr, max_workgroups);
    UNET_CHECK_ARG(c0 >= 0 && c1 > c0 && c1 <= Cin && c0 % 64 == 0 && c1 % 64 == 0 && ldr >= c1 - c0 && ldr % 4 == 0 && unet_aligned16(r_prev));
    const int rows = wino_stats_rows(N, H, W, Cout, Cin, max_workgroups);
    UNET_CHECK_ARG(rows > 0);
    if (stat_bytes < (size_t)(Cin / 64) * rows * 128 * sizeof(float)) return UNET_ENOSPC;
    const WinoBnBwd bb{r_prev, ldr, c0, c1};
    return run_wino_x6(dz, lddz, (const uint16_t*)U6d, nullptr, dx, lddx, N, H, W, Cout, Cin, 0, stat_part, (hipStream_t)stream, &bb, nullptr, max_workgroups);
}
extern "C" int unet_conv3x3_dgrad_winograd_x6(const float* dz, int lddz, const void* U6d, float* dx, int lddx,
        int N, int H, int W, int Cin, int Cout, const float* r_prev, int ldr, int c0, int c1,
        float* stat_part, size_t stat_bytes, void* stream) {
    return unet_conv3x3_dgrad_winograd_x6_wg(dz, lddz, U6d, dx, lddx, N, H, W, Cin, Cout, r_prev, ldr, c0, c1, stat_part, stat_bytes, 0, stream);
}
