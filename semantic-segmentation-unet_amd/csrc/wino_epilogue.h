// Shared between the fp32-MFMA fused Winograd kernels (winograd.hip) and the BF16x6 ones (winograd_x6.hip): the argument block,
// the zero page the out-of-image halo pointers walk over, and the per-tile epilogue (output transform A^T m A, bias, ReLU, fused
// BatchNorm sums).  Both kernel families keep a wave's [32 channels x 32 tiles] x 16 Winograd points in 256 accumulator registers in
// the same element order, so the epilogue is the same code.
#pragma once
#include "common.h"

namespace {

struct WinoFusedArgs {
    const float* x; const float* Uc; const float* bias; float* out;
    int ldx, ldo, N, H, W, K, Nout, relu;
    int tbx, tby, nt;            // tile-block grid
    float* stat_part;            // BatchNorm statistics of the output (persistent kernel only), see wf_write_stats; or null
    const float* bn_r; int bn_ldr, bn_c0, bn_c1;      // STATS == 2 (data gradient): saved activation of the producer layer, channels [c0, c1)
    const float* pad;            // per-input-channel value of the positions outside the image (K floats + 8), or null = zeros: see unet_winograd_weight_fold
};
typedef __attribute__((address_space(3))) void lds_void_f;
constexpr int kWinoFusedMaxK = 4096;
__device__ __attribute__((aligned(256))) float g_zero_page_f[kWinoFusedMaxK + 8];   // zero source that out-of-image halo pointers walk over

// Epilogue of one output tile block: lane (li, lh) of wave (mi, ni) holds tile 32*mi + li and channels
// n0 + 32*ni + 8*g + 4*lh + {0..3}, g = 0..3, in accumulator elements 4g..4g+3 of every point.  Output transform A^T m A,
// bias and ReLU on channel pairs (packed fp32; the subtractions as inline asm - the compiler splits them into scalar
// v_sub_f32), then one 16-byte store per pixel and channel quad.  `bias4` = the lane's 16 bias values, loaded by the
// caller before the chunk loop so their latency is not paid here.
__device__ __forceinline__ f32x2 wf_pk_sub(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ void wf_load_bias(const WinoFusedArgs& p, int n0, int ni, int lh, f32x4 (&bias4)[4]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        bias4[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias) bias4[g] = *reinterpret_cast<const f32x4*>(p.bias + n0 + 32 * ni + 4 * lh + 8 * g);
    }
}
// STATS: also accumulate, per lane, the sum and the sum of squares of the stored values per channel pair (s1 / s2 [2g + h]):
// the BatchNorm that follows the layer (UNet/model.py:36) needs exactly these over all pixels, and this is the only place
// the values pass through registers anyway (saves a full read of the activation tensor per layer).
// STATS == 2 (data gradient whose output is the dy of a BatchNorm layer): the sums are sum(dy) and sum(dy * r) with r the saved
// activation of that layer at the same pixels (rv, loaded by the caller before the chunk loop) -- what the BatchNorm backward
// reduction needs (dbeta = sum dy, dgamma = invstd (sum dy r - mean sum dy)), again one full read of two tensors saved.
template <int STATS>
__device__ __forceinline__ void wf_epilogue(const f32x16 (&acc)[16], const WinoFusedArgs& p, int img, int by, int bx, int n0,
                                            int mi, int ni, int li, int lh, const f32x4 (&bias4)[4], f32x2 (&s1)[8], f32x2 (&s2)[8],
                                            const f32x4 (&rv)[4][4]) {
    const int lt = 32 * mi + li;
    const int ty = 8 * by + (lt >> 3), tx = 8 * bx + (lt & 7);
    if (ty >= (p.H >> 1) || tx >= (p.W >> 1)) return;
    float* o = p.out + ((size_t)(img * p.H + 2 * ty) * p.W + 2 * tx) * p.ldo + n0 + 32 * ni + 4 * lh;
    const size_t rowstride = (size_t)p.W * p.ldo;
    const float lo = p.relu ? 0.f : -__builtin_inff();        // one code path: max(y, -inf) = y  (two instantiations made the
#pragma unroll                                                 //  compiler stage all 256 accumulators through scratch)
    for (int g = 0; g < 4; ++g) {
        f32x2 y[2][2][2];                         // [out row][out col][channel pair]
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x2 m[16];
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) {
                // explicit accumulator reads: element extraction left to the compiler round-trips whole accumulators
                // through VGPRs and back (~270 extra moves per tile at ~8 cycles each)
                float e0, e1;
                asm("v_accvgpr_read_b32 %0, %1" : "=v"(e0) : "a"(acc[xi][4 * g + 2 * h]));
                asm("v_accvgpr_read_b32 %0, %1" : "=v"(e1) : "a"(acc[xi][4 * g + 2 * h + 1]));
                m[xi] = f32x2{e0, e1};
            }
            const f32x2 b2 = h ? f32x2{bias4[g][2], bias4[g][3]} : f32x2{bias4[g][0], bias4[g][1]};
            f32x2 rr[2][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x2 s48 = m[4 + c] + m[8 + c], d48 = wf_pk_sub(m[4 + c], m[8 + c]);
                rr[0][c] = m[0 + c] + s48;
                rr[1][c] = wf_pk_sub(d48, m[12 + c]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const f32x2 s12 = rr[i][1] + rr[i][2], d12 = wf_pk_sub(rr[i][1], rr[i][2]);
                f32x2 y0 = (rr[i][0] + b2) + s12, y1 = wf_pk_sub(d12 + b2, rr[i][3]);
                y[i][0][h] = f32x2{fmaxf(y0.x, lo), fmaxf(y0.y, lo)}; y[i][1][h] = f32x2{fmaxf(y1.x, lo), fmaxf(y1.y, lo)};
            }
            if (STATS == 1) {
                s1[2 * g + h] += (y[0][0][h] + y[0][1][h]) + (y[1][0][h] + y[1][1][h]);
                s2[2 * g + h] += (y[0][0][h] * y[0][0][h] + y[0][1][h] * y[0][1][h]) + (y[1][0][h] * y[1][0][h] + y[1][1][h] * y[1][1][h]);
            }
            if (STATS == 2) {
                s1[2 * g + h] += (y[0][0][h] + y[0][1][h]) + (y[1][0][h] + y[1][1][h]);
                f32x2 q = f32x2{0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) q += y[i][j][h] * (h ? f32x2{rv[2 * i + j][g][2], rv[2 * i + j][g][3]} : f32x2{rv[2 * i + j][g][0], rv[2 * i + j][g][1]});
                s2[2 * g + h] += q;
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                *reinterpret_cast<f32x4*>(o + i * rowstride + (size_t)j * p.ldo + 8 * g) = f32x4{y[i][j][0].x, y[i][j][0].y, y[i][j][1].x, y[i][j][1].y};
        __builtin_amdgcn_sched_barrier(0);        // one channel quad at a time: hoisting all 256 accumulator reads costs spills
    }
}
// STATS == 2: the lane's 4 pixels x 16 channels of the producer layer's saved activation (zero for lanes outside the image or
// for channel tiles outside [bn_c0, bn_c1), whose sums are never read)
__device__ __forceinline__ void wf_load_r(const WinoFusedArgs& p, int img, int by, int bx, int n0, int mi, int ni, int li, int lh,
                                          f32x4 (&rv)[4][4]) {
    const int lt = 32 * mi + li;
    const int ty = 8 * by + (lt >> 3), tx = 8 * bx + (lt & 7);
    const bool ok = ty < (p.H >> 1) && tx < (p.W >> 1) && n0 >= p.bn_c0 && n0 < p.bn_c1;
    const float* r = p.bn_r + ((size_t)(img * p.H + 2 * ty) * p.W + 2 * tx) * p.bn_ldr + (n0 - p.bn_c0) + 32 * ni + 4 * lh;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                rv[2 * i + j][g] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (ok) rv[2 * i + j][g] = *reinterpret_cast<const f32x4*>(r + ((size_t)i * p.W + j) * p.bn_ldr + 8 * g);
            }
}

// Per-lane running sums -> one row of partials per wave.  A persistent workgroup only ever sees ONE 64-channel output tile
// (tile ids advance by gridDim.x, a multiple of nt), so the sums run over all its tiles and are reduced across the 32 lanes
// of a half-wave once, at the end.  Layout: stat_part[tn][row][64 channels][2], row = 2 * (first tile / nt) + mi.
__device__ __forceinline__ void wf_write_stats(const WinoFusedArgs& p, int t0, int rows_per_tn, int mi, int ni, int li, int lh,
                                               f32x2 (&s1)[8], f32x2 (&s2)[8]) {
    float v[32];
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[4 * i] = s1[i].x; v[4 * i + 1] = s1[i].y; v[4 * i + 2] = s2[i].x; v[4 * i + 3] = s2[i].y; }
#pragma unroll
    for (int i = 0; i < 32; ++i)
#pragma unroll
        for (int m = 1; m < 32; m <<= 1) v[i] += __shfl_xor(v[i], m, 32);
    if (li != 0) return;
    const int tn = t0 % p.nt, row = 2 * (t0 / p.nt) + mi;
    float* o = p.stat_part + ((size_t)tn * rows_per_tn + row) * 128;
#pragma unroll
    for (int i = 0; i < 8; ++i) {                       // pair i = 2g + h -> channels 32*ni + 8g + 4lh + 2h + {0,1}
        const int ch = 32 * ni + 8 * (i >> 1) + 4 * lh + 2 * (i & 1);
        o[2 * ch] = v[4 * i]; o[2 * ch + 1] = v[4 * i + 2]; o[2 * ch + 2] = v[4 * i + 1]; o[2 * ch + 3] = v[4 * i + 3];
    }
}

int wino_stream_cus() {
    static const int cus = [] { int d = 0, n = 0; if (hipGetDevice(&d) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || n <= 0) n = 256; return n; }();
    return cus;
}
// rows of statistics partials per 64-channel tile the persistent kernel would write (0: shape not taken by it / grid not a
// multiple of the n-tile count)
int wino_stats_rows(int N, int H, int W, int K, int Nout, int max_workgroups = 0) {
    if (!(K % 16 == 0 && K >= 32 && H % 2 == 0 && W % 2 == 0 && Nout % 64 == 0)) return 0;
    const int nt = Nout / 64;
    const long blocks = (long)N * ((H / 2 + 7) / 8) * ((W / 2 + 7) / 8) * nt;
    const long slots = unet_grid_slots(wino_stream_cus(), max_workgroups);
    const long grid = blocks < slots ? blocks : slots;
    return grid % nt == 0 ? (int)(2 * (grid / nt)) : 0;
}

struct WinoBnBwd { const float* r; int ldr, c0, c1; };

}  // namespace
