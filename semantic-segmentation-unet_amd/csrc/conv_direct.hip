// HBM-bound convolutions that are not GEMM-shaped (SURVEY.md 8(d) "exceptions"):
//   * the first 3x3 layer, Cin = number_channels (1..few)  -> 64   (UNet/model.py:88): K = 9*Cin is far too short
//     for the matrix cores; one output pixel is 256 B written for 9*Cin*64 FMAs -> VALU stencil, float4 stores;
//   * the 1x1 class-map layer 64 -> number_classes               (UNet/model.py:136): 1-3 flop/B.
// Thread mapping everywhere: consecutive lanes own consecutive channel quads (float4) of one pixel so global
// accesses are whole 64..256-B pixel rows.
#include "common.h"

namespace {

constexpr int CI_CHUNK = 8;

// channel quad q of a pixel row whose elements are fp32 (16 B) or, B16, bf16 (8 B; `base` then points at 2-byte elements): the bf16
// activation-storage mode keeps the first layer's conv output and the class-map layer's input / input gradient as bf16 tensors
template <int B16> __device__ __forceinline__ f32x4 ld_quad(const float* base, size_t elem) {
    if constexpr (B16) {
        const uint2 h = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + elem);
        return f32x4{__builtin_bit_cast(float, h.x << 16), __builtin_bit_cast(float, h.x & 0xffff0000u),
                     __builtin_bit_cast(float, h.y << 16), __builtin_bit_cast(float, h.y & 0xffff0000u)};
    } else return *reinterpret_cast<const f32x4*>(base + elem);
}
template <int B16> __device__ __forceinline__ void st_quad(float* base, size_t elem, f32x4 v) {
    if constexpr (B16) {
        uint2 h;
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h.x) : "v"(v[0]), "v"(v[1]));
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h.y) : "v"(v[2]), "v"(v[3]));
        *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(base) + elem) = h;
    } else *reinterpret_cast<f32x4*>(base + elem) = v;
}

// ---- 3x3 'same' conv, small Cin, forward ---------------------------------------------------------------------
// Cin = 1..4 (the first layer): the 9*Cin weight quads live in registers and a thread walks a strip of 4 consecutive
// pixels of one image row, so the 3x6 input window is loaded once per strip and each output quad costs 9*Cin FMAs x 4 plus
// its store (the generic kernel below spends ~100 instructions per output quad on tap addressing and LDS weight reads,
// which capped it at 1.5 TB/s of the 6.3 the output stream could take).
template <int CIN, int OUT16>
__global__ __launch_bounds__(256) void conv3x3_direct_fwd_strip_kernel(const float* __restrict__ x, int ldx,
        const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ out, int ldo,
        int N, int H, int W, int Cout, int relu, float* __restrict__ stat_part) {
    extern __shared__ __attribute__((aligned(16))) float sStat[];  // [spb][Cout][2] when stat_part != null
    const int tpp = Cout >> 2, spb = 256 / tpp;                   // strips per block pass
    const int q = threadIdx.x % tpp, sl = threadIdx.x / tpp;
    f32x4 wr[9][CIN];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) wr[t][ci] = *reinterpret_cast<const f32x4*>(w + ((size_t)t * CIN + ci) * Cout + 4 * q);
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias) bv = *reinterpret_cast<const f32x4*>(bias + 4 * q);
    const int SW = W >> 2;                                        // strips per row (W % 4 == 0)
    const long strips = (long)N * H * SW;
    const float lo = relu ? 0.f : -__builtin_inff();
    f32x4 st1 = {0.f, 0.f, 0.f, 0.f}, st2 = {0.f, 0.f, 0.f, 0.f};     // BatchNorm sums of the thread's channel quad (UNet/model.py:36)
    for (long s = (long)blockIdx.x * spb + sl; s < strips; s += (long)gridDim.x * spb) {
        long t = s; const int sx = (int)(t % SW); t /= SW; const int y = (int)(t % H); const int n = (int)(t / H);
        const int x0 = 4 * sx;
        float v[3][6][CIN];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int gy = y + r - 1;
            const bool rok = (unsigned)gy < (unsigned)H;
            const float* row = x + ((size_t)(n * H + (rok ? gy : y)) * W) * ldx;
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const int gx = x0 + c - 1;
                const bool ok = rok && (unsigned)gx < (unsigned)W;
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) v[r][c][ci] = ok ? row[(size_t)gx * ldx + ci] : 0.f;
            }
        }
        const size_t o = ((size_t)(n * H + y) * W + x0) * ldo + 4 * q;
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            f32x4 acc = bv;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int ci = 0; ci < CIN; ++ci) acc += v[r][px + c][ci] * wr[3 * r + c][ci];
            acc[0] = fmaxf(acc[0], lo); acc[1] = fmaxf(acc[1], lo); acc[2] = fmaxf(acc[2], lo); acc[3] = fmaxf(acc[3], lo);
            st_quad<OUT16>(out, o + (size_t)px * ldo, acc);
            st1 += acc; st2 += acc * acc;                         // (sums of the fp32 values, before any rounding of the stored tensor)
        }
    }
    if (stat_part) {            // one row of partials per block, layout of unet_bn_train_finalize_partials: [C/64][rows][64][2]
#pragma unroll
        for (int e = 0; e < 4; ++e) { sStat[(sl * Cout + 4 * q + e) * 2] = st1[e]; sStat[(sl * Cout + 4 * q + e) * 2 + 1] = st2[e]; }
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * Cout; i += 256) {
            float t = 0.f;
            for (int l = 0; l < spb; ++l) t += sStat[l * 2 * Cout + i];
            const int c = i >> 1;
            stat_part[((size_t)(c >> 6) * gridDim.x + blockIdx.x) * 128 + (c & 63) * 2 + (i & 1)] = t;
        }
    }
}

__global__ __launch_bounds__(256) void conv3x3_direct_fwd_kernel(const float* __restrict__ x, int ldx,
        const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ out, int ldo,
        int N, int H, int W, int Cin, int Cout, int relu) {
    extern __shared__ __attribute__((aligned(16))) float sW[];        // [9][cc][Cout]
    const int tpp = Cout >> 2, ppb = 256 / tpp;
    const int q = threadIdx.x % tpp, pl = threadIdx.x / tpp;
    const long P = (long)N * H * W;
    const long groups = (P + ppb - 1) / ppb;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias) bv = *reinterpret_cast<const f32x4*>(bias + 4 * q);
    if (Cin <= CI_CHUNK) {
        // common case (first layer): weights staged once per block, blocks stride over pixel groups
        for (int i = threadIdx.x; i < 9 * Cin * Cout; i += 256) sW[i] = w[i];       // [9][Cin][Cout] is already the layout
        __syncthreads();
        for (long grp = blockIdx.x; grp < groups; grp += gridDim.x) {
            const long pix = grp * ppb + pl;
            if (pix >= P) continue;
            long t = pix; const int xx = (int)(t % W); t /= W; const int y = (int)(t % H); const int n = (int)(t / H);
            f32x4 acc = bv;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int gy = y + tap / 3 - 1, gx = xx + tap % 3 - 1;
                if (gy < 0 || gy >= H || gx < 0 || gx >= W) continue;
                const float* xp = x + ((size_t)(n * H + gy) * W + gx) * ldx;
                for (int ci = 0; ci < Cin; ++ci)
                    acc += xp[ci] * *reinterpret_cast<const f32x4*>(sW + (tap * Cin + ci) * Cout + 4 * q);
            }
            if (relu) { acc[0] = fmaxf(acc[0], 0.f); acc[1] = fmaxf(acc[1], 0.f); acc[2] = fmaxf(acc[2], 0.f); acc[3] = fmaxf(acc[3], 0.f); }
            *reinterpret_cast<f32x4*>(out + (size_t)pix * ldo + 4 * q) = acc;
        }
        return;
    }
    // general Cin: chunks of CI_CHUNK input channels through LDS, one pixel group per block pass
    for (long grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const long pix = grp * ppb + pl;
        const bool live = pix < P;
        int n = 0, y = 0, xx = 0;
        if (live) { long t = pix; xx = (int)(t % W); t /= W; y = (int)(t % H); n = (int)(t / H); }
        f32x4 acc = bv;
        for (int c0 = 0; c0 < Cin; c0 += CI_CHUNK) {
            const int cc = min(CI_CHUNK, Cin - c0);
            __syncthreads();
            for (int i = threadIdx.x; i < 9 * cc * Cout; i += 256) {
                const int co = i % Cout, r = i / Cout, ci = r % cc, tap = r / cc;
                sW[i] = w[((size_t)tap * Cin + c0 + ci) * Cout + co];
            }
            __syncthreads();
            if (live) {
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int gy = y + tap / 3 - 1, gx = xx + tap % 3 - 1;
                    if (gy < 0 || gy >= H || gx < 0 || gx >= W) continue;
                    const float* xp = x + ((size_t)(n * H + gy) * W + gx) * ldx + c0;
                    for (int ci = 0; ci < cc; ++ci)
                        acc += xp[ci] * *reinterpret_cast<const f32x4*>(sW + (tap * cc + ci) * Cout + 4 * q);
                }
            }
        }
        if (live) {
            if (relu) { acc[0] = fmaxf(acc[0], 0.f); acc[1] = fmaxf(acc[1], 0.f); acc[2] = fmaxf(acc[2], 0.f); acc[3] = fmaxf(acc[3], 0.f); }
            *reinterpret_cast<f32x4*>(out + (size_t)pix * ldo + 4 * q) = acc;
        }
    }
}

// ---- 3x3 conv, small Cin, weight gradient: blockIdx.y = ci, per-block partials -> fixed-order reduce ----------
template <int Z16>
__global__ __launch_bounds__(256) void conv3x3_direct_wgrad_kernel(const float* __restrict__ x, int ldx,
        const float* __restrict__ dz, int lddz, float* __restrict__ part, int N, int H, int W, int Cin, int Cout,
        long pix_per_block) {
    extern __shared__ __attribute__((aligned(16))) float sR[];        // [pl][9][Cout]
    const int tpp = Cout >> 2, npl = 256 / tpp;
    const int q = threadIdx.x % tpp, pl = threadIdx.x / tpp;
    const int ci = blockIdx.y;
    const long P = (long)N * H * W;
    const long p0 = (long)blockIdx.x * pix_per_block;
    long p1 = p0 + pix_per_block; if (p1 > P) p1 = P;
    f32x4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if ((W & 3) == 0 && (p0 & 3) == 0 && ((p1 - p0) & 3) == 0) {
        // strips of 4 consecutive pixels of a row: the 3x6 input window is loaded once per strip (the per-pixel form below
        // spends most of its instructions on tap addressing: 2.2 TB/s of dz instead of ~4.5)
        for (long pix = p0 + 4 * pl; pix < p1; pix += 4 * npl) {
            long t = pix; const int x0 = (int)(t % W); t /= W; const int y = (int)(t % H); const int n = (int)(t / H);
            f32x4 g[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) g[u] = ld_quad<Z16>(dz, (size_t)(pix + u) * lddz + 4 * q);
            float v[3][6];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int gy = y + r - 1;
                const bool rok = (unsigned)gy < (unsigned)H;
                const float* row = x + ((size_t)(n * H + (rok ? gy : y)) * W) * ldx + ci;
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    const int gx = x0 + c - 1;
                    v[r][c] = (rok && (unsigned)gx < (unsigned)W) ? row[(size_t)gx * ldx] : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) acc[tap] += v[tap / 3][u + tap % 3] * g[u];
        }
    } else {
        for (long pix = p0 + pl; pix < p1; pix += npl) {
            long t = pix; const int xx = (int)(t % W); t /= W; const int y = (int)(t % H); const int n = (int)(t / H);
            const f32x4 g = ld_quad<Z16>(dz, (size_t)pix * lddz + 4 * q);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int gy = y + tap / 3 - 1, gx = xx + tap % 3 - 1;
                float xv = 0.f;
                if (gy >= 0 && gy < H && gx >= 0 && gx < W) xv = x[((size_t)(n * H + gy) * W + gx) * ldx + ci];
                acc[tap] += xv * g;
            }
        }
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) *reinterpret_cast<f32x4*>(sR + ((size_t)pl * 9 + tap) * Cout + 4 * q) = acc[tap];
    __syncthreads();
    // part[blk][tap][ci][co]
    for (int i = threadIdx.x; i < 9 * Cout; i += 256) {
        const int tap = i / Cout, co = i % Cout;
        float s = 0.f;
        for (int l = 0; l < npl; ++l) s += sR[((size_t)l * 9 + tap) * Cout + co];
        part[(((size_t)blockIdx.x * 9 + tap) * Cin + ci) * Cout + co] = s;
    }
}

// the same for 2..4 input channels in ONE pass over dz (the per-channel grid above reads dz once per input channel): a thread keeps
// 9 x CIN accumulator quads; requires the 4-pixel strip geometry (W % 4 == 0, block ranges multiples of 4)
template <int CIN, int Z16>
__global__ __launch_bounds__(256) void conv3x3_direct_wgrad_multi_kernel(const float* __restrict__ x, int ldx,
        const float* __restrict__ dz, int lddz, float* __restrict__ part, int N, int H, int W, int Cout, long pix_per_block) {
    extern __shared__ __attribute__((aligned(16))) float sR[];        // [pl][9][Cout]
    const int tpp = Cout >> 2, npl = 256 / tpp;
    const int q = threadIdx.x % tpp, pl = threadIdx.x / tpp;
    const long P = (long)N * H * W;
    const long p0 = (long)blockIdx.x * pix_per_block;
    long p1 = p0 + pix_per_block; if (p1 > P) p1 = P;
    f32x4 acc[CIN][9];
#pragma unroll
    for (int c = 0; c < CIN; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (long pix = p0 + 4 * pl; pix < p1; pix += 4 * npl) {
        long t = pix; const int x0 = (int)(t % W); t /= W; const int y = (int)(t % H); const int n = (int)(t / H);
        f32x4 g[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) g[u] = ld_quad<Z16>(dz, (size_t)(pix + u) * lddz + 4 * q);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int gy = y + r - 1;
            const bool rok = (unsigned)gy < (unsigned)H;
            const float* row = x + ((size_t)(n * H + (rok ? gy : y)) * W) * ldx;
            float v[6][CIN];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const int gx = x0 + c - 1;
                const bool ok = rok && (unsigned)gx < (unsigned)W;
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) v[c][ci] = ok ? row[(size_t)gx * ldx + ci] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int b = 0; b < 3; ++b)
#pragma unroll
                    for (int ci = 0; ci < CIN; ++ci) acc[ci][3 * r + b] += v[u + b][ci] * g[u];
        }
    }
    // part[blk][tap][ci][co], one input channel at a time through the same LDS tree as the per-channel kernel
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) {
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) *reinterpret_cast<f32x4*>(sR + ((size_t)pl * 9 + tap) * Cout + 4 * q) = acc[ci][tap];
        __syncthreads();
        for (int i = threadIdx.x; i < 9 * Cout; i += 256) {
            const int tap = i / Cout, co = i % Cout;
            float s = 0.f;
            for (int l = 0; l < npl; ++l) s += sR[((size_t)l * 9 + tap) * Cout + co];
            part[(((size_t)blockIdx.x * 9 + tap) * CIN + ci) * Cout + co] = s;
        }
    }
}

// one wave per output element: lanes stride over the per-block partials, fixed shuffle tree (deterministic)
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ part, float* __restrict__ out, long n, int nparts) {
    const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int lane = threadIdx.x & 63;
    double s = 0.0;
    for (int k = lane; k < nparts; k += 64) s += (double)part[(size_t)k * n + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[i] = (float)s;
}

// ---- 1x1 conv, small Cout (class map) -------------------------------------------------------------------------
// forward: 16 lanes per pixel, each lane takes channel quads sub, sub+16, ...; partial dots reduced by shuffles.
template <int X16>
__global__ __launch_bounds__(256) void conv1x1_narrow_fwd_kernel(const float* __restrict__ x, int ldx,
        const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ out, int ldo,
        long P, int Cin, int K, int relu) {
    extern __shared__ __attribute__((aligned(16))) float sW[];        // [Cin][K]
    for (int i = threadIdx.x; i < Cin * K; i += 256) sW[i] = w[i];
    __syncthreads();
    const int sub = threadIdx.x & 15;
    const int nq = Cin >> 2;
    constexpr int PU = 4;                                             // pixels per 16-lane group per pass
    const long pstride = (long)gridDim.x * 16 * PU;
    // the 16 lanes of a group share `base`, so a shuffle group is always entirely active or entirely exited
    for (long base = ((long)blockIdx.x * 16 + (threadIdx.x >> 4)) * PU; base < P; base += pstride) {
        for (int k0 = 0; k0 < K; k0 += 4) {
            float acc[PU][4];
#pragma unroll
            for (int u = 0; u < PU; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[u][j] = 0.f;
            for (int cq = sub; cq < nq; cq += 16) {
                f32x4 xv[PU];
#pragma unroll
                for (int u = 0; u < PU; ++u) {
                    const long pix = base + u < P ? base + u : P - 1;
                    xv[u] = ld_quad<X16>(x, (size_t)pix * ldx + 4 * cq);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float wv = (k0 + j < K) ? sW[(4 * cq + e) * K + k0 + j] : 0.f;
#pragma unroll
                        for (int u = 0; u < PU; ++u) acc[u][j] += xv[u][e] * wv;
                    }
            }
#pragma unroll
            for (int u = 0; u < PU; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = acc[u][j];
                    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
                    acc[u][j] = v;
                }
            // lane sub = 4*u + j writes pixel u, class k0 + j
            const int u = sub >> 2, j = sub & 3;
            float v = 0.f;
#pragma unroll
            for (int uu = 0; uu < PU; ++uu)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) if (uu == u && jj == j) v = acc[uu][jj];
            if (base + u < P && k0 + j < K) {
                v += bias ? bias[k0 + j] : 0.f;
                if (relu) v = fmaxf(v, 0.f);
                out[(size_t)(base + u) * ldo + k0 + j] = v;
            }
        }
    }
}

// Forward for the shape the network has (Cin = 64 .. 512, K <= 8): Cin / 8 lanes per pixel, a lane owns 8 channels (one 16-byte load of a
// bf16 tensor, two of an fp32 one) with its 8 x K weights in registers, the K partial dots are reduced over the pixel's lanes by xor
// shuffles and lane k stores class k.  (The 16-lanes-per-pixel kernel above reads the weights from LDS per element and spends most of
// its instructions on 16-value shuffle trees: 146 us for 8 x 512^2 x 64 -> 4, three times the time its 268 MB take on HBM.)
template <int X16, int LPP>          // LPP = lanes per pixel = Cin / 8 (8, 16, 32 or 64)
__global__ __launch_bounds__(256) void conv1x1_narrow_fwd8_kernel(const float* __restrict__ x, int ldx,
        const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ out, int ldo, long P, int K, int relu) {
    const int sub = threadIdx.x % LPP;                               // this lane's channel octet
    float wr[8][8];
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int k = 0; k < 8; ++k) wr[e][k] = k < K ? w[(8 * sub + e) * K + k] : 0.f;
    const float bk = (bias && sub < K) ? bias[sub] : 0.f;
    constexpr int PPB = 256 / LPP;                                   // pixels per block pass
    const long stride = (long)gridDim.x * PPB;
    for (long pix0 = (long)blockIdx.x * PPB + threadIdx.x / LPP; pix0 < P + 0; pix0 += 2 * stride) {     // two pixels per trip (both loads first)
        const long pa = pix0, pb = pix0 + stride < P ? pix0 + stride : pix0;
        float xa[8], xb[8];
        if constexpr (X16) {
            const uint4 ha = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(x) + (size_t)pa * ldx + 8 * sub);
            const uint4 hb = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(x) + (size_t)pb * ldx + 8 * sub);
            const unsigned ua[4] = {ha.x, ha.y, ha.z, ha.w}, ub[4] = {hb.x, hb.y, hb.z, hb.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xa[2 * j] = __builtin_bit_cast(float, ua[j] << 16); xa[2 * j + 1] = __builtin_bit_cast(float, ua[j] & 0xffff0000u);
                xb[2 * j] = __builtin_bit_cast(float, ub[j] << 16); xb[2 * j + 1] = __builtin_bit_cast(float, ub[j] & 0xffff0000u);
            }
        } else {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(x + (size_t)pa * ldx + 8 * sub), a1 = *reinterpret_cast<const f32x4*>(x + (size_t)pa * ldx + 8 * sub + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(x + (size_t)pb * ldx + 8 * sub), b1 = *reinterpret_cast<const f32x4*>(x + (size_t)pb * ldx + 8 * sub + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { xa[j] = a0[j]; xa[4 + j] = a1[j]; xb[j] = b0[j]; xb[4 + j] = b1[j]; }
        }
        float sa[8], sb[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            sa[k] = 0.f; sb[k] = 0.f;
            if (k < K) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { sa[k] = fmaf(xa[e], wr[e][k], sa[k]); sb[k] = fmaf(xb[e], wr[e][k], sb[k]); }
#pragma unroll
                for (int o = 1; o < LPP; o <<= 1) { sa[k] += __shfl_xor(sa[k], o); sb[k] += __shfl_xor(sb[k], o); }
            }
        }
        float va = 0.f, vb = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) if (k == sub) { va = sa[k]; vb = sb[k]; }
        if (sub < K) {
            va += bk; vb += bk;
            if (relu) { va = fmaxf(va, 0.f); vb = fmaxf(vb, 0.f); }
            out[(size_t)pa * ldo + sub] = va;
            if (pix0 + stride < P) out[(size_t)pb * ldo + sub] = vb;
        }
    }
}

// dgrad: dx[p][ci quad] = sum_k dz[p][k] * w[ci][k]
template <int DX16>
__global__ __launch_bounds__(256) void conv1x1_narrow_dgrad_kernel(const float* __restrict__ dz, int lddz,
        const float* __restrict__ w, float* __restrict__ dx, int lddx, long P, int Cin, int K) {
    extern __shared__ __attribute__((aligned(16))) float sW[];        // [Cin][K]
    for (int i = threadIdx.x; i < Cin * K; i += 256) sW[i] = w[i];
    __syncthreads();
    const int nq = Cin >> 2;
    const long total = P * nq;
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (K <= 8 && stride % nq == 0 && i < total) {
        // the grid stride is a multiple of the quads per pixel: a thread keeps its channel quad, so its 4 x K weights live in
        // registers (the LDS reads of sW, 16 per element at K = 4, were what bounded this kernel, not HBM)
        const int cq = (int)(i % nq);
        float wr[4][8];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int k = 0; k < 8; ++k) wr[e][k] = k < K ? sW[(4 * cq + e) * K + k] : 0.f;
        for (; i < total; i += stride) {
            const long pix = i / nq;
            float g[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) g[k] = k < K ? dz[(size_t)pix * lddz + k] : 0.f;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k < K) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] += g[k] * wr[e][k];
                }
            st_quad<DX16>(dx, (size_t)pix * lddx + 4 * cq, acc);
        }
        return;
    }
    for (; i < total; i += stride) {
        const long pix = i / nq; const int cq = (int)(i - pix * nq);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < K; ++k) {
            const float g = dz[(size_t)pix * lddz + k];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += g * sW[(4 * cq + e) * K + k];
        }
        st_quad<DX16>(dx, (size_t)pix * lddx + 4 * cq, acc);
    }
}

// wgrad: part[blk][ci][k] = sum over the block's pixels of x[p][ci] * dz[p][k]   (k0..k0+8 per pass)
template <int X16>
__global__ __launch_bounds__(256) void conv1x1_narrow_wgrad_kernel(const float* __restrict__ x, int ldx,
        const float* __restrict__ dz, int lddz, float* __restrict__ part, long P, int Cin, int K, long pix_per_block) {
    extern __shared__ __attribute__((aligned(16))) float sR[];        // [npl][4*tpp][8]
    const int nq = Cin >> 2;
    const int tpp = nq < 256 ? nq : 256, npl = 256 / tpp;
    const int q = threadIdx.x % tpp, pl = threadIdx.x / tpp;
    const long p0 = (long)blockIdx.x * pix_per_block;
    long p1 = p0 + pix_per_block; if (p1 > P) p1 = P;
    for (int qq = q; qq < nq; qq += tpp) {
        for (int k0 = 0; k0 < K; k0 += 8) {
            float acc[4][8];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[e][j] = 0.f;
            long pix = p0 + pl;
            // four pixels per trip, all their loads issued before the first use (one pixel per trip left the kernel latency-bound
            // at ~1 TB/s; the summation order per accumulator is unchanged)
            for (; pix + 3 * (long)npl < p1; pix += 4 * (long)npl) {
                f32x4 xv[4]; float g[4][8];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    xv[u] = ld_quad<X16>(x, (size_t)(pix + u * (long)npl) * ldx + 4 * qq);
#pragma unroll
                    for (int j = 0; j < 8; ++j) g[u][j] = (k0 + j < K) ? dz[(size_t)(pix + u * (long)npl) * lddz + k0 + j] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (k0 + j < K) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[e][j] += xv[u][e] * g[u][j];
                        }
            }
            for (; pix < p1; pix += npl) {
                const f32x4 xv = ld_quad<X16>(x, (size_t)pix * ldx + 4 * qq);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (k0 + j < K) {
                        const float g = dz[(size_t)pix * lddz + k0 + j];
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[e][j] += xv[e] * g;
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < 8; ++j) sR[((size_t)pl * 4 * tpp + 4 * q + e) * 8 + j] = acc[e][j];
            __syncthreads();
            for (int i = threadIdx.x; i < 4 * tpp * 8; i += 256) {
                const int j = i & 7, c = i >> 3;          // c in [0, 4*tpp): channel 4*q'+e of this qq pass
                float s = 0.f;
                for (int l = 0; l < npl; ++l) s += sR[((size_t)l * 4 * tpp + c) * 8 + j];
                const int ci = 4 * (qq - q) + c;          // qq - q is the pass base (multiple of tpp)
                if (k0 + j < K && ci < Cin) part[((size_t)blockIdx.x * Cin + ci) * K + k0 + j] = s;
            }
        }
    }
}

}  // namespace

static int direct_fwd_launch(const float* x, int ldx, const float* w, const float* bias, float* out, int ldo,
                             int N, int H, int W, int Cin, int Cout, int relu, float* stat_part, size_t stat_bytes, void* stream, int out_bf16 = 0) {
    UNET_CHECK_ARG(x && w && out && N > 0 && H > 0 && W > 0 && Cin > 0 && ldx >= Cin && ldo >= Cout);
    const int tpp = Cout / 4;
    UNET_CHECK_ARG(Cout % 4 == 0 && tpp >= 1 && tpp <= 256 && 256 % tpp == 0 && ldo % 4 == 0 && unet_aligned16(out));
    UNET_CHECK_ARG(!bias || unet_aligned16(bias));
    const long P = (long)N * H * W;
    const int ppb = 256 / tpp;
    const size_t smem = (size_t)9 * CI_CHUNK * Cout * sizeof(float);
    UNET_CHECK_ARG(smem <= 64 * 1024);
    long blocks = (P + ppb - 1) / ppb; if (blocks > 4096) blocks = 4096;
    if (Cin <= 4 && W % 4 == 0 && unet_aligned16(w)) {
        long b1 = (P / 4 + ppb - 1) / ppb; if (b1 > 4096) b1 = 4096;
        hipStream_t st = (hipStream_t)stream;
        size_t sm = 0;
        if (stat_part) {
            UNET_CHECK_ARG(Cout % 64 == 0);
            if (stat_bytes < (size_t)(Cout / 64) * b1 * 128 * sizeof(float)) return UNET_ENOSPC;
            sm = (size_t)ppb * Cout * 2 * sizeof(float);
        }
#define UNET_STRIP(CI) do { if (out_bf16) conv3x3_direct_fwd_strip_kernel<CI, 1><<<(int)b1, 256, sm, st>>>(x, ldx, w, bias, out, ldo, N, H, W, Cout, relu, stat_part); \
                            else          conv3x3_direct_fwd_strip_kernel<CI, 0><<<(int)b1, 256, sm, st>>>(x, ldx, w, bias, out, ldo, N, H, W, Cout, relu, stat_part); } while (0)
        if (Cin == 1) UNET_STRIP(1); else if (Cin == 2) UNET_STRIP(2); else if (Cin == 3) UNET_STRIP(3); else UNET_STRIP(4);
#undef UNET_STRIP
        return UNET_LAUNCH_STATUS();
    }
    if (stat_part || out_bf16) return UNET_EINVAL;
    conv3x3_direct_fwd_kernel<<<(int)blocks, 256, smem, (hipStream_t)stream>>>(x, ldx, w, bias, out, ldo, N, H, W, Cin, Cout, relu);
    return UNET_LAUNCH_STATUS();
}

extern "C" int unet_conv3x3_fwd_direct(const float* x, int ldx, const float* w, const float* bias, float* out, int ldo,
                                       int N, int H, int W, int Cin, int Cout, int relu, void* stream) {
    return direct_fwd_launch(x, ldx, w, bias, out, ldo, N, H, W, Cin, Cout, relu, nullptr, 0, stream);
}

// rows of BatchNorm partial sums the strip kernel would write per 64-channel block (0: the generic kernel would run)
extern "C" int unet_conv3x3_fwd_direct_stats_rows(int N, int H, int W, int Cin, int Cout) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin < 1 || Cin > 4 || W % 4 != 0 || Cout % 64 != 0 || Cout > 1024) return 0;
    const int ppb = 256 / (Cout / 4);
    long b1 = ((long)N * H * W / 4 + ppb - 1) / ppb; if (b1 > 4096) b1 = 4096;
    return (int)b1;
}

// forward + BatchNorm sums of the output (stat_part: (Cout/64) * rows * 128 floats; finish with unet_bn_train_finalize_partials);
// out_bf16: the output tensor is stored as bf16 (ldo in elements; the sums are those of the fp32 values)
extern "C" int unet_conv3x3_fwd_direct_stats(const float* x, int ldx, const float* w, const float* bias, void* out, int ldo, int out_bf16,
        int N, int H, int W, int Cin, int Cout, int relu, float* stat_part, size_t stat_bytes, void* stream) {
    UNET_CHECK_ARG(stat_part && unet_conv3x3_fwd_direct_stats_rows(N, H, W, Cin, Cout) > 0 && unet_aligned16(w));
    return direct_fwd_launch(x, ldx, w, bias, (float*)out, ldo, N, H, W, Cin, Cout, relu, stat_part, stat_bytes, stream, out_bf16 ? 1 : 0);
}

static int direct_wgrad_blocks(long P) { long b = (P + 1023) / 1024; if (b > 1024) b = 1024; if (b < 1) b = 1; return (int)b; }

extern "C" size_t unet_conv3x3_wgrad_direct_workspace(int N, int H, int W, int Cin, int Cout) {
    return (size_t)direct_wgrad_blocks((long)N * H * W) * 9 * Cin * Cout * sizeof(float);
}

// dz_bf16: dz is stored as bf16 (lddz in elements); products and sums in fp32
extern "C" int unet_conv3x3_wgrad_direct(const float* xin, int ldx, const void* dzv, int lddz, int dz_bf16, float* dw,
                                         int N, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream) {
    const float* dz = (const float*)dzv;
    UNET_CHECK_ARG(xin && dz && dw && ws && N > 0 && H > 0 && W > 0 && Cin > 0 && Cin <= 65535);
    const int tpp = Cout / 4;
    UNET_CHECK_ARG(Cout % 4 == 0 && tpp >= 1 && tpp <= 256 && 256 % tpp == 0 && lddz % 4 == 0 && unet_aligned16(dz));
    const long P = (long)N * H * W;
    const int blocks = direct_wgrad_blocks(P);
    if (ws_bytes < unet_conv3x3_wgrad_direct_workspace(N, H, W, Cin, Cout)) return UNET_ENOSPC;
    const long ppb = ((P + blocks - 1) / blocks + 3) & ~3L;         // multiple of 4: a block's range is whole 4-pixel strips
    const size_t smem = (size_t)(256 / tpp) * 9 * Cout * sizeof(float);
    UNET_CHECK_ARG(smem <= 64 * 1024);
    hipStream_t st = (hipStream_t)stream;
    if (Cin >= 2 && Cin <= 4 && W % 4 == 0) {                       // every block range is whole 4-pixel strips (ppb % 4 == 0, P % 4 == 0)
#define UNET_WG(CI) do { if (dz_bf16) conv3x3_direct_wgrad_multi_kernel<CI, 1><<<blocks, 256, smem, st>>>(xin, ldx, dz, lddz, (float*)ws, N, H, W, Cout, ppb); \
                         else         conv3x3_direct_wgrad_multi_kernel<CI, 0><<<blocks, 256, smem, st>>>(xin, ldx, dz, lddz, (float*)ws, N, H, W, Cout, ppb); } while (0)
        if (Cin == 2) UNET_WG(2); else if (Cin == 3) UNET_WG(3); else UNET_WG(4);
#undef UNET_WG
    } else if (dz_bf16)
        conv3x3_direct_wgrad_kernel<1><<<dim3(blocks, Cin), 256, smem, st>>>(xin, ldx, dz, lddz, (float*)ws, N, H, W, Cin, Cout, ppb);
    else
        conv3x3_direct_wgrad_kernel<0><<<dim3(blocks, Cin), 256, smem, st>>>(xin, ldx, dz, lddz, (float*)ws, N, H, W, Cin, Cout, ppb);
    int rc = UNET_LAUNCH_STATUS(); if (rc) return rc;
    const long n = 9L * Cin * Cout;
    sum_partials_kernel<<<unet_cdiv(n, 4), 256, 0, (hipStream_t)stream>>>((const float*)ws, dw, n, blocks);
    return UNET_LAUNCH_STATUS();
}

// x_bf16 / dx_bf16 below: the 64-channel side of the class-map layer is stored as bf16 (leading dimension in elements); arithmetic fp32
extern "C" int unet_conv1x1_fwd(const void* x, int ldx, int x_bf16, const float* w, const float* bias, float* out, int ldo,
                                long P, int Cin, int Cout, int relu, void* stream) {
    UNET_CHECK_ARG(x && w && out && P > 0 && Cin > 0 && Cout > 0 && Cin % 4 == 0 && ldx % 4 == 0 && ldx >= Cin && ldo >= Cout);
    UNET_CHECK_ARG(unet_aligned16(x) && (size_t)Cin * Cout * 4 <= 64 * 1024);
    if (Cin == 64 && Cout <= 8 && ldx % 8 == 0) {            // the network's class map: 8 lanes per pixel, weights in registers
        long b8 = (P + 63) / 64; if (b8 > 8192) b8 = 8192;           // (32 pixels per block pass, two pixels per trip)
        if (x_bf16) conv1x1_narrow_fwd8_kernel<1, 8><<<(int)b8, 256, 0, (hipStream_t)stream>>>((const float*)x, ldx, w, bias, out, ldo, P, Cout, relu);
        else        conv1x1_narrow_fwd8_kernel<0, 8><<<(int)b8, 256, 0, (hipStream_t)stream>>>((const float*)x, ldx, w, bias, out, ldo, P, Cout, relu);
        return UNET_LAUNCH_STATUS();
    }
    long blocks = (P + 63) / 64; if (blocks > 4096) blocks = 4096;
    if (x_bf16) conv1x1_narrow_fwd_kernel<1><<<(int)blocks, 256, (size_t)Cin * Cout * 4, (hipStream_t)stream>>>((const float*)x, ldx, w, bias, out, ldo, P, Cin, Cout, relu);
    else        conv1x1_narrow_fwd_kernel<0><<<(int)blocks, 256, (size_t)Cin * Cout * 4, (hipStream_t)stream>>>((const float*)x, ldx, w, bias, out, ldo, P, Cin, Cout, relu);
    return UNET_LAUNCH_STATUS();
}

extern "C" int unet_conv1x1_dgrad(const float* dz, int lddz, const float* w, void* dx, int lddx, int dx_bf16,
                                  long P, int Cin, int Cout, void* stream) {
    UNET_CHECK_ARG(dz && w && dx && P > 0 && Cin > 0 && Cout > 0 && Cin % 4 == 0 && lddx % 4 == 0 && lddx >= Cin && lddz >= Cout);
    UNET_CHECK_ARG(unet_aligned16(dx) && (size_t)Cin * Cout * 4 <= 64 * 1024);
    long blocks = (P * (Cin / 4) + 255) / 256; if (blocks > 8192) blocks = 8192;
    if (dx_bf16) conv1x1_narrow_dgrad_kernel<1><<<(int)blocks, 256, (size_t)Cin * Cout * 4, (hipStream_t)stream>>>(dz, lddz, w, (float*)dx, lddx, P, Cin, Cout);
    else         conv1x1_narrow_dgrad_kernel<0><<<(int)blocks, 256, (size_t)Cin * Cout * 4, (hipStream_t)stream>>>(dz, lddz, w, (float*)dx, lddx, P, Cin, Cout);
    return UNET_LAUNCH_STATUS();
}

extern "C" size_t unet_conv1x1_wgrad_workspace(long P, int Cin, int Cout) {
    return (size_t)direct_wgrad_blocks(P) * Cin * Cout * sizeof(float);
}

extern "C" int unet_conv1x1_wgrad(const void* xin, int ldx, int x_bf16, const float* dz, int lddz, float* dw,
                                  long P, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream) {
    UNET_CHECK_ARG(xin && dz && dw && ws && P > 0 && Cin > 0 && Cout > 0 && Cin % 4 == 0 && ldx % 4 == 0);
    const int nq = Cin / 4, tpp = nq < 256 ? nq : 256;
    UNET_CHECK_ARG(256 % tpp == 0 && nq % tpp == 0 && unet_aligned16(xin));
    const int blocks = direct_wgrad_blocks(P);
    if (ws_bytes < unet_conv1x1_wgrad_workspace(P, Cin, Cout)) return UNET_ENOSPC;
    const long ppb = (P + blocks - 1) / blocks;
    const size_t smem = (size_t)(256 / tpp) * 4 * tpp * 8 * sizeof(float);
    if (x_bf16) conv1x1_narrow_wgrad_kernel<1><<<blocks, 256, smem, (hipStream_t)stream>>>((const float*)xin, ldx, dz, lddz, (float*)ws, P, Cin, Cout, ppb);
    else        conv1x1_narrow_wgrad_kernel<0><<<blocks, 256, smem, (hipStream_t)stream>>>((const float*)xin, ldx, dz, lddz, (float*)ws, P, Cin, Cout, ppb);
    int rc = UNET_LAUNCH_STATUS(); if (rc) return rc;
    const long n = (long)Cin * Cout;
    sum_partials_kernel<<<unet_cdiv(n, 4), 256, 0, (hipStream_t)stream>>>((const float*)ws, dw, n, blocks);
    return UNET_LAUNCH_STATUS();
}
