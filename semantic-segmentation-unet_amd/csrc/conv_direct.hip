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
        unet_store<UNET_NT_FIRST>(reinterpret_cast<unet_u32x2*>(reinterpret_cast<uint16_t*>(base) + elem), unet_u32x2{h.x, h.y});
    } else unet_store<UNET_NT_FIRST>(reinterpret_cast<f32x4*>(base + elem), v);          // (st_quad's users only write: first layer, class-map input gradient)
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

// ---- class map of the network's shape: 64 channels, K <= 8 classes ----------------------------------------------------------------------
// Eight lanes per pixel, a lane owns a channel octet (one 16-byte access of a bf16 tensor, two of an fp32 one), so a wave moves 8 whole pixel
// rows per instruction; 32-bit pixel arithmetic (the generic kernels above divide a 64-bit index per element), four pixels in flight per lane.
__device__ __forceinline__ void cm_load8(const float* x, size_t elem, int b16, float (&v)[8]) {
    if (b16) {
        const uint4 h = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(x) + elem);
        const unsigned u[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[2 * j] = __builtin_bit_cast(float, u[j] << 16); v[2 * j + 1] = __builtin_bit_cast(float, u[j] & 0xffff0000u); }
    } else {
        const f32x4 a = *reinterpret_cast<const f32x4*>(x + elem), b = *reinterpret_cast<const f32x4*>(x + elem + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
    }
}
// the K class values of a pixel: one 16-byte load when the tensor is dense with K = 4
template <int KK> __device__ __forceinline__ void cm_loadk(const float* dz, size_t elem, int K, bool vec4, float (&g)[KK]) {
    if (vec4) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(dz + elem);
#pragma unroll
        for (int k = 0; k < KK; ++k) g[k] = k < 4 ? a[k & 3] : 0.f;
    } else {
#pragma unroll
        for (int k = 0; k < KK; ++k) g[k] = k < K ? dz[elem + k] : 0.f;
    }
}
// sum over the 8 lanes of a pixel, every lane ends with the total: two quad permutes and a half-row mirror, all folded into the add (DPP)
__device__ __forceinline__ float cm_sum8(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));     // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));     // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));    // row_half_mirror
    return v;
}

template <int KK>                     // KK = 4 or 8: accumulators per lane
__global__ __launch_bounds__(256) void classmap64_fwd_kernel(const float* __restrict__ x, int ldx, int x16, const float* __restrict__ w,
        const float* __restrict__ bias, float* __restrict__ out, int ldo, int P, int K, int relu) {
    const int sub = threadIdx.x & 7, pl = threadIdx.x >> 3;
    float wr[8][KK];
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int k = 0; k < KK; ++k) wr[e][k] = k < K ? w[(8 * sub + e) * K + k] : 0.f;
    const float bk = (bias && sub < K) ? bias[sub] : 0.f;
    const int stride = (int)gridDim.x * 32;
    constexpr int U = 4;
    for (int pix0 = (int)blockIdx.x * 32 + pl; pix0 < P; pix0 += U * stride) {
        float xv[U][8];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int pix = pix0 + u * stride < P ? pix0 + u * stride : pix0; cm_load8(x, (size_t)pix * ldx + 8 * sub, x16, xv[u]); }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float mine = 0.f;
#pragma unroll
            for (int k = 0; k < KK; ++k) {
                float sk = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) sk = fmaf(xv[u][e], wr[e][k], sk);
                sk = cm_sum8(sk);
                if (k == sub) mine = sk;
            }
            const int pix = pix0 + u * stride;
            if (sub < K && pix < P) {
                mine += bk;
                out[(size_t)pix * ldo + sub] = relu ? fmaxf(mine, 0.f) : mine;
            }
        }
    }
}

template <int KK>
__global__ __launch_bounds__(256) void classmap64_dgrad_kernel(const float* __restrict__ dz, int lddz, const float* __restrict__ w,
        float* __restrict__ dx, int lddx, int dx16, int P, int K) {
    const int sub = threadIdx.x & 7, pl = threadIdx.x >> 3;
    float wr[8][KK];
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int k = 0; k < KK; ++k) wr[e][k] = k < K ? w[(8 * sub + e) * K + k] : 0.f;
    const bool vec4 = K == 4 && lddz == 4;
    const int stride = (int)gridDim.x * 32;
    constexpr int U = 4;
    for (int pix0 = (int)blockIdx.x * 32 + pl; pix0 < P; pix0 += U * stride) {
        float g[U][KK];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int pix = pix0 + u * stride < P ? pix0 + u * stride : pix0; cm_loadk<KK>(dz, (size_t)pix * lddz, K, vec4, g[u]); }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int pix = pix0 + u * stride;
            if (pix >= P) break;
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                acc[e] = 0.f;
#pragma unroll
                for (int k = 0; k < KK; ++k) acc[e] = fmaf(g[u][k], wr[e][k], acc[e]);
            }
            const size_t o = (size_t)pix * lddx + 8 * sub;
            if (dx16) {
                uint4 h;
                asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h.x) : "v"(acc[0]), "v"(acc[1]));
                asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h.y) : "v"(acc[2]), "v"(acc[3]));
                asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h.z) : "v"(acc[4]), "v"(acc[5]));
                asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h.w) : "v"(acc[6]), "v"(acc[7]));
                unet_store<UNET_NT_FIRST>(reinterpret_cast<unet_u32x4*>(reinterpret_cast<uint16_t*>(dx) + o), unet_u32x4{h.x, h.y, h.z, h.w});
            } else {
                *reinterpret_cast<f32x4*>(dx + o) = f32x4{acc[0], acc[1], acc[2], acc[3]};
                *reinterpret_cast<f32x4*>(dx + o + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
            }
        }
    }
}

// part[blk][ci][k] = sum over the block's pixels of x[p][ci] * dz[p][k]: per-lane sums, then the 8 pixel lanes of a wave (xor shuffles), then the
// 4 waves through LDS, all in a fixed order
template <int KK>
__global__ __launch_bounds__(256) void classmap64_wgrad_kernel(const float* __restrict__ x, int ldx, int x16, const float* __restrict__ dz, int lddz,
        float* __restrict__ part, int P, int K, int pix_per_block) {
    __shared__ float sR[4][64][KK];
    const int sub = threadIdx.x & 7, pl = threadIdx.x >> 3, wv = threadIdx.x >> 6;
    const bool vec4 = K == 4 && lddz == 4;
    const int p0 = (int)blockIdx.x * pix_per_block;
    const int p1 = p0 + pix_per_block < P ? p0 + pix_per_block : P;
    float acc[8][KK];
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int k = 0; k < KK; ++k) acc[e][k] = 0.f;
    constexpr int U = 4;
    for (int pix0 = p0 + pl; pix0 < p1; pix0 += U * 32) {
        float xv[U][8], g[U][KK];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int pix = pix0 + u * 32 < p1 ? pix0 + u * 32 : pix0;
            cm_load8(x, (size_t)pix * ldx + 8 * sub, x16, xv[u]);
            cm_loadk<KK>(dz, (size_t)pix * lddz, K, vec4, g[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (pix0 + u * 32 >= p1) break;
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int k = 0; k < KK; ++k) acc[e][k] = fmaf(xv[u][e], g[u][k], acc[e][k]);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int k = 0; k < KK; ++k) {
            float v = acc[e][k];
            v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
            if ((threadIdx.x & 63) < 8) sR[wv][8 * sub + e][k] = v;
        }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * KK; i += 256) {
        const int ci = i / KK, k = i % KK;
        if (k < K) part[((size_t)blockIdx.x * 64 + ci) * K + k] = (sR[0][ci][k] + sR[1][ci][k]) + (sR[2][ci][k] + sR[3][ci][k]);
    }
}

// ---- first 3x3 layer (Cin = 1..4 -> 64) on the fp32 matrix cores ----------------------------------------------------------------------
// The stencil kernels above keep 9 * Cin weight quads in registers (108 VGPRs at Cin = 3: two waves per SIMD) and gather their 3 x 6 x Cin
// window with one conditional scalar load per value: at Cin = 3 and a bf16 output (268 MB written at 8 x 512^2) the forward ran 0.19 ms and the
// weight gradient 0.19 ms against 0.06 ms of HBM time -- latency-bound at that occupancy.  Here the contraction (K = 9 * Cin, padded to an
// even number) runs on v_mfma_f32_32x32x2_f32 with fp32 operands (the layer is outside the bf16 contract: its input is the image):
//   forward   D[pixel][co] : A = window values gathered straight from global memory (the image is L2-resident: 25 MB), one dword per lane and
//             k-step, zeros outside the image through the buffer range check; B = the whole filter in 2 * K/2 registers per lane;
//   wgrad     D[k][co]     : the contraction runs over pixels, two per MFMA; A = window value k of the two pixels, B = their dz rows.
// No LDS in the loops, every wave walks 32-pixel row segments on its own.  Output channel of MFMA column j, accumulator c = 2 j + c,
// so a lane's pair of results is one 4-byte (bf16) or 8-byte (fp32) store and a row of 32 lanes writes the pixel's whole 64-channel row.
#ifndef UNET_FIRST_ABLATE
#define UNET_FIRST_ABLATE 0      /* diagnostic builds (scripts/build_variant.sh; results wrong): 1 no stores, 2 no MFMAs, 4 no window loads, 8 no statistics */
#endif
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef int first_rsrc_t __attribute__((ext_vector_type(4)));
// The window / dz loads of these kernels are asm volatile with hand-counted waits: written as builtins the compiler SANK each prefetch down to
// its consumer (the loads of the next segment landed right in front of that segment's MFMAs, one vmcnt wait per MFMA pair: matrix pipe 46 %
// busy, 0.11-0.12 ms per launch).  Loads, stores and DMA count together in issue order, so "all but the N youngest" is exact.
__device__ __forceinline__ first_rsrc_t first_rsrc(const void* p, size_t bytes) {
    const unsigned long long a = (unsigned long long)p;
    return first_rsrc_t{(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xffffu), (int)(unsigned)bytes, 0x00020000};
}
__device__ __forceinline__ void first_ld32(float& dst, int voff, first_rsrc_t r) {
    asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(dst) : "v"(voff), "s"(r) : "memory");
}
__device__ __forceinline__ void first_ld32u(unsigned& dst, int voff, first_rsrc_t r) {
    asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(dst) : "v"(voff), "s"(r) : "memory");
}
// (the pair's part of the address as the instruction's immediate offset, < 4096: no vector instruction per load)
#define FIRST_LD32_IMM(dst, voff, r, imm) asm volatile("buffer_load_dword %0, %1, %2, 0 offen offset:%3" : "=v"(dst) : "v"(voff), "s"(r), "n"(imm) : "memory")
template <int N> __device__ __forceinline__ void first_wait() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <int CIN> struct FirstGeom {
    static constexpr int K = 9 * CIN, S = (K + 1) / 2;
    // window element k = tap * CIN + ci (the HWIO order of the filter): row offset da, column offset db, channel ci
    static __device__ __forceinline__ int da(int k) { return k / (3 * CIN) - 1; }
    static __device__ __forceinline__ int db(int k) { return (k / CIN) % 3 - 1; }
    static __device__ __forceinline__ int ci(int k) { return k % CIN; }
};

template <int CIN, int OUT16>
__global__ __launch_bounds__(256) void conv3x3_first_mfma_fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
        const float* __restrict__ bias, void* __restrict__ out, int ldo, int N, int H, int W, int relu, float* __restrict__ stat_part) {
    typedef FirstGeom<CIN> G;
    constexpr int K = G::K, S = G::S;
    __shared__ float sStat[4 * 64 * 2 * 2];
    const int lane = threadIdx.x & 63, j = lane & 31, lh = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float wr[S][2];
    int koff[S];
    unsigned top = 0, bot = 0, left = 0, right = 0, pad = 0;          // bit s: window element k = 2 s + lh of this lane looks up / down / left / right / is padding
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int k = 2 * s + lh;
        const bool real = k < K;
        const int kk = real ? k : 0;
        wr[s][0] = real ? w[kk * 64 + 2 * j] : 0.f; wr[s][1] = real ? w[kk * 64 + 2 * j + 1] : 0.f;
        koff[s] = ((G::da(kk) * W + G::db(kk)) * ldx + G::ci(kk)) * 4;
        top |= (unsigned)(G::da(kk) < 0) << s; bot |= (unsigned)(G::da(kk) > 0) << s;
        left |= (unsigned)(G::db(kk) < 0) << s; right |= (unsigned)(G::db(kk) > 0) << s; pad |= (unsigned)(!real) << s;
    }
    const float b0 = bias ? bias[2 * j] : 0.f, b1 = bias ? bias[2 * j + 1] : 0.f;
    const float lo = relu ? 0.f : -__builtin_inff();
    const first_rsrc_t srd_x = first_rsrc(x, (size_t)N * H * W * ldx * 4);
    const __amdgpu_buffer_rsrc_t srd_o = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)((size_t)N * H * W * ldo * (OUT16 ? 2 : 4)), 0x00020000);
    const int SW = (W + 31) >> 5;
    const int nseg = N * H * SW, stride = (int)gridDim.x * 4;
    float st1[2] = {0.f, 0.f}, st2[2] = {0.f, 0.f};
    float a[2][S];
    auto gather = [&](float (&dst)[S], int seg) {
        const int sx = seg % SW, row = seg / SW, y = row % H;            // row = n * H + y
        const int px = 32 * sx + j;
        unsigned bad = pad | (y == 0 ? top : 0u) | (y == H - 1 ? bot : 0u) | (px == 0 ? left : 0u) | (px + 1 >= W ? right : 0u) | (px >= W ? ~0u : 0u);
        const int base = (row * W + px) * ldx * 4;
#pragma unroll
        for (int s = 0; s < S; ++s)
            if (UNET_FIRST_ABLATE & 4) dst[s] = __builtin_bit_cast(float, base + koff[s] + (int)((bad >> s) & 1));
            else first_ld32(dst[s], ((bad >> s) & 1) ? (int)0x80000000 : base + koff[s], srd_x);
    };
    // One segment: the NEXT segment's window loads are issued first (into the other buffer), then the MFMAs and the epilogue of this one.
    // Measured at 8 x 512^2 x 3 -> 64 with a bf16 output (ms per launch): 0.045 without MFMAs and stores, 0.105 with the MFMAs, 0.117 with
    // both -- the parts ADD UP whatever the arrangement: the epilogue placed between the MFMA pairs out of a second accumulator set (clean
    // interleave in the ISA, two waves per SIMD) ran the same 0.118 as this form with three waves per SIMD.  v_mfma_f32_32x32x2_f32 runs on the
    // vector ALUs' fp32 multipliers (its rate IS the packed-fp32 vector rate), so vector instructions do not hide behind it as they do behind
    // the bf16 matrix instructions; the lever left is fewer vector instructions (~15 per stored pixel pair today).
    // Vector-memory operations younger than `cur`'s loads at the wait: the previous segment's 16 stores (none before the first) and the next
    // segment's S loads (none for the last).
    const int pstep = ldo * (OUT16 ? 2 : 4);
    auto step = [&](float (&cur)[S], float (&nxt)[S], int seg, bool first) {
        const bool more = seg + stride < nseg;
        if (more) gather(nxt, seg + stride);
        if (first) { if (more) first_wait<S>(); else first_wait<0>(); }
        else       { if (more) first_wait<S + 16>(); else first_wait<16>(); }
#pragma unroll
        for (int s = 0; s < S; ++s) asm volatile("" : "+v"(cur[s]));       // (the MFMAs below read `cur` after the wait)
        const f32x16 zero = {};
        f32x16 acc0, acc1;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (UNET_FIRST_ABLATE & 2) { acc0[s & 15] += cur[s] * wr[s][0]; acc1[s & 15] += cur[s] * wr[s][1]; continue; }
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[s], wr[s][0], s == 0 ? zero : acc0, 0, 0, 0);   // (starts from the constant 0: no 32 register writes)
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[s], wr[s][1], s == 0 ? zero : acc1, 0, 0, 0);
        }
        const int sx = seg % SW, row = seg / SW;
        const int px0 = 32 * sx + 4 * lh;
        const int obase = ((row * W + px0) * ldo + 2 * j) * (OUT16 ? 2 : 4);
        const int wrem = W - px0;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int dp = (e & 3) + 8 * (e >> 2);
            const bool ok = dp < wrem;
            const float v0 = fmaxf(acc0[e] + b0, lo), v1 = fmaxf(acc1[e] + b1, lo);
            const float m0 = ok ? v0 : 0.f, m1 = ok ? v1 : 0.f;
            if (!(UNET_FIRST_ABLATE & 8)) { st1[0] += m0; st2[0] += m0 * m0; st1[1] += m1; st2[1] += m1 * m1; }    // (sums of the fp32 values, before any rounding of the stored tensor)
            else { st1[0] += m0; }
            if ((UNET_FIRST_ABLATE & 1) && st1[0] != 1.2345e38f) continue;
            const int vo = ok ? obase + dp * pstep : (int)0x80000000;
            if constexpr (OUT16) {
                unsigned h;
                asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h) : "v"(v0), "v"(v1));
                __builtin_amdgcn_raw_buffer_store_b32(h, srd_o, vo, 0, UNET_NT_AUX(UNET_NT_FIRST));
            } else {
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                u32x2 o; o[0] = __builtin_bit_cast(unsigned, v0); o[1] = __builtin_bit_cast(unsigned, v1);
                __builtin_amdgcn_raw_buffer_store_b64(o, srd_o, vo, 0, UNET_NT_AUX(UNET_NT_FIRST));
            }
        }
    };
    int seg = (int)blockIdx.x * 4 + wv;
    if (seg < nseg) gather(a[0], seg);
    bool first = true;
    while (seg < nseg) {
        step(a[0], a[1], seg, first); seg += stride; first = false;
        if (seg >= nseg) break;
        step(a[1], a[0], seg, false); seg += stride;
    }
    if (stat_part) {            // one row of partials per block, layout of unet_bn_train_finalize_partials: [C/64][rows][64][2]; fixed order
#pragma unroll
        for (int c = 0; c < 2; ++c) { st1[c] += __shfl_xor(st1[c], 32); st2[c] += __shfl_xor(st2[c], 32); }
        if (lh == 0) {
#pragma unroll
            for (int c = 0; c < 2; ++c) { sStat[(wv * 64 + 2 * j + c) * 2] = st1[c]; sStat[(wv * 64 + 2 * j + c) * 2 + 1] = st2[c]; }
        }
        __syncthreads();
        if (threadIdx.x < 128) {
            float t = 0.f;
#pragma unroll
            for (int l = 0; l < 4; ++l) t += sStat[l * 128 + threadIdx.x];
            stat_part[(size_t)blockIdx.x * 128 + threadIdx.x] = t;
        }
    }
}

// weight gradient (dz stored as bf16): dw[k][co] = sum over pixels of window value k of the pixel x dz[pixel][co]; partials per block ->
// sum_partials_kernel
template <int CIN>
__global__ __launch_bounds__(256) void conv3x3_first_mfma_wgrad_kernel(const float* __restrict__ x, int ldx, const void* __restrict__ dz, int lddz,
        float* __restrict__ part, int N, int H, int W) {
    typedef FirstGeom<CIN> G;
    constexpr int K = G::K;
    static_assert(K <= 32, "window elements are MFMA rows");
    __shared__ float sAcc[4][2][16][64];
    const int lane = threadIdx.x & 63, j = lane & 31, lh = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // A operand: MFMA row i = lane & 31 = window element k, reduce index lane >> 5 = which of the two pixels
    const bool real = j < K;
    const int kk = real ? j : 0;
    const int da = G::da(kk), db = G::db(kk);
    const int koff = ((da * W + db + lh) * ldx + G::ci(kk)) * 4;
    const first_rsrc_t srd_x = first_rsrc(x, (size_t)N * H * W * ldx * 4);
    const first_rsrc_t srd_z = first_rsrc(dz, (size_t)N * H * W * lddz * 2);
    const int SW = (W + 31) >> 5;
    const int nseg = N * H * SW, stride = (int)gridDim.x * 4;
    f32x16 acc0, acc1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
    struct Raw { float a[16]; unsigned z[16]; };
    auto issue = [&](Raw& r, int seg) {
        const int sx = seg % SW, row = seg / SW, y = row % H;
        const bool rowok = real && (unsigned)(y + da) < (unsigned)H;
        const int x0 = 32 * sx;
        const int abase = (row * W + x0) * ldx * 4 + koff;
        const int zbase = ((row * W + x0 + lh) * lddz + 2 * j) * 2;
        if (x0 + 32 <= W && ldx == CIN && lddz == 64) {
            // whole segment inside the row, dense tensors: only the first and the last pair can look past the row's ends, and the pair's
            // part of every address is an instruction immediate (the fp32 matrix instructions run on the vector ALUs' multipliers: vector
            // instructions do not hide behind them -- per-load address arithmetic was 190 of this loop's 230 vector instructions per segment)
            const int va = rowok ? abase : (int)0x80000000;
            const int va0 = (rowok && x0 + lh + db >= 0) ? abase : (int)0x80000000, va15 = (rowok && x0 + 30 + lh + db < W) ? abase : (int)0x80000000;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                FIRST_LD32_IMM(r.a[t], t == 0 ? va0 : t == 15 ? va15 : va, srd_x, 2 * t * CIN * 4);
                FIRST_LD32_IMM(r.z[t], zbase, srd_z, t * 256);
            }
            return;
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int col = x0 + 2 * t + lh;                          // this lane's pixel of pair t
            const bool aok = rowok && (unsigned)(col + db) < (unsigned)W;
            first_ld32(r.a[t], aok ? abase + 2 * t * ldx * 4 : (int)0x80000000, srd_x);
            const int zo = col < W ? zbase + 2 * t * lddz * 2 : (int)0x80000000;
            first_ld32u(r.z[t], zo, srd_z);
        }
    };
    // one segment: the NEXT segment's 32 loads go out first, then this one's 32 MFMAs
    auto step = [&](Raw& cur, Raw& nxt, int seg) {
        if (seg + stride < nseg) { issue(nxt, seg + stride); first_wait<32>(); } else first_wait<0>();
#pragma unroll
        for (int t = 0; t < 16; ++t) asm volatile("" : "+v"(cur.a[t]), "+v"(cur.z[t]));   // (the MFMAs below read `cur` after the wait)
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const unsigned h = cur.z[t];                              // channels 2 j (low half) and 2 j + 1 of the pixel
            const float z0 = __builtin_bit_cast(float, h << 16), z1 = __builtin_bit_cast(float, h & 0xffff0000u);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[t], z0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[t], z1, acc1, 0, 0, 0);
        }
    };
    Raw r0, r1;
    int seg = (int)blockIdx.x * 4 + wv;
    if (seg < nseg) issue(r0, seg);
    while (seg < nseg) {
        step(r0, r1, seg); seg += stride;
        if (seg >= nseg) break;
        step(r1, r0, seg); seg += stride;
    }
    // the four waves in a fixed order; accumulator e of column j = dw[k = (e & 3) + 8 (e >> 2) + 4 lh][co = 2 j + c]
#pragma unroll
    for (int e = 0; e < 16; ++e) { sAcc[wv][lh][e][2 * j] = acc0[e]; sAcc[wv][lh][e][2 * j + 1] = acc1[e]; }
    __syncthreads();
    for (int i = threadIdx.x; i < K * 64; i += 256) {
        const int k = i >> 6, co = i & 63;
        const int h = (k >> 2) & 1, e = (k & 3) + 4 * (k >> 3);
        part[(size_t)blockIdx.x * K * 64 + i] = (sAcc[0][h][e][co] + sAcc[1][h][e][co]) + (sAcc[2][h][e][co] + sAcc[3][h][e][co]);
    }
}

// the matrix-core forms serve Cout = 64 with tensors below 2 GiB (32-bit buffer offsets)
static bool first_mfma_shape(int N, int H, int W, int Cin, int Cout) {       // (sizes at the tightest leading dimensions: what the row-count query can know)
#ifdef UNET_FIRST_STENCIL        /* diagnostic builds (scripts/build_variant.sh): the stencil kernels everywhere */
    return false;
#endif
    return Cin >= 1 && Cin <= 4 && Cout == 64 && (size_t)N * H * W * 64 * 4 < ((size_t)1 << 31);
}
static bool first_mfma_fits(int N, int H, int W, int ldx, int ld64, int es64) {
    return (size_t)N * H * W * ldx * 4 < ((size_t)1 << 31) && (size_t)N * H * W * ld64 * es64 < ((size_t)1 << 31);
}
// one resident wave of workgroups (register-limited: 3-5 per CU), each wave of which walks its share of the segments: a grid of 1024
// on 768 slots ran two rounds, the second a third full (0.123 ms instead of ~0.09 at 8 x 512^2 x 3)
template <int CIN> static int first_fwd_slots() {
    static int slots = 0;
    if (!slots) {
        int per_cu = 0, dev = 0; hipDeviceProp_t pr;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv3x3_first_mfma_fwd_kernel<CIN, 1>, 256, 0) != hipSuccess || per_cu < 1) per_cu = 2;
        (void)hipGetDevice(&dev);
        const int cus = (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
        slots = per_cu * cus;
    }
    return slots;
}
static int first_fwd_blocks(int N, int H, int W, int Cin) {
    const long seg = (long)N * H * ((W + 31) / 32);
    const int slots = Cin == 1 ? first_fwd_slots<1>() : Cin == 2 ? first_fwd_slots<2>() : Cin == 3 ? first_fwd_slots<3>() : first_fwd_slots<4>();
    long b = (seg + 3) / 4; if (b > slots) b = slots; if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

// one pass of 32 pixels per block and trip, four trips in flight: enough blocks to fill the chip a few times over, few enough to amortise the weights
static int classmap64_blocks(long P) { long b = (P + 127) / 128; if (b > 4096) b = 4096; if (b < 1) b = 1; return (int)b; }

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
    if (first_mfma_shape(N, H, W, Cin, Cout) && !(first_mfma_fits(N, H, W, ldx, ldo, out_bf16 ? 2 : 4) && ldo % 2 == 0) && stat_part)
        return UNET_EINVAL;                                            // (the row count promised by unet_conv3x3_fwd_direct_stats_rows is the matrix-core kernel's)
    if (first_mfma_shape(N, H, W, Cin, Cout) && first_mfma_fits(N, H, W, ldx, ldo, out_bf16 ? 2 : 4) && ldo % 2 == 0) {
        const int b1 = first_fwd_blocks(N, H, W, Cin);
        hipStream_t st = (hipStream_t)stream;
        if (stat_part && stat_bytes < (size_t)b1 * 128 * sizeof(float)) return UNET_ENOSPC;
#define UNET_FIRST(CI) do { if (out_bf16) conv3x3_first_mfma_fwd_kernel<CI, 1><<<b1, 256, 0, st>>>(x, ldx, w, bias, out, ldo, N, H, W, relu, stat_part); \
                            else          conv3x3_first_mfma_fwd_kernel<CI, 0><<<b1, 256, 0, st>>>(x, ldx, w, bias, out, ldo, N, H, W, relu, stat_part); } while (0)
        if (Cin == 1) UNET_FIRST(1); else if (Cin == 2) UNET_FIRST(2); else if (Cin == 3) UNET_FIRST(3); else UNET_FIRST(4);
#undef UNET_FIRST
        return UNET_LAUNCH_STATUS();
    }
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
    if (N > 0 && H > 0 && W > 0 && first_mfma_shape(N, H, W, Cin, Cout)) return first_fwd_blocks(N, H, W, Cin);
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
    if (dz_bf16 && first_mfma_shape(N, H, W, Cin, Cout) && 9 * Cin <= 32 && first_mfma_fits(N, H, W, ldx, lddz, 2) && lddz % 2 == 0) {
        // (9 * Cin window elements are the 32 MFMA rows: Cin <= 3; the workspace holds `blocks` partial filters either way.  With an fp32 dz
        // the stencil kernels below are at the tensor's HBM time already -- 0.120 ms for 537 MB at 8 x 512^2 against 0.136 here)
        if (Cin == 1)      conv3x3_first_mfma_wgrad_kernel<1><<<blocks, 256, 0, st>>>(xin, ldx, dzv, lddz, (float*)ws, N, H, W);
        else if (Cin == 2) conv3x3_first_mfma_wgrad_kernel<2><<<blocks, 256, 0, st>>>(xin, ldx, dzv, lddz, (float*)ws, N, H, W);
        else               conv3x3_first_mfma_wgrad_kernel<3><<<blocks, 256, 0, st>>>(xin, ldx, dzv, lddz, (float*)ws, N, H, W);
    } else
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
    if (Cin == 64 && Cout <= 8 && ldx % 8 == 0 && P < (1L << 31)) {            // the network's class map
        const int b8 = classmap64_blocks(P);
        if (Cout <= 4) classmap64_fwd_kernel<4><<<b8, 256, 0, (hipStream_t)stream>>>((const float*)x, ldx, x_bf16, w, bias, out, ldo, (int)P, Cout, relu);
        else           classmap64_fwd_kernel<8><<<b8, 256, 0, (hipStream_t)stream>>>((const float*)x, ldx, x_bf16, w, bias, out, ldo, (int)P, Cout, relu);
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
    // (bf16 dx only: an fp32 row of 64 channels is two 16-byte stores per lane at a 32-byte stride, and the quad-per-lane kernel below is
    // faster there -- 0.144 against 0.160 ms at 8 x 512^2; same products in the same order, so the two agree bit for bit)
    if (dx_bf16 && Cin == 64 && Cout <= 8 && lddx % 8 == 0 && P < (1L << 31) && (Cout != 4 || lddz != 4 || unet_aligned16(dz))) {
        const int b8 = classmap64_blocks(P);
        if (Cout <= 4) classmap64_dgrad_kernel<4><<<b8, 256, 0, (hipStream_t)stream>>>(dz, lddz, w, (float*)dx, lddx, dx_bf16, (int)P, Cout);
        else           classmap64_dgrad_kernel<8><<<b8, 256, 0, (hipStream_t)stream>>>(dz, lddz, w, (float*)dx, lddx, dx_bf16, (int)P, Cout);
        return UNET_LAUNCH_STATUS();
    }
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
    if (Cin == 64 && Cout <= 8 && ldx % 8 == 0 && P < (1L << 31) && (Cout != 4 || lddz != 4 || unet_aligned16(dz))) {
        if (Cout <= 4) classmap64_wgrad_kernel<4><<<blocks, 256, 0, (hipStream_t)stream>>>((const float*)xin, ldx, x_bf16, dz, lddz, (float*)ws, (int)P, Cout, (int)ppb);
        else           classmap64_wgrad_kernel<8><<<blocks, 256, 0, (hipStream_t)stream>>>((const float*)xin, ldx, x_bf16, dz, lddz, (float*)ws, (int)P, Cout, (int)ppb);
        int rc2 = UNET_LAUNCH_STATUS(); if (rc2) return rc2;
        sum_partials_kernel<<<unet_cdiv((long)Cin * Cout, 4), 256, 0, (hipStream_t)stream>>>((const float*)ws, dw, (long)Cin * Cout, blocks);
        return UNET_LAUNCH_STATUS();
    }
    if (x_bf16) conv1x1_narrow_wgrad_kernel<1><<<blocks, 256, smem, (hipStream_t)stream>>>((const float*)xin, ldx, dz, lddz, (float*)ws, P, Cin, Cout, ppb);
    else        conv1x1_narrow_wgrad_kernel<0><<<blocks, 256, smem, (hipStream_t)stream>>>((const float*)xin, ldx, dz, lddz, (float*)ws, P, Cin, Cout, ppb);
    int rc = UNET_LAUNCH_STATUS(); if (rc) return rc;
    const long n = (long)Cin * Cout;
    sum_partials_kernel<<<unet_cdiv(n, 4), 256, 0, (hipStream_t)stream>>>((const float*)ws, dw, n, blocks);
    return UNET_LAUNCH_STATUS();
}
