// HBM-bound convolutions that are not GEMM-shaped (SURVEY.md 8(d) "exceptions"):
//   * the first 3x3 layer, Cin = number_channels (1..few)  -> 64   (UNet/model.py:88): K = 9*Cin is far too short
//     for the matrix cores; one output pixel is 256 B written for 9*Cin*64 FMAs -> VALU stencil, float4 stores;
//   * the 1x1 class-map layer 64 -> number_classes               (UNet/model.py:136): 1-3 flop/B.
// Thread mapping everywhere: consecutive lanes own consecutive channel quads (float4) of one pixel so global
// accesses are whole 64..256-B pixel rows.
#include "common.h"

namespace {

constexpr int CI_CHUNK = 8;

// ---- 3x3 'same' conv, small Cin, forward ---------------------------------------------------------------------
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
    for (long pix = p0 + pl; pix < p1; pix += npl) {
        long t = pix; const int xx = (int)(t % W); t /= W; const int y = (int)(t % H); const int n = (int)(t / H);
        const f32x4 g = *reinterpret_cast<const f32x4*>(dz + (size_t)pix * lddz + 4 * q);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int gy = y + tap / 3 - 1, gx = xx + tap % 3 - 1;
            float xv = 0.f;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) xv = x[((size_t)(n * H + gy) * W + gx) * ldx + ci];
            acc[tap] += xv * g;
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
                    xv[u] = *reinterpret_cast<const f32x4*>(x + (size_t)pix * ldx + 4 * cq);
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
__global__ __launch_bounds__(256) void conv1x1_narrow_dgrad_kernel(const float* __restrict__ dz, int lddz,
        const float* __restrict__ w, float* __restrict__ dx, int lddx, long P, int Cin, int K) {
    extern __shared__ __attribute__((aligned(16))) float sW[];        // [Cin][K]
    for (int i = threadIdx.x; i < Cin * K; i += 256) sW[i] = w[i];
    __syncthreads();
    const int nq = Cin >> 2;
    const long total = P * nq;
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const long pix = i / nq; const int cq = (int)(i - pix * nq);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < K; ++k) {
            const float g = dz[(size_t)pix * lddz + k];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += g * sW[(4 * cq + e) * K + k];
        }
        *reinterpret_cast<f32x4*>(dx + (size_t)pix * lddx + 4 * cq) = acc;
    }
}

// wgrad: part[blk][ci][k] = sum over the block's pixels of x[p][ci] * dz[p][k]   (k0..k0+8 per pass)
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
            for (long pix = p0 + pl; pix < p1; pix += npl) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (size_t)pix * ldx + 4 * qq);
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

extern "C" int unet_conv3x3_fwd_direct(const float* x, int ldx, const float* w, const float* bias, float* out, int ldo,
                                       int N, int H, int W, int Cin, int Cout, int relu, void* stream) {
    UNET_CHECK_ARG(x && w && out && N > 0 && H > 0 && W > 0 && Cin > 0 && ldx >= Cin && ldo >= Cout);
    const int tpp = Cout / 4;
    UNET_CHECK_ARG(Cout % 4 == 0 && tpp >= 1 && tpp <= 256 && 256 % tpp == 0 && ldo % 4 == 0 && unet_aligned16(out));
    UNET_CHECK_ARG(!bias || unet_aligned16(bias));
    const long P = (long)N * H * W;
    const int ppb = 256 / tpp;
    const size_t smem = (size_t)9 * CI_CHUNK * Cout * sizeof(float);
    UNET_CHECK_ARG(smem <= 64 * 1024);
    long blocks = (P + ppb - 1) / ppb; if (blocks > 4096) blocks = 4096;
    conv3x3_direct_fwd_kernel<<<(int)blocks, 256, smem, (hipStream_t)stream>>>(x, ldx, w, bias, out, ldo, N, H, W, Cin, Cout, relu);
    return UNET_LAUNCH_STATUS();
}

static int direct_wgrad_blocks(long P) { long b = (P + 1023) / 1024; if (b > 1024) b = 1024; if (b < 1) b = 1; return (int)b; }

extern "C" size_t unet_conv3x3_wgrad_direct_workspace(int N, int H, int W, int Cin, int Cout) {
    return (size_t)direct_wgrad_blocks((long)N * H * W) * 9 * Cin * Cout * sizeof(float);
}

extern "C" int unet_conv3x3_wgrad_direct(const float* xin, int ldx, const float* dz, int lddz, float* dw,
                                         int N, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream) {
    UNET_CHECK_ARG(xin && dz && dw && ws && N > 0 && H > 0 && W > 0 && Cin > 0 && Cin <= 65535);
    const int tpp = Cout / 4;
    UNET_CHECK_ARG(Cout % 4 == 0 && tpp >= 1 && tpp <= 256 && 256 % tpp == 0 && lddz % 4 == 0 && unet_aligned16(dz));
    const long P = (long)N * H * W;
    const int blocks = direct_wgrad_blocks(P);
    if (ws_bytes < unet_conv3x3_wgrad_direct_workspace(N, H, W, Cin, Cout)) return UNET_ENOSPC;
    const long ppb = (P + blocks - 1) / blocks;
    const size_t smem = (size_t)(256 / tpp) * 9 * Cout * sizeof(float);
    UNET_CHECK_ARG(smem <= 64 * 1024);
    conv3x3_direct_wgrad_kernel<<<dim3(blocks, Cin), 256, smem, (hipStream_t)stream>>>(xin, ldx, dz, lddz, (float*)ws, N, H, W, Cin, Cout, ppb);
    int rc = UNET_LAUNCH_STATUS(); if (rc) return rc;
    const long n = 9L * Cin * Cout;
    sum_partials_kernel<<<unet_cdiv(n, 4), 256, 0, (hipStream_t)stream>>>((const float*)ws, dw, n, blocks);
    return UNET_LAUNCH_STATUS();
}

extern "C" int unet_conv1x1_fwd(const float* x, int ldx, const float* w, const float* bias, float* out, int ldo,
                                long P, int Cin, int Cout, int relu, void* stream) {
    UNET_CHECK_ARG(x && w && out && P > 0 && Cin > 0 && Cout > 0 && Cin % 4 == 0 && ldx % 4 == 0 && ldx >= Cin && ldo >= Cout);
    UNET_CHECK_ARG(unet_aligned16(x) && (size_t)Cin * Cout * 4 <= 64 * 1024);
    long blocks = (P + 63) / 64; if (blocks > 4096) blocks = 4096;
    conv1x1_narrow_fwd_kernel<<<(int)blocks, 256, (size_t)Cin * Cout * 4, (hipStream_t)stream>>>(x, ldx, w, bias, out, ldo, P, Cin, Cout, relu);
    return UNET_LAUNCH_STATUS();
}

extern "C" int unet_conv1x1_dgrad(const float* dz, int lddz, const float* w, float* dx, int lddx,
                                  long P, int Cin, int Cout, void* stream) {
    UNET_CHECK_ARG(dz && w && dx && P > 0 && Cin > 0 && Cout > 0 && Cin % 4 == 0 && lddx % 4 == 0 && lddx >= Cin && lddz >= Cout);
    UNET_CHECK_ARG(unet_aligned16(dx) && (size_t)Cin * Cout * 4 <= 64 * 1024);
    long blocks = (P * (Cin / 4) + 255) / 256; if (blocks > 8192) blocks = 8192;
    conv1x1_narrow_dgrad_kernel<<<(int)blocks, 256, (size_t)Cin * Cout * 4, (hipStream_t)stream>>>(dz, lddz, w, dx, lddx, P, Cin, Cout);
    return UNET_LAUNCH_STATUS();
}

extern "C" size_t unet_conv1x1_wgrad_workspace(long P, int Cin, int Cout) {
    return (size_t)direct_wgrad_blocks(P) * Cin * Cout * sizeof(float);
}

extern "C" int unet_conv1x1_wgrad(const float* xin, int ldx, const float* dz, int lddz, float* dw,
                                  long P, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream) {
    UNET_CHECK_ARG(xin && dz && dw && ws && P > 0 && Cin > 0 && Cout > 0 && Cin % 4 == 0 && ldx % 4 == 0);
    const int nq = Cin / 4, tpp = nq < 256 ? nq : 256;
    UNET_CHECK_ARG(256 % tpp == 0 && nq % tpp == 0 && unet_aligned16(xin));
    const int blocks = direct_wgrad_blocks(P);
    if (ws_bytes < unet_conv1x1_wgrad_workspace(P, Cin, Cout)) return UNET_ENOSPC;
    const long ppb = (P + blocks - 1) / blocks;
    const size_t smem = (size_t)(256 / tpp) * 4 * tpp * 8 * sizeof(float);
    conv1x1_narrow_wgrad_kernel<<<blocks, 256, smem, (hipStream_t)stream>>>(xin, ldx, dz, lddz, (float*)ws, P, Cin, Cout, ppb);
    int rc = UNET_LAUNCH_STATUS(); if (rc) return rc;
    const long n = (long)Cin * Cout;
    sum_partials_kernel<<<unet_cdiv(n, 4), 256, 0, (hipStream_t)stream>>>((const float*)ws, dw, n, blocks);
    return UNET_LAUNCH_STATUS();
}
