// Winograd F(2x2, 3x3) for the 3x3 'same' convolutions with many channels (reference layer: UNet/model.py:28-35).
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A        per 2x2 output tile, 4x4 input patch d, 3x3 kernel g
//
// Exact fp32 arithmetic, 16 multiplies per tile and channel pair instead of 36 (2.25x fewer MACs on the matrix cores).
// Unfused pipeline: input transform (HBM-bound, writes 16 planes [tile][Cin] = 4x the input bytes) -> 16 independent
// fp32-MFMA GEMMs [T x Cin] x [Cin x Cout] (conv_igemm.hip, MODE 3) -> output transform (+bias, ReLU).  The transforms
// move ~14x the activation bytes, so the engine uses this path only where channels are wide enough (>= 256) for the
// saved matrix work to dominate.  dgrad is the same pipeline on dz with the 180-degree rotated, in/out-swapped kernel.
//
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]   A^T = [1 1 1 0; 0 1 -1 -1]
#include "common.h"

namespace {

// U[xi][k][n], xi = 4*i + j.  mode 0 (forward): k = ci, n = co, g[a][b] = w[a][b][ci][co];
// mode 1 (dgrad): k = co, n = ci, g[a][b] = w[2-a][2-b][ci][co].
__global__ __launch_bounds__(256) void wino_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int Ci, int Co, int mode) {
    const long total = (long)Ci * Co, stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const int ci = (int)(i / Co), co = (int)(i % Co);
        float g[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int aa = mode ? 2 - a : a, bb = mode ? 2 - b : b;
                g[a][b] = w[((size_t)(aa * 3 + bb) * Ci + ci) * Co + co];
            }
        float s[4][3];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            s[0][b] = g[0][b];
            s[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
            s[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
            s[3][b] = g[2][b];
        }
        const size_t plane = (size_t)Ci * Co;
        const size_t off = mode ? (size_t)co * Ci + ci : (size_t)ci * Co + co;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            U[(size_t)(4 * r + 0) * plane + off] = s[r][0];
            U[(size_t)(4 * r + 1) * plane + off] = 0.5f * (s[r][0] + s[r][1] + s[r][2]);
            U[(size_t)(4 * r + 2) * plane + off] = 0.5f * (s[r][0] - s[r][1] + s[r][2]);
            U[(size_t)(4 * r + 3) * plane + off] = s[r][2];
        }
    }
}

// V[xi][tile][c] = (B^T d B)[xi]; thread = (tile, channel quad); patch rows 2ty-1..2ty+2, zero outside the image
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x, int ldx, float* __restrict__ V,
                                                         int N, int H, int W, int C) {
    const int Th = H >> 1, Tw = W >> 1, nq = C >> 2;
    const long T = (long)N * Th * Tw, total = T * nq, stride = (long)gridDim.x * 256;
    const size_t plane = (size_t)T * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const long tile = i / nq; const int c0 = (int)(i - tile * nq) * 4;
        long t = tile; const int tx = (int)(t % Tw); t /= Tw; const int ty = (int)(t % Th); const int n = (int)(t / Th);
        f32x4 d[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gy = 2 * ty - 1 + r;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int gx = 2 * tx - 1 + c;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
                    v = *reinterpret_cast<const f32x4*>(x + ((size_t)(n * H + gy) * W + gx) * ldx + c0);
                d[r][c] = v;
            }
        }
        f32x4 tt[4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            tt[0][c] = d[0][c] - d[2][c]; tt[1][c] = d[1][c] + d[2][c];
            tt[2][c] = d[2][c] - d[1][c]; tt[3][c] = d[1][c] - d[3][c];
        }
        float* dst = V + (size_t)tile * C + c0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            *reinterpret_cast<f32x4*>(dst + (size_t)(4 * r + 0) * plane) = tt[r][0] - tt[r][2];
            *reinterpret_cast<f32x4*>(dst + (size_t)(4 * r + 1) * plane) = tt[r][1] + tt[r][2];
            *reinterpret_cast<f32x4*>(dst + (size_t)(4 * r + 2) * plane) = tt[r][2] - tt[r][1];
            *reinterpret_cast<f32x4*>(dst + (size_t)(4 * r + 3) * plane) = tt[r][1] - tt[r][3];
        }
    }
}

// out[2ty+i][2tx+j] = (A^T m A)[i][j] + bias, ReLU; thread = (tile, channel quad)
__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ M, const float* __restrict__ bias,
        float* __restrict__ out, int ldo, int N, int H, int W, int C, int relu) {
    const int Th = H >> 1, Tw = W >> 1, nq = C >> 2;
    const long T = (long)N * Th * Tw, total = T * nq, stride = (long)gridDim.x * 256;
    const size_t plane = (size_t)T * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const long tile = i / nq; const int c0 = (int)(i - tile * nq) * 4;
        long t = tile; const int tx = (int)(t % Tw); t /= Tw; const int ty = (int)(t % Th); const int n = (int)(t / Th);
        const float* src = M + (size_t)tile * C + c0;
        f32x4 m[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) m[r][c] = *reinterpret_cast<const f32x4*>(src + (size_t)(4 * r + c) * plane);
        f32x4 rr[2][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { rr[0][c] = m[0][c] + m[1][c] + m[2][c]; rr[1][c] = m[1][c] - m[2][c] - m[3][c]; }
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (bias) bv = *reinterpret_cast<const f32x4*>(bias + c0);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            f32x4 y0 = rr[r][0] + rr[r][1] + rr[r][2] + bv, y1 = rr[r][1] - rr[r][2] - rr[r][3] + bv;
            if (relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { y0[e] = fmaxf(y0[e], 0.f); y1[e] = fmaxf(y1[e], 0.f); }
            }
            float* o = out + ((size_t)(n * H + 2 * ty + r) * W + 2 * tx) * ldo + c0;
            *reinterpret_cast<f32x4*>(o) = y0;
            *reinterpret_cast<f32x4*>(o + ldo) = y1;
        }
    }
}

int grid_for(long total, int cap) { long b = (total + 255) / 256; if (b > cap) b = cap; if (b < 1) b = 1; return (int)b; }

bool wino_ok(int N, int H, int W, int Ci, int Co) {
    const long T = (long)N * (H / 2) * (W / 2);
    return H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && T % 32 == 0 && Ci % 32 == 0 && Co % 64 == 0 && Ci % 4 == 0;
}

int run_wino(const float* x, int ldx, const float* U, const float* bias, float* out, int ldo, int N, int H, int W,
             int Kc, int Nc, int relu, float* V, float* M, hipStream_t st) {
    const long T = (long)N * (H / 2) * (W / 2);
    wino_input_kernel<<<grid_for(T * (Kc / 4), 16384), 256, 0, st>>>(x, ldx, V, N, H, W, Kc);
    int rc = UNET_LAUNCH_STATUS(); if (rc) return rc;
    rc = unet_igemm_batched_planes(V, U, M, T, Kc, Nc, 16, st); if (rc) return rc;
    wino_output_kernel<<<grid_for(T * (Nc / 4), 16384), 256, 0, st>>>(M, bias, out, ldo, N, H, W, Nc, relu);
    return UNET_LAUNCH_STATUS();
}

}  // namespace

extern "C" int unet_winograd_supported(int N, int H, int W, int Cin, int Cout) {
    return (wino_ok(N, H, W, Cin, Cout) && Cin % 64 == 0) ? 1 : 0;
}

// U must hold 16*Cin*Cout floats.  mode 0: forward kernel transform; mode 1: data-gradient kernel transform.
extern "C" int unet_winograd_weight_transform(const float* w, float* U, int Cin, int Cout, int mode, void* stream) {
    UNET_CHECK_ARG(w && U && Cin > 0 && Cout > 0 && (mode == 0 || mode == 1));
    wino_weight_kernel<<<grid_for((long)Cin * Cout, 4096), 256, 0, (hipStream_t)stream>>>(w, U, Cin, Cout, mode);
    return UNET_LAUNCH_STATUS();
}

extern "C" size_t unet_conv3x3_winograd_workspace(int N, int H, int W, int Cin, int Cout) {
    const size_t T = (size_t)N * (H / 2) * (W / 2);
    return 16 * T * ((size_t)Cin + Cout) * sizeof(float);
}

// forward: out = relu?(conv3x3_same(x, w) + bias) with U = unet_winograd_weight_transform(w, mode 0)
extern "C" int unet_conv3x3_fwd_winograd(const float* x, int ldx, const float* U, const float* bias, float* out, int ldo,
        int N, int H, int W, int Cin, int Cout, int relu, void* ws, size_t ws_bytes, void* stream) {
    UNET_CHECK_ARG(x && U && out && ws && N > 0 && wino_ok(N, H, W, Cin, Cout) && ldx >= Cin && ldo >= Cout);
    UNET_CHECK_ARG(ldx % 4 == 0 && ldo % 4 == 0 && unet_aligned16(x) && unet_aligned16(out) && unet_aligned16(ws) && unet_aligned16(U));
    UNET_CHECK_ARG(!bias || unet_aligned16(bias));
    if (ws_bytes < unet_conv3x3_winograd_workspace(N, H, W, Cin, Cout)) return UNET_ENOSPC;
    const size_t T = (size_t)N * (H / 2) * (W / 2);
    float* V = (float*)ws; float* M = V + 16 * T * Cin;
    return run_wino(x, ldx, U, bias, out, ldo, N, H, W, Cin, Cout, relu, V, M, (hipStream_t)stream);
}

// dgrad: dx = conv3x3_same(dz, rot180(w)^T) with Ud = unet_winograd_weight_transform(w, mode 1)
extern "C" int unet_conv3x3_dgrad_winograd(const float* dz, int lddz, const float* Ud, float* dx, int lddx,
        int N, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream) {
    UNET_CHECK_ARG(dz && Ud && dx && ws && N > 0 && wino_ok(N, H, W, Cout, Cin) && lddz >= Cout && lddx >= Cin);
    UNET_CHECK_ARG(lddz % 4 == 0 && lddx % 4 == 0 && unet_aligned16(dz) && unet_aligned16(dx) && unet_aligned16(ws) && unet_aligned16(Ud));
    if (ws_bytes < unet_conv3x3_winograd_workspace(N, H, W, Cin, Cout)) return UNET_ENOSPC;
    const size_t T = (size_t)N * (H / 2) * (W / 2);
    float* V = (float*)ws; float* M = V + 16 * T * Cout;
    return run_wino(dz, lddz, Ud, nullptr, dx, lddx, N, H, W, Cout, Cin, 0, V, M, (hipStream_t)stream);
}
