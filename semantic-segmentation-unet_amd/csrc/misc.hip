// Remaining HBM-bound ops of the train / inference step:
//   2x2 max-pool fwd/bwd (UNet/model.py:50-53), dropout (UNet/model.py:60-63), softmax + categorical cross-entropy
//   + accuracy + d(loss)/d(logits) (UNet/model.py:142,77,211-215,226), argmax (UNet/inference.py:107,166),
//   Keras-Adam over the flat parameter buffer (UNet/model.py:79,223), NCHW -> NHWC input permute.
#include "common.h"

namespace {

template <int VEC> __device__ __forceinline__ void vload(float (&v)[VEC], const float* p) {
    if constexpr (VEC == 4) { const f32x4 t = *reinterpret_cast<const f32x4*>(p); v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; }
    else v[0] = *p;
}
template <int VEC> __device__ __forceinline__ void vstore(float* p, const float (&v)[VEC]) {
    if constexpr (VEC == 4) { f32x4 t = {v[0], v[1], v[2], v[3]}; *reinterpret_cast<f32x4*>(p) = t; }
    else *p = v[0];
}

// the same on a tensor stored as bf16 (B16: `p` addresses 2-byte elements; VEC == 4 only): level 4 of the bf16 mode keeps the tensors
// around its dropout / unfused pool as bf16 like every other level
template <int VEC, int B16> __device__ __forceinline__ void tload(float (&v)[VEC], const float* base, size_t elem) {
    if constexpr (B16) {
        const uint2 h = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + elem);
        v[0] = __builtin_bit_cast(float, h.x << 16); v[1] = __builtin_bit_cast(float, h.x & 0xffff0000u);
        v[2] = __builtin_bit_cast(float, h.y << 16); v[3] = __builtin_bit_cast(float, h.y & 0xffff0000u);
    } else vload<VEC>(v, base + elem);
}
template <int VEC, int B16> __device__ __forceinline__ void tstore(float* base, size_t elem, const float (&v)[VEC]) {
    if constexpr (B16) {
        uint2 h;
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h.x) : "v"(v[0]), "v"(v[1]));
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h.y) : "v"(v[2]), "v"(v[3]));
        *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(base) + elem) = h;
    } else vstore<VEC>(base + elem, v);
}

// ---- max-pool 2x2 stride 2; ties -> first max in row-major window order (a*2+b) --------------------------------
template <int VEC, int B16>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy,
        uint8_t* __restrict__ idx, int N, int H, int W, int C) {
    const int H2 = H / 2, W2 = W / 2, nq = C / VEC;
    const long total = (long)N * H2 * W2 * nq, stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        long t = i; const int cq = (int)(t % nq); t /= nq;
        const int ox = (int)(t % W2); t /= W2; const int oy = (int)(t % H2); const int n = (int)(t / H2);
        const long opix = ((long)n * H2 + oy) * W2 + ox;
        float best[VEC]; uint8_t bi[VEC];
#pragma unroll
        for (int pos = 0; pos < 4; ++pos) {
            float v[VEC];
            tload<VEC, B16>(v, x, ((size_t)((long)n * H + 2 * oy + (pos >> 1)) * W + 2 * ox + (pos & 1)) * ldx + cq * VEC);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                if (pos == 0 || v[e] > best[e]) { best[e] = v[e]; bi[e] = (uint8_t)pos; }
            }
        }
        tstore<VEC, B16>(y, (size_t)opix * ldy + cq * VEC, best);
#pragma unroll
        for (int e = 0; e < VEC; ++e) idx[(size_t)opix * C + cq * VEC + e] = bi[e];
    }
}

template <int VEC, int B16>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, int lddy, const uint8_t* __restrict__ idx,
        float* __restrict__ dx, int lddx, int N, int H, int W, int C, int accumulate) {
    const int H2 = H / 2, W2 = W / 2, nq = C / VEC;
    const long total = (long)N * H2 * W2 * nq, stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        long t = i; const int cq = (int)(t % nq); t /= nq;
        const int ox = (int)(t % W2); t /= W2; const int oy = (int)(t % H2); const int n = (int)(t / H2);
        const long opix = ((long)n * H2 + oy) * W2 + ox;
        float g[VEC]; tload<VEC, B16>(g, dy, (size_t)opix * lddy + cq * VEC);
        uint8_t bi[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) bi[e] = idx[(size_t)opix * C + cq * VEC + e];
#pragma unroll
        for (int pos = 0; pos < 4; ++pos) {
            const size_t dst = ((size_t)((long)n * H + 2 * oy + (pos >> 1)) * W + 2 * ox + (pos & 1)) * lddx + cq * VEC;
            float v[VEC];
            if (accumulate) tload<VEC, B16>(v, dx, dst);
            else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[e] = 0.f;
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) if (bi[e] == pos) v[e] += g[e];
            tstore<VEC, B16>(dx, dst, v);
        }
    }
}

// ---- dropout: out = x * keep / (1 - rate); keep from an explicit mask (tests) or the counter hash --------------
template <int VEC, int B16>
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, int ldx, float* __restrict__ out, int ldo,
        long P, int C, const uint8_t* __restrict__ mask, uint32_t seed, float rate, float scale) {
    const int nq = C / VEC;
    const long total = P * nq, stride = (long)gridDim.x * 256;
    const uint32_t thr = (uint32_t)((double)rate * 4294967296.0 > 4294967295.0 ? 4294967295.0 : (double)rate * 4294967296.0);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const long pix = i / nq; const int c0 = (int)(i - pix * nq) * VEC;
        float v[VEC]; tload<VEC, B16>(v, x, (size_t)pix * ldx + c0);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const uint64_t el = (uint64_t)pix * C + c0 + e;
            const bool keep = mask ? (mask[el] != 0) : (unet_hash32(seed, el) >= thr);
            v[e] = keep ? v[e] * scale : 0.f;
        }
        tstore<VEC, B16>(out, (size_t)pix * ldo + c0, v);
    }
}

// ---- softmax + CE (from logits) + accuracy + dlogits ------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_ce_kernel(const float* __restrict__ z, int ldz, const int* __restrict__ labels,
        float* __restrict__ prob, float* __restrict__ dz, int lddz, long P, int K, float label_smoothing, float grad_scale,
        float clip_eps, double* __restrict__ part_loss, unsigned long long* __restrict__ part_correct) {
    __shared__ double sL[256];
    __shared__ unsigned int sC[256];
    double lsum = 0.0; unsigned int csum = 0;
    const long stride = (long)gridDim.x * 256;
    for (long pix = (long)blockIdx.x * 256 + threadIdx.x; pix < P; pix += stride) {
        const float* zp = z + (size_t)pix * ldz;
        float m = zp[0]; int am = 0;
        for (int k = 1; k < K; ++k) { const float v = zp[k]; if (v > m) { m = v; am = k; } }
        float se = 0.f;
        for (int k = 0; k < K; ++k) se += expf(zp[k] - m);
        const float lse = logf(se), inv = 1.f / se;
        float ysum = 0.f, ell = 0.f; int al = 0; int lbest = 0;
        if (labels) {
            lbest = labels[(size_t)pix * K];
            for (int k = 0; k < K; ++k) {
                const int li = labels[(size_t)pix * K + k];
                if (li > lbest) { lbest = li; al = k; }
                float y = (float)li;
                if (label_smoothing > 0.f) y = y * (1.f - label_smoothing) + label_smoothing / (float)K;
                ysum += y;
                if (clip_eps > 0.f) {
                    // Keras backend categorical_crossentropy on probabilities: renormalise (sum p = 1 here), clip to
                    // [eps, 1-eps], -sum y log q; the clip passes no gradient outside its range
                    const float q = expf(zp[k] - m) * inv;
                    ell -= y * logf(fminf(fmaxf(q, clip_eps), 1.f - clip_eps));
                } else {
                    ell -= y * ((zp[k] - m) - lse);
                }
            }
        }
        float gdot = 0.f;                        // clip path: sum_j g_j p_j with g_j = -y_j / q_j inside the clip range
        if (dz && clip_eps > 0.f) {
            for (int k = 0; k < K; ++k) {
                const float pk = expf(zp[k] - m) * inv;
                float y = (float)labels[(size_t)pix * K + k];
                if (label_smoothing > 0.f) y = y * (1.f - label_smoothing) + label_smoothing / (float)K;
                if (pk >= clip_eps && pk <= 1.f - clip_eps) gdot -= y;          // g_j p_j = -y_j
            }
        }
        for (int k = 0; k < K; ++k) {
            const float pk = expf(zp[k] - m) * inv;
            if (prob) prob[(size_t)pix * K + k] = pk;
            if (dz) {
                float y = (float)labels[(size_t)pix * K + k];
                if (label_smoothing > 0.f) y = y * (1.f - label_smoothing) + label_smoothing / (float)K;
                if (clip_eps > 0.f) {
                    const float gp = (pk >= clip_eps && pk <= 1.f - clip_eps) ? -y : 0.f;     // p_k g_k
                    dz[(size_t)pix * lddz + k] = (gp - pk * gdot) * grad_scale;
                } else {
                    dz[(size_t)pix * lddz + k] = (pk * ysum - y) * grad_scale;
                }
            }
        }
        lsum += (double)ell; csum += (labels && al == am) ? 1u : 0u;
    }
    sL[threadIdx.x] = lsum; sC[threadIdx.x] = csum;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) { sL[threadIdx.x] += sL[threadIdx.x + s]; sC[threadIdx.x] += sC[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { part_loss[blockIdx.x] = sL[0]; part_correct[blockIdx.x] = sC[0]; }
}

__global__ __launch_bounds__(64) void softmax_ce_finalize_kernel(const double* part_loss, const unsigned long long* part_correct, int nblk,
                                           float loss_scale, float* loss_out, float* correct_out) {
    double s = 0.0; unsigned long long c = 0;
    for (int k = threadIdx.x; k < nblk; k += 64) { s += part_loss[k]; c += part_correct[k]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); c += __shfl_xor(c, o); }
    if (threadIdx.x == 0) {
        if (loss_out) loss_out[0] = (float)(s * (double)loss_scale);
        if (correct_out) correct_out[0] = (float)c;
    }
}

__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ p, int ldp, int* __restrict__ out, long P, int K) {
    const long stride = (long)gridDim.x * 256;
    for (long pix = (long)blockIdx.x * 256 + threadIdx.x; pix < P; pix += stride) {
        const float* zp = p + (size_t)pix * ldp;
        float m = zp[0]; int am = 0;
        for (int k = 1; k < K; ++k) { const float v = zp[k]; if (v > m) { m = v; am = k; } }
        out[pix] = am;
    }
}

// ---- Keras Adam: m += (g-m)(1-b1); v += (g*g-v)(1-b2); theta -= alpha*m/(sqrt(v)+eps), alpha from the host ----
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ theta, const float* __restrict__ g, float* __restrict__ m,
        float* __restrict__ v, long n4, float alpha, float one_minus_b1, float one_minus_b2, float eps) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        f32x4 t = reinterpret_cast<f32x4*>(theta)[i];
        const f32x4 gg = reinterpret_cast<const f32x4*>(g)[i];
        f32x4 mm = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            mm[e] = mm[e] + (gg[e] - mm[e]) * one_minus_b1;
            vv[e] = vv[e] + (gg[e] * gg[e] - vv[e]) * one_minus_b2;
            t[e] = t[e] - alpha * mm[e] / (sqrtf(vv[e]) + eps);
        }
        reinterpret_cast<f32x4*>(theta)[i] = t;
        reinterpret_cast<f32x4*>(m)[i] = mm; reinterpret_cast<f32x4*>(v)[i] = vv;
    }
}

__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int C, long HW) {
    const long total = (long)N * C * HW, stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        long t = i; const int c = (int)(t % C); t /= C; const long hw = t % HW; const long n = t / HW;
        y[i] = x[((size_t)n * C + c) * HW + hw];
    }
}

// ---- eval-mode backward pieces (input-gradient probe of UNet.estimate_radius, reference UNet/model.py:165-202) ----
// softmax backward: dz_k = p_k * (g_k - sum_j g_j p_j)
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ p, const float* __restrict__ g,
                                                          float* __restrict__ dz, int lddz, long P, int K) {
    const long stride = (long)gridDim.x * 256;
    for (long pix = (long)blockIdx.x * 256 + threadIdx.x; pix < P; pix += stride) {
        float dot = 0.f;
        for (int k = 0; k < K; ++k) dot += g[(size_t)pix * K + k] * p[(size_t)pix * K + k];
        for (int k = 0; k < K; ++k) dz[(size_t)pix * lddz + k] = p[(size_t)pix * K + k] * (g[(size_t)pix * K + k] - dot);
    }
}

// BatchNorm with fixed (moving) statistics is an affine map: dz = dy * scale, times the ReLU mask of the layer below
template <int VEC>
__global__ __launch_bounds__(256) void bn_eval_bwd_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ r, int ldr,
        const float* __restrict__ scale, float* __restrict__ dz, int lddz, long P, int C, int relu) {
    const int nq = C / VEC;
    const long total = P * nq, stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const long pix = i / nq; const int c0 = (int)(i - pix * nq) * VEC;
        float g[VEC], v[VEC], a[VEC];
        vload<VEC>(g, dy + (size_t)pix * lddy + c0); vload<VEC>(v, r + (size_t)pix * ldr + c0); vload<VEC>(a, scale + c0);
#pragma unroll
        for (int e = 0; e < VEC; ++e) g[e] = (relu && !(v[e] > 0.f)) ? 0.f : g[e] * a[e];
        vstore<VEC>(dz + (size_t)pix * lddz + c0, g);
    }
}

// data gradient of the first 3x3 layer (tiny Cin): dx[p][ci] = sum_{tap,co} dz[p - shift(tap)][co] * w[tap][ci][co]
__global__ __launch_bounds__(256) void conv3x3_direct_dgrad_kernel(const float* __restrict__ dz, int lddz, const float* __restrict__ w,
        float* __restrict__ dx, int lddx, int N, int H, int W, int Cin, int Cout) {
    const long total = (long)N * H * W * Cin, stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        long t = i; const int ci = (int)(t % Cin); t /= Cin;
        const int xx = (int)(t % W); t /= W; const int y = (int)(t % H); const int n = (int)(t / H);
        float acc = 0.f;
        for (int tap = 0; tap < 9; ++tap) {
            const int gy = y - (tap / 3 - 1), gx = xx - (tap % 3 - 1);
            if (gy < 0 || gy >= H || gx < 0 || gx >= W) continue;
            const float* gp = dz + ((size_t)(n * H + gy) * W + gx) * lddz;
            const float* wp = w + ((size_t)tap * Cin + ci) * Cout;
            for (int co = 0; co < Cout; ++co) acc += gp[co] * wp[co];
        }
        dx[((size_t)(n * H + y) * W + xx) * lddx + ci] = acc;
    }
}

int grid_for(long total, int cap) { long b = (total + 255) / 256; if (b > cap) b = cap; if (b < 1) b = 1; return (int)b; }

}  // namespace

// t_bf16 (these three): the activation tensors are stored as bf16 (leading dimensions in elements, C % 4 == 0); idx / mask unchanged
extern "C" int unet_maxpool2x2_fwd(const void* xv, int ldx, void* yv, int ldy, uint8_t* idx, int N, int H, int W, int C, int t_bf16, void* stream) {
    const float* x = (const float*)xv; float* y = (float*)yv;
    UNET_CHECK_ARG(x && y && idx && N > 0 && H > 0 && W > 0 && C > 0 && H % 2 == 0 && W % 2 == 0 && ldx >= C && ldy >= C);
    const bool v4 = C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && unet_aligned16(x) && unet_aligned16(y);
    UNET_CHECK_ARG(!t_bf16 || v4);
    const long total = (long)N * (H / 2) * (W / 2) * (v4 ? C / 4 : C);
    if (t_bf16)  maxpool_fwd_kernel<4, 1><<<grid_for(total, 8192), 256, 0, (hipStream_t)stream>>>(x, ldx, y, ldy, idx, N, H, W, C);
    else if (v4) maxpool_fwd_kernel<4, 0><<<grid_for(total, 8192), 256, 0, (hipStream_t)stream>>>(x, ldx, y, ldy, idx, N, H, W, C);
    else         maxpool_fwd_kernel<1, 0><<<grid_for(total, 8192), 256, 0, (hipStream_t)stream>>>(x, ldx, y, ldy, idx, N, H, W, C);
    return UNET_LAUNCH_STATUS();
}

// H, W are the dims of dx (the pool input); dy is [N, H/2, W/2, C].
extern "C" int unet_maxpool2x2_bwd(const void* dyv, int lddy, const uint8_t* idx, void* dxv, int lddx,
                                   int N, int H, int W, int C, int accumulate, int t_bf16, void* stream) {
    const float* dy = (const float*)dyv; float* dx = (float*)dxv;
    UNET_CHECK_ARG(dy && dx && idx && N > 0 && H > 0 && W > 0 && C > 0 && H % 2 == 0 && W % 2 == 0 && lddx >= C && lddy >= C);
    const bool v4 = C % 4 == 0 && lddx % 4 == 0 && lddy % 4 == 0 && unet_aligned16(dx) && unet_aligned16(dy);
    UNET_CHECK_ARG(!t_bf16 || v4);
    const long total = (long)N * (H / 2) * (W / 2) * (v4 ? C / 4 : C);
    if (t_bf16)  maxpool_bwd_kernel<4, 1><<<grid_for(total, 8192), 256, 0, (hipStream_t)stream>>>(dy, lddy, idx, dx, lddx, N, H, W, C, accumulate);
    else if (v4) maxpool_bwd_kernel<4, 0><<<grid_for(total, 8192), 256, 0, (hipStream_t)stream>>>(dy, lddy, idx, dx, lddx, N, H, W, C, accumulate);
    else         maxpool_bwd_kernel<1, 0><<<grid_for(total, 8192), 256, 0, (hipStream_t)stream>>>(dy, lddy, idx, dx, lddx, N, H, W, C, accumulate);
    return UNET_LAUNCH_STATUS();
}

extern "C" int unet_dropout(const void* xv, int ldx, void* outv, int ldo, long P, int C, const uint8_t* mask,
                            uint32_t seed, float rate, int t_bf16, void* stream) {
    const float* x = (const float*)xv; float* out = (float*)outv;
    UNET_CHECK_ARG(x && out && P > 0 && C > 0 && ldx >= C && ldo >= C && rate >= 0.f && rate < 1.f);
    const float scale = 1.f / (1.f - rate);
    const bool v4 = C % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && unet_aligned16(x) && unet_aligned16(out);
    UNET_CHECK_ARG(!t_bf16 || v4);
    const long total = P * (v4 ? C / 4 : C);
    if (t_bf16)  dropout_kernel<4, 1><<<grid_for(total, 8192), 256, 0, (hipStream_t)stream>>>(x, ldx, out, ldo, P, C, mask, seed, rate, scale);
    else if (v4) dropout_kernel<4, 0><<<grid_for(total, 8192), 256, 0, (hipStream_t)stream>>>(x, ldx, out, ldo, P, C, mask, seed, rate, scale);
    else         dropout_kernel<1, 0><<<grid_for(total, 8192), 256, 0, (hipStream_t)stream>>>(x, ldx, out, ldo, P, C, mask, seed, rate, scale);
    return UNET_LAUNCH_STATUS();
}

extern "C" size_t unet_softmax_ce_workspace(long P) { return (size_t)grid_for(P, 1024) * 16; }

// prob, dlogits, labels, loss_out, correct_out may each be null (inference: prob only).
// ce_clip_eps == 0: cross-entropy from the softmax's logits (graph-mode Keras sees the Softmax op and calls
// softmax_cross_entropy_with_logits); ce_clip_eps > 0 (Keras epsilon 1e-7): the probability path of
// keras.backend.categorical_crossentropy -- clip to [eps, 1-eps], -sum y log q, no gradient outside the clip range.
extern "C" int unet_softmax_ce(const float* logits, int ldz, const int* labels_onehot, float* prob, float* dlogits, int lddz,
        long P, int K, float label_smoothing, float loss_scale, float grad_scale, float ce_clip_eps, float* loss_out,
        float* correct_out, void* ws, size_t ws_bytes, void* stream) {
    UNET_CHECK_ARG(logits && ws && P > 0 && K > 0 && ldz >= K && (!dlogits || (labels_onehot && lddz >= K)));
    UNET_CHECK_ARG(ce_clip_eps >= 0.f && ce_clip_eps < 0.5f);
    UNET_CHECK_ARG((!loss_out && !correct_out) || labels_onehot);
    const int nblk = grid_for(P, 1024);
    if (ws_bytes < unet_softmax_ce_workspace(P)) return UNET_ENOSPC;
    double* pl = (double*)ws;
    unsigned long long* pc = (unsigned long long*)((char*)ws + (size_t)nblk * 8);
    softmax_ce_kernel<<<nblk, 256, 0, (hipStream_t)stream>>>(logits, ldz, labels_onehot, prob, dlogits, lddz, P, K,
                                                             label_smoothing, grad_scale, ce_clip_eps, pl, pc);
    int rc = UNET_LAUNCH_STATUS(); if (rc) return rc;
    if (loss_out || correct_out) {
        softmax_ce_finalize_kernel<<<1, 64, 0, (hipStream_t)stream>>>(pl, pc, nblk, loss_scale, loss_out, correct_out);
        rc = UNET_LAUNCH_STATUS();
    }
    return rc;
}

// class map (one uint8 label per pixel) -> int32 one-hot [P][K]; `bad` counts labels >= K (the reader contract's IndexError,
// UNet/imagereader.py:302-312, raised by the caller)
__global__ __launch_bounds__(256) void onehot_kernel(const uint8_t* __restrict__ cls, int* __restrict__ onehot, long P, int K,
                                                     unsigned* __restrict__ bad) {
    const long total = P * K, stride = (long)gridDim.x * 256;
    unsigned nbad = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const long px = i / K; const int k = (int)(i - px * K);
        const int c = cls[px];
        onehot[i] = c == k ? 1 : 0;
        if (k == 0 && c >= K) ++nbad;
    }
    if (nbad && bad) atomicAdd(bad, nbad);
}

extern "C" int unet_labels_onehot(const uint8_t* classmap, int* onehot, long P, int K, unsigned* out_of_range, void* stream) {
    UNET_CHECK_ARG(classmap && onehot && P > 0 && K > 0 && K <= 256);
    onehot_kernel<<<grid_for(P * K, 4096), 256, 0, (hipStream_t)stream>>>(classmap, onehot, P, K, out_of_range);
    return UNET_LAUNCH_STATUS();
}

extern "C" int unet_argmax(const float* p, int ldp, int* out, long P, int K, void* stream) {
    UNET_CHECK_ARG(p && out && P > 0 && K > 0 && ldp >= K);
    argmax_kernel<<<grid_for(P, 4096), 256, 0, (hipStream_t)stream>>>(p, ldp, out, P, K);
    return UNET_LAUNCH_STATUS();
}

extern "C" int unet_adam_keras(float* theta, const float* grad, float* m, float* v, long n, float alpha, float beta1,
                               float beta2, float eps, void* stream) {
    UNET_CHECK_ARG(theta && grad && m && v && n > 0 && n % 4 == 0);
    UNET_CHECK_ARG(unet_aligned16(theta) && unet_aligned16(grad) && unet_aligned16(m) && unet_aligned16(v));
    adam_kernel<<<grid_for(n / 4, 4096), 256, 0, (hipStream_t)stream>>>(theta, grad, m, v, n / 4, alpha, 1.f - beta1, 1.f - beta2, eps);
    return UNET_LAUNCH_STATUS();
}

extern "C" int unet_nchw_to_nhwc(const float* x, float* y, int N, int C, int H, int W, void* stream) {
    UNET_CHECK_ARG(x && y && N > 0 && C > 0 && H > 0 && W > 0);
    nchw_to_nhwc_kernel<<<grid_for((long)N * C * H * W, 8192), 256, 0, (hipStream_t)stream>>>(x, y, N, C, (long)H * W);
    return UNET_LAUNCH_STATUS();
}

extern "C" int unet_softmax_bwd(const float* prob, const float* dprob, float* dlogits, int lddz, long P, int K, void* stream) {
    UNET_CHECK_ARG(prob && dprob && dlogits && P > 0 && K > 0 && lddz >= K);
    softmax_bwd_kernel<<<grid_for(P, 4096), 256, 0, (hipStream_t)stream>>>(prob, dprob, dlogits, lddz, P, K);
    return UNET_LAUNCH_STATUS();
}

extern "C" int unet_bn_eval_bwd(const float* dy, int lddy, const float* r, int ldr, const float* scale, float* dz, int lddz,
                                long P, int C, int relu, void* stream) {
    UNET_CHECK_ARG(dy && r && scale && dz && P > 0 && C > 0 && lddy >= C && ldr >= C && lddz >= C);
    const bool v4 = C % 4 == 0 && lddy % 4 == 0 && ldr % 4 == 0 && lddz % 4 == 0 && unet_aligned16(dy) && unet_aligned16(r) &&
                    unet_aligned16(dz) && unet_aligned16(scale);
    const long total = P * (v4 ? C / 4 : C);
    if (v4) bn_eval_bwd_kernel<4><<<grid_for(total, 8192), 256, 0, (hipStream_t)stream>>>(dy, lddy, r, ldr, scale, dz, lddz, P, C, relu);
    else    bn_eval_bwd_kernel<1><<<grid_for(total, 8192), 256, 0, (hipStream_t)stream>>>(dy, lddy, r, ldr, scale, dz, lddz, P, C, relu);
    return UNET_LAUNCH_STATUS();
}

extern "C" int unet_conv3x3_dgrad_direct(const float* dz, int lddz, const float* w, float* dx, int lddx,
                                         int N, int H, int W, int Cin, int Cout, void* stream) {
    UNET_CHECK_ARG(dz && w && dx && N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && lddz >= Cout && lddx >= Cin);
    conv3x3_direct_dgrad_kernel<<<grid_for((long)N * H * W * Cin, 8192), 256, 0, (hipStream_t)stream>>>(dz, lddz, w, dx, lddx, N, H, W, Cin, Cout);
    return UNET_LAUNCH_STATUS();
}

extern "C" int unet_hip_abi_version(void) { return 9; }       // == UNET_HIP_ABI_VERSION of include/unet_hip.h

// ---- stand-in for a collective's kernel (measurement aid; include/unet_hip.h) -----------------------------------------------------------
// RCCL's all-reduce kernels are a few dozen large workgroups that stay resident while the data crosses xGMI.  This one reproduces the
// occupancy, not the traffic: every wave sleeps until the constant-rate wall clock (hipDeviceAttributeWallClockRate) has advanced by
// `ticks` since the wave started, then leaves -- an exit condition every wave reaches.
namespace {
__global__ __launch_bounds__(512) void standin_collective_kernel(long long ticks) {
    extern __shared__ int standin_lds[];
    if (threadIdx.x == 0) reinterpret_cast<volatile int*>(standin_lds)[0] = 0;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
}  // namespace

extern "C" int unet_standin_collective(int workgroups, int lds_bytes, int microseconds, void* stream) {
    UNET_CHECK_ARG(workgroups > 0 && workgroups <= 1024 && lds_bytes >= 0 && lds_bytes <= 65536 && microseconds >= 0 && microseconds <= 100000);
    int dev = 0, khz = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000;
    const long long ticks = (long long)microseconds * khz / 1000;
    standin_collective_kernel<<<dim3((unsigned)workgroups), 512, (size_t)(lds_bytes < 4 ? 4 : lds_bytes), (hipStream_t)stream>>>(ticks);
    return UNET_LAUNCH_STATUS();
}
