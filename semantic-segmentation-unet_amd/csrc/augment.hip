// Augmentation stage on the device (reference UNet/augment.py).  HBM-bound image kernels, batched
// over N images [N][H][W][C] fp32 with per-image parameters in device arrays (no host round trip between stages):
//   warp      UNet/augment.py:160-174: skimage.transform.rotate / warp, order 1, mode='reflect' = output pixel (r, c) samples
//             the input at M.(c, r, 1) (float32 arithmetic in the reference's operation order, no FMA contraction), bilinear,
//             taps mirrored without repeating the edge sample; optional left-right / up-down flips of the result and
//             round-half-even of the values (the mask path, :152-155);
//   blur      :124-136: scipy.ndimage.gaussian_filter(sigma, mode='reflect') = separable, radius int(4 sigma + 0.5), along
//             H, W AND the channel axis, edge sample repeated at the boundary, double sums;
//   minmax    per-image range for the noise / intensity scales (:113-116,138-139);
//   noise     :113-122 and intensity :138-150 as one pass: img += sigma_n * field + delta_n.
// Parity: oracle/augment_numpy.py restates the reference and is pinned to it by tests/golden/augment_ref.npz (made by running
// the reference itself); the kernels are tested against both.
#include "common.h"

namespace {

__device__ __forceinline__ int mirror_index(int i, int n) {          // skimage 'reflect': period 2(n-1)
    if (n == 1) return 0;
    const int cmax = n - 1;
    const int a = i < 0 ? -i : i;
    const int q = a / cmax, r = a - q * cmax;
    return (q & 1) ? cmax - r : r;
}

__device__ __forceinline__ int symmetric_index(int i, int n) {        // scipy 'reflect': d c b a | a b c d, period 2n
    int m = i % (2 * n);
    if (m < 0) m += 2 * n;
    return m >= n ? 2 * n - 1 - m : m;
}

// one thread per output pixel, all channels
__global__ __launch_bounds__(256) void warp_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int H, int W,
                                                   int C, const float* __restrict__ mats, const int* __restrict__ flips, int do_round) {
    // HIP contracts a*b+c into FMA by default (also through __fmul_rn / __fadd_rn); the reference rounds every product and sum
    // separately and a 1-ulp coordinate is ~1e-2 of image value on a noisy image: NOFMA pins each product in a register
#define NOFMA(x) asm volatile("" : "+v"(x))
    const long total = (long)N * H * W, stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        long t = i; const int oc = (int)(t % W); t /= W; const int orow = (int)(t % H); const int n = (int)(t / H);
        const int fl = flips ? flips[n] : 0;
        // the flips act on the warped image: output (orow, oc) shows warped (tr, tc)
        const int tr = (fl & 2) ? H - 1 - orow : orow, tc = (fl & 1) ? W - 1 - oc : oc;
        const float* m = mats + 6 * n;
        const float fc = (float)tc, fr = (float)tr;
        float c0 = m[0] * fc, c1 = m[1] * fr, r0 = m[3] * fc, r1 = m[4] * fr;
        NOFMA(c0); NOFMA(c1); NOFMA(r0); NOFMA(r1);
        float cs = c0 + c1, rs = r0 + r1;
        NOFMA(cs); NOFMA(rs);
        const float c = cs + m[2], r = rs + m[5];
        const float minr = floorf(r), minc = floorf(c);
        const float dr = r - minr, dc = c - minc;
        const int i0 = mirror_index((int)minr, H), i1 = mirror_index((int)ceilf(r), H);
        const int j0 = mirror_index((int)minc, W), j1 = mirror_index((int)ceilf(c), W);
        const float* base = src + (size_t)n * H * W * C;
        const float* ptl = base + ((size_t)i0 * W + j0) * C; const float* ptr_ = base + ((size_t)i0 * W + j1) * C;
        const float* pbl = base + ((size_t)i1 * W + j0) * C; const float* pbr = base + ((size_t)i1 * W + j1) * C;
        float* o = dst + (size_t)i * C;
        const float wc0 = 1.f - dc, wr0 = 1.f - dr;
        for (int k = 0; k < C; ++k) {
            float a0 = wc0 * ptl[k], a1 = dc * ptr_[k], b0 = wc0 * pbl[k], b1 = dc * pbr[k];
            NOFMA(a0); NOFMA(a1); NOFMA(b0); NOFMA(b1);
            float top = a0 + a1, bot = b0 + b1;
            NOFMA(top); NOFMA(bot);
            float v0 = wr0 * top, v1 = dr * bot;
            NOFMA(v0); NOFMA(v1);
            float v = v0 + v1;
            if (do_round) v = rintf(v);                               // np.round: half to even
            o[k] = v;
        }
    }
#undef NOFMA
}

// one gaussian pass along `axis` (0 = H, 1 = W, 2 = C); element strides are those of [N][H][W][C]
__global__ __launch_bounds__(256) void blur_pass_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int H, int W,
                                                        int C, int axis, const float* __restrict__ sigmas) {
    const long per = (long)H * W * C, total = (long)N * per, stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const int n = (int)(i / per);
        const float sg = sigmas[n];
        if (!(sg > 0.f)) { dst[i] = src[i]; continue; }
        long t = i - (long)n * per; const int k = (int)(t % C); t /= C; const int x = (int)(t % W); const int y = (int)(t / W);
        const int len = axis == 0 ? H : (axis == 1 ? W : C);
        const int pos = axis == 0 ? y : (axis == 1 ? x : k);
        const long step = axis == 0 ? (long)W * C : (axis == 1 ? C : 1);
        const float* line = src + (i - (long)pos * step);
        const double sd = (double)sg;
        const int radius = (int)(4.0 * sd + 0.5);
        const double e = -0.5 / (sd * sd);
        double acc = 0.0, wsum = 0.0;
        for (int d = -radius; d <= radius; ++d) {
            const double wgt = exp(e * (double)(d * d));
            wsum += wgt;
            acc += wgt * (double)line[(long)symmetric_index(pos + d, len) * step];
        }
        dst[i] = (float)(acc / wsum);
    }
}

// per-image min / max: one block per (image, slice), fixed combine order
__global__ __launch_bounds__(256) void minmax_kernel(const float* __restrict__ img, long per, int slices, float* __restrict__ part) {
    __shared__ float smn[256], smx[256];
    const int n = blockIdx.x / slices, s = blockIdx.x % slices;
    const long chunk = (per + slices - 1) / slices, a = (long)s * chunk;
    long b = a + chunk; if (b > per) b = per;
    float mn = __builtin_inff(), mx = -__builtin_inff();
    for (long i = a + threadIdx.x; i < b; i += 256) { const float v = img[(size_t)n * per + i]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
    smn[threadIdx.x] = mn; smx[threadIdx.x] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { smn[threadIdx.x] = fminf(smn[threadIdx.x], smn[threadIdx.x + o]); smx[threadIdx.x] = fmaxf(smx[threadIdx.x], smx[threadIdx.x + o]); }
        __syncthreads();
    }
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = smn[0]; part[2 * blockIdx.x + 1] = smx[0]; }
}
__global__ void minmax_final_kernel(const float* __restrict__ part, int slices, float* __restrict__ out, int N) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float mn = __builtin_inff(), mx = -__builtin_inff();
    for (int s = 0; s < slices; ++s) { mn = fminf(mn, part[2 * (n * slices + s)]); mx = fmaxf(mx, part[2 * (n * slices + s) + 1]); }
    out[2 * n] = mn; out[2 * n + 1] = mx;
}

// img[n] += scale[n] * range[n] * field + ... : value = img + field * (coef_noise[n] * range) + coef_add[n] * range,
// range = max - min of image n (from minmax); either coefficient array may be NULL
__global__ __launch_bounds__(256) void noise_intensity_kernel(float* __restrict__ img, const float* __restrict__ field, long per, int N,
                                                              const float* __restrict__ minmax, const float* __restrict__ coef_noise,
                                                              const float* __restrict__ coef_add) {
    const long total = (long)N * per, stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const int n = (int)(i / per);
        const double rng = (double)minmax[2 * n + 1] - (double)minmax[2 * n];
        double v = (double)img[i];
        if (coef_noise && field) v += (double)field[i] * ((double)coef_noise[n] * rng);
        if (coef_add) v += (double)coef_add[n] * rng;
        img[i] = (float)v;
    }
}

// ---- per-image, per-channel z-score (UNet/imagereader.py:33-49) fused with the NHWC -> NCHW transpose the network input wants --
// stats: one block per (image, channel, slice): double sum / sum of squares; final: mean, std (population); apply:
// out[n][c][y][x] = (in[n][y][x][c] - mean) / (std > 1 ? std : 1)       (the reference only subtracts the mean when std <= 1)
__global__ __launch_bounds__(256) void zscore_stats_kernel(const float* __restrict__ img, int C, long hw, int slices, double* __restrict__ part) {
    __shared__ double s1[256], s2[256];
    const int nc = blockIdx.x / slices, s = blockIdx.x % slices;
    const int n = nc / C, c = nc % C;
    const long chunk = (hw + slices - 1) / slices, a = (long)s * chunk;
    long b = a + chunk; if (b > hw) b = hw;
    double t1 = 0.0, t2 = 0.0;
    for (long i = a + threadIdx.x; i < b; i += 256) { const double v = (double)img[((size_t)n * hw + i) * C + c]; t1 += v; t2 += v * v; }
    s1[threadIdx.x] = t1; s2[threadIdx.x] = t2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { s1[threadIdx.x] += s1[threadIdx.x + o]; s2[threadIdx.x] += s2[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = s1[0]; part[2 * blockIdx.x + 1] = s2[0]; }
}
__global__ void zscore_final_kernel(const double* __restrict__ part, int slices, long hw, float* __restrict__ coef, int NC) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NC) return;
    double t1 = 0.0, t2 = 0.0;
    for (int s = 0; s < slices; ++s) { t1 += part[2 * (i * slices + s)]; t2 += part[2 * (i * slices + s) + 1]; }
    const double mean = t1 / (double)hw;
    double var = t2 / (double)hw - mean * mean; if (var < 0.0) var = 0.0;
    const double sd = sqrt(var);
    coef[2 * i] = (float)mean; coef[2 * i + 1] = sd <= 1.0 ? 1.f : (float)(1.0 / sd);
}
__global__ __launch_bounds__(256) void zscore_apply_kernel(const float* __restrict__ img, float* __restrict__ out, int N, int C, long hw,
                                                           const float* __restrict__ coef) {
    const long total = (long)N * C * hw, stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {         // i indexes the NCHW output
        const long px = i % hw; const int nc = (int)(i / hw); const int n = nc / C, c = nc % C;
        out[i] = (img[((size_t)n * hw + px) * C + c] - coef[2 * nc]) * coef[2 * nc + 1];
    }
}

int grid_of(long total, int cap) { long b = (total + 255) / 256; if (b > cap) b = cap; if (b < 1) b = 1; return (int)b; }
constexpr int kSlices = 64;

}  // namespace

extern "C" int unet_augment_warp(const float* src, float* dst, int N, int H, int W, int C, const float* mats,
                                 const int* flips, int round_output, void* stream) {
    UNET_CHECK_ARG(src && dst && mats && src != dst && N > 0 && H > 0 && W > 0 && C > 0);
    warp_kernel<<<grid_of((long)N * H * W, 8192), 256, 0, (hipStream_t)stream>>>(src, dst, N, H, W, C, mats, flips, round_output);
    return UNET_LAUNCH_STATUS();
}

extern "C" int unet_augment_gaussian_blur(float* img, float* tmp, int N, int H, int W, int C, const float* sigmas, void* stream) {
    UNET_CHECK_ARG(img && tmp && img != tmp && sigmas && N > 0 && H > 0 && W > 0 && C > 0);
    hipStream_t st = (hipStream_t)stream;
    const int g = grid_of((long)N * H * W * C, 8192);
    blur_pass_kernel<<<g, 256, 0, st>>>(img, tmp, N, H, W, C, 0, sigmas);
    blur_pass_kernel<<<g, 256, 0, st>>>(tmp, img, N, H, W, C, 1, sigmas);
    blur_pass_kernel<<<g, 256, 0, st>>>(img, tmp, N, H, W, C, 2, sigmas);
    int rc = UNET_LAUNCH_STATUS(); if (rc) return rc;
    return (int)hipMemcpyAsync(img, tmp, (size_t)N * H * W * C * sizeof(float), hipMemcpyDeviceToDevice, st);
}

extern "C" size_t unet_augment_minmax_workspace(int N) { return (size_t)N * kSlices * 2 * sizeof(float); }

extern "C" int unet_augment_minmax(const float* img, int N, long elems_per_image, float* minmax, void* ws, size_t ws_bytes, void* stream) {
    UNET_CHECK_ARG(img && minmax && ws && N > 0 && elems_per_image > 0);
    if (ws_bytes < unet_augment_minmax_workspace(N)) return UNET_ENOSPC;
    hipStream_t st = (hipStream_t)stream;
    minmax_kernel<<<N * kSlices, 256, 0, st>>>(img, elems_per_image, kSlices, (float*)ws);
    minmax_final_kernel<<<unet_cdiv(N, 64), 64, 0, st>>>((const float*)ws, kSlices, minmax, N);
    return UNET_LAUNCH_STATUS();
}

extern "C" int unet_augment_noise_intensity(float* img, const float* field, int N, long elems_per_image, const float* minmax,
                                            const float* coef_noise, const float* coef_add, void* stream) {
    UNET_CHECK_ARG(img && minmax && N > 0 && elems_per_image > 0 && (coef_noise || coef_add) && (!coef_noise || field));
    noise_intensity_kernel<<<grid_of((long)N * elems_per_image, 8192), 256, 0, (hipStream_t)stream>>>(img, field, elems_per_image, N, minmax, coef_noise, coef_add);
    return UNET_LAUNCH_STATUS();
}

extern "C" size_t unet_zscore_workspace(int N, int C) { return (size_t)N * C * (kSlices * 2 * sizeof(double) + 2 * sizeof(float)); }

// out [N][C][H][W] = per-(image, channel) z-score of img [N][H][W][C] (reader contract, UNet/imagereader.py:33-49,298-301)
extern "C" int unet_zscore_nhwc_to_nchw(const float* img, float* out, int N, int H, int W, int C, void* ws, size_t ws_bytes, void* stream) {
    UNET_CHECK_ARG(img && out && ws && img != out && N > 0 && H > 0 && W > 0 && C > 0);
    if (ws_bytes < unet_zscore_workspace(N, C)) return UNET_ENOSPC;
    hipStream_t st = (hipStream_t)stream;
    double* part = (double*)ws;
    float* coef = (float*)(part + (size_t)N * C * kSlices * 2);
    const long hw = (long)H * W;
    zscore_stats_kernel<<<N * C * kSlices, 256, 0, st>>>(img, C, hw, kSlices, part);
    zscore_final_kernel<<<unet_cdiv(N * C, 64), 64, 0, st>>>(part, kSlices, hw, coef, N * C);
    zscore_apply_kernel<<<grid_of((long)N * C * hw, 8192), 256, 0, st>>>(img, out, N, C, hw, coef);
    return UNET_LAUNCH_STATUS();
}
