// BatchNormalization(axis=channels) as used by every layer of the reference (UNet/model.py:36,47): it follows the
// ReLU, so statistics are over the post-activation tensor r.  Training mode: biased batch variance, eps inside the
// sqrt, moving stats updated with momentum (unbiased variance fed to moving_variance, Keras fused path).
//
// All kernels are HBM-bound streams over NHWC tensors with channel stride `ld`.  Thread layout: `tpp` consecutive
// lanes cover one pixel's channels (VEC=4: one float4 quad per lane; VEC=1: one channel per lane for the narrow
// class-map tensors), 256/tpp pixels in flight per block.  Per-channel sums are accumulated in fp64 per lane,
// combined through LDS, written as per-block partials and summed in a fixed order (bit-stable run to run).
#include "common.h"
#include <stdlib.h>

namespace {

typedef int i32x4v __attribute__((ext_vector_type(4)));
constexpr int kApplyNT = (UNET_NT & UNET_NT_LDAPPLY) != 0;

template <class T, int NT> __device__ __forceinline__ T bn_ld(const T* p) {
    if constexpr (NT != 0) return __builtin_nontemporal_load(p); else return *p;
}
template <int VEC, int NT = 0> __device__ __forceinline__ void vload(float (&v)[VEC], const float* p) {
    if constexpr (VEC == 8) {
        const f32x4 t = bn_ld<f32x4, NT>(reinterpret_cast<const f32x4*>(p)), u = bn_ld<f32x4, NT>(reinterpret_cast<const f32x4*>(p + 4));
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; v[4] = u[0]; v[5] = u[1]; v[6] = u[2]; v[7] = u[3];
    } else if constexpr (VEC == 4) { const f32x4 t = bn_ld<f32x4, NT>(reinterpret_cast<const f32x4*>(p)); v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; }
    else v[0] = *p;
}
__device__ __forceinline__ float bf_lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
// load from an fp32 tensor or (in16, VEC >= 4) from a bf16 tensor of the same logical layout (element index idx).  VEC == 8 is the
// 16-byte-per-lane form for bf16 tensors (8 channels per lane; an fp32 tensor is then two 16-byte loads): 8-byte accesses run at
// 0.54-0.70 of the 16-byte rate on this memory system, which is what held the bf16-storage mode back in round 1
template <int VEC, int NT = 0> __device__ __forceinline__ void vload_dt(float (&v)[VEC], const float* base, size_t idx, int in16) {
    if constexpr (VEC == 8) {
        if (in16) {
            const unet_u32x4 t = bn_ld<unet_u32x4, NT>(reinterpret_cast<const unet_u32x4*>(reinterpret_cast<const uint16_t*>(base) + idx));
            v[0] = bf_lo(t[0]); v[1] = bf_hi(t[0]); v[2] = bf_lo(t[1]); v[3] = bf_hi(t[1]);
            v[4] = bf_lo(t[2]); v[5] = bf_hi(t[2]); v[6] = bf_lo(t[3]); v[7] = bf_hi(t[3]);
            return;
        }
    }
    if constexpr (VEC == 4) {
        if (in16) {
            const unet_u32x2 t = bn_ld<unet_u32x2, NT>(reinterpret_cast<const unet_u32x2*>(reinterpret_cast<const uint16_t*>(base) + idx));
            v[0] = bf_lo(t[0]); v[1] = bf_hi(t[0]); v[2] = bf_lo(t[1]); v[3] = bf_hi(t[1]);
            return;
        }
    }
    vload<VEC, NT>(v, base + idx);
}
template <int VEC> __device__ __forceinline__ void vstore(float* p, const float (&v)[VEC]) {
    if constexpr (VEC == 8) {
        f32x4 t = {v[0], v[1], v[2], v[3]}, u = {v[4], v[5], v[6], v[7]};
        unet_store<UNET_NT_BN>(reinterpret_cast<f32x4*>(p), t); unet_store<UNET_NT_BN>(reinterpret_cast<f32x4*>(p + 4), u);
    } else if constexpr (VEC == 4) { f32x4 t = {v[0], v[1], v[2], v[3]}; unet_store<UNET_NT_BN>(reinterpret_cast<f32x4*>(p), t); }
    else *p = v[0];
}

// store to an fp32 tensor or (out16, VEC >= 4) to a bf16 tensor of the same logical layout: element index idx, nearest-even
// rounding by the instruction the bf16 conv kernels use when they stage fp32 operands, so a bf16-stored tensor is bit-for-bit
// what those kernels would have made of the fp32 one
__device__ __forceinline__ unsigned bn_pack2(float lo, float hi) { unsigned r; asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi)); return r; }
template <int VEC> __device__ __forceinline__ void vstore_dt(float* base, size_t idx, const float (&v)[VEC], int out16) {
    if constexpr (VEC == 8) {
        if (out16) {
            const unet_u32x4 t = {bn_pack2(v[0], v[1]), bn_pack2(v[2], v[3]), bn_pack2(v[4], v[5]), bn_pack2(v[6], v[7])};
            unet_store<UNET_NT_BN>(reinterpret_cast<unet_u32x4*>(reinterpret_cast<uint16_t*>(base) + idx), t); return;
        }
    }
    if constexpr (VEC == 4) {
        if (out16) { const unet_u32x2 t = {bn_pack2(v[0], v[1]), bn_pack2(v[2], v[3])}; unet_store<UNET_NT_BN>(reinterpret_cast<unet_u32x2*>(reinterpret_cast<uint16_t*>(base) + idx), t); return; }
    }
    vstore<VEC>(base + idx, v);
}

struct Lay { int tpp, npl, q, pl, c0; bool active; };

template <int VEC> __device__ __forceinline__ Lay make_lay(int C, int tpp) {
    Lay l; l.tpp = tpp; l.npl = 256 / tpp; l.q = threadIdx.x % tpp; l.pl = threadIdx.x / tpp;
    l.c0 = l.q * VEC; l.active = l.c0 < C; return l;
}

// Optional second gradient source for the BatchNorm-backward kernels: the layer's output also went through MaxPool2D(2)
// (encoder levels, UNet/model.py:50-53,89-91), so dy(pixel) = dy_skip(pixel) + [pixel is its window's first max] * pooled_dy.
// Reading it here saves the separate pool-backward pass that would read-modify-write the whole skip gradient.
struct PoolGrad { const float* pdy; int ldp; const uint8_t* idx; int H, W, C; int p16; };

template <int VEC> __device__ __forceinline__ void add_pool_grad(float (&g)[VEC], const PoolGrad& pg, long pix, int c0) {
    if (!pg.pdy) return;
    const int x = (int)(pix % pg.W); const long t = pix / pg.W; const int y = (int)(t % pg.H); const long n = t / pg.H;
    const long opix = (n * (pg.H >> 1) + (y >> 1)) * (pg.W >> 1) + (x >> 1);
    const int pos = ((y & 1) << 1) | (x & 1);
    float q[VEC];
    vload_dt<VEC>(q, pg.pdy, (size_t)opix * pg.ldp + c0, pg.p16);
#pragma unroll
    for (int e = 0; e < VEC; ++e) if (pg.idx[(size_t)opix * pg.C + c0 + e] == pos) g[e] += q[e];
}

// block-level combine of NV per-lane fp64 vectors (each VEC wide) over the pixel lanes; result to part[v][blk][C]
template <int VEC, int NV>
__device__ __forceinline__ void block_combine(double (&acc)[NV][VEC], const Lay& l, int C, double* part, int nblk, double* sR) {
    // sR: [npl][NV][tpp*VEC]
    const int cw = l.tpp * VEC;
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int e = 0; e < VEC; ++e) sR[((size_t)l.pl * NV + v) * cw + l.c0 + e] = acc[v][e];
    __syncthreads();
    for (int i = threadIdx.x; i < NV * cw; i += 256) {
        const int v = i / cw, c = i % cw;
        if (c < C) {
            double s = 0.0;
            for (int k = 0; k < l.npl; ++k) s += sR[((size_t)k * NV + v) * cw + c];
            part[((size_t)v * nblk + blockIdx.x) * C + c] = s;
        }
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ r, int ldr, long P, int C, int tpp,
                                                       long ppb, double* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) double sRd[];
    const Lay l = make_lay<VEC>(C, tpp);
    const long p0 = (long)blockIdx.x * ppb; long p1 = p0 + ppb; if (p1 > P) p1 = P;
    double acc[2][VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { acc[0][e] = 0.0; acc[1][e] = 0.0; }
    if (l.active) {
        long pix = p0 + l.pl;
        const long st = l.npl;
        for (; pix + 3 * st < p1; pix += 4 * st) {          // 4 independent loads in flight per lane
            float v[4][VEC];
#pragma unroll
            for (int u = 0; u < 4; ++u) vload<VEC>(v[u], r + (size_t)(pix + u * st) * ldr + l.c0);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < VEC; ++e) { acc[0][e] += (double)v[u][e]; acc[1][e] += (double)v[u][e] * (double)v[u][e]; }
        }
        for (; pix < p1; pix += st) {
            float v[VEC]; vload<VEC>(v, r + (size_t)pix * ldr + l.c0);
#pragma unroll
            for (int e = 0; e < VEC; ++e) { acc[0][e] += (double)v[e]; acc[1][e] += (double)v[e] * (double)v[e]; }
        }
    }
    block_combine<VEC, 2>(acc, l, C, part, gridDim.x, sRd);
}

// The finalize kernels run one 64-lane wave per channel: lanes stride over the per-block partials, then a fixed
// xor-shuffle tree combines them (same order every run -> bit-stable).  blockDim = 64, gridDim = C.
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// 256 threads per channel: each lane adds a strided share of the partials (a handful of independent loads), then the fixed
// xor-shuffle tree per wave and the four wave sums in order (64 threads per channel walked up to 2048 partials in 32 dependent rounds:
// 12-15 us per launch, 23 launches per step)
__device__ __forceinline__ double block256_sum(double v, double* sh4) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh4[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((sh4[0] + sh4[1]) + sh4[2]) + sh4[3];
}

__global__ __launch_bounds__(64) void bn_train_finalize_kernel(const double* __restrict__ part, int nblk, long P, int C,
        const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum, int unbiased,
        float* moving_mean, float* moving_var, float* mean, float* invstd, float* scale, float* shift) {
    const int c = blockIdx.x;
    double s = 0.0, ss = 0.0;
    for (int k = threadIdx.x; k < nblk; k += 64) { s += part[(size_t)k * C + c]; ss += part[((size_t)nblk + k) * C + c]; }
    s = wave_sum(s); ss = wave_sum(ss);
    if (threadIdx.x != 0) return;
    const double m = s / (double)P;
    double var = ss / (double)P - m * m;
    if (var < 0.0) var = 0.0;
    const double inv = 1.0 / sqrt(var + (double)eps);
    mean[c] = (float)m; invstd[c] = (float)inv;
    const double a = (double)gamma[c] * inv;
    scale[c] = (float)a; shift[c] = (float)((double)beta[c] - m * a);
    if (moving_mean) {
        const double uv = (unbiased && P > 1) ? var * ((double)P / (double)(P - 1)) : var;
        moving_mean[c] = (float)((double)moving_mean[c] * momentum + m * (1.0 - (double)momentum));
        moving_var[c] = (float)((double)moving_var[c] * momentum + uv * (1.0 - (double)momentum));
    }
}

// training statistics from the partials the fused conv kernel wrote (winograd.hip, wf_write_stats): part[C/64][rows][64][2]
// (256 threads per channel: the bf16 kernels leave one row per 16x32-pixel tile, thousands of rows at full resolution; the four
// waves' sums are added in a fixed order)
__global__ __launch_bounds__(256) void bn_train_finalize_partials_kernel(const float* __restrict__ part, int rows, long P, int C,
        const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum, int unbiased,
        float* moving_mean, float* moving_var, float* mean, float* invstd, float* scale, float* shift) {
    __shared__ double sh[4][2];
    const int c = blockIdx.x;
    const float* base = part + ((size_t)(c >> 6) * rows * 64 + (c & 63)) * 2;
    double s = 0.0, ss = 0.0;
    for (int k = threadIdx.x; k < rows; k += 256) { const float2 v = *reinterpret_cast<const float2*>(base + (size_t)k * 128); s += (double)v.x; ss += (double)v.y; }
    s = wave_sum(s); ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6][0] = s; sh[threadIdx.x >> 6][1] = ss; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    s = ((sh[0][0] + sh[1][0]) + sh[2][0]) + sh[3][0]; ss = ((sh[0][1] + sh[1][1]) + sh[2][1]) + sh[3][1];
    const double m = s / (double)P;
    double var = ss / (double)P - m * m;
    if (var < 0.0) var = 0.0;
    const double inv = 1.0 / sqrt(var + (double)eps);
    mean[c] = (float)m; invstd[c] = (float)inv;
    const double a = (double)gamma[c] * inv;
    scale[c] = (float)a; shift[c] = (float)((double)beta[c] - m * a);
    if (moving_mean) {
        const double uv = (unbiased && P > 1) ? var * ((double)P / (double)(P - 1)) : var;
        moving_mean[c] = (float)((double)moving_mean[c] * momentum + m * (1.0 - (double)momentum));
        moving_var[c] = (float)((double)moving_var[c] * momentum + uv * (1.0 - (double)momentum));
    }
}

// dgamma / dbeta from the sums a fused data-gradient kernel left (winograd.hip, STATS == 2): part[C/64][rows][64][2] holds
// sum(dy) and sum(dy * r) per channel; dbeta = sum dy, dgamma = sum dy xhat = invstd * (sum dy r - mean * sum dy)
__global__ __launch_bounds__(256) void bn_bwd_finalize_partials_kernel(const float* __restrict__ part, int rows, int C,
        const float* __restrict__ mean, const float* __restrict__ invstd, float* dgamma, float* dbeta) {
    __shared__ double sh[4][2];
    const int c = blockIdx.x;
    const float* base = part + ((size_t)(c >> 6) * rows * 64 + (c & 63)) * 2;
    double s = 0.0, sr = 0.0;
    for (int k = threadIdx.x; k < rows; k += 256) { const float2 v = *reinterpret_cast<const float2*>(base + (size_t)k * 128); s += (double)v.x; sr += (double)v.y; }
    s = wave_sum(s); sr = wave_sum(sr);
    if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6][0] = s; sh[threadIdx.x >> 6][1] = sr; }
    __syncthreads();
    if (threadIdx.x == 0) {
        s = ((sh[0][0] + sh[1][0]) + sh[2][0]) + sh[3][0]; sr = ((sh[0][1] + sh[1][1]) + sh[2][1]) + sh[3][1];
        dbeta[c] = (float)s; dgamma[c] = (float)((double)invstd[c] * (sr - (double)mean[c] * s));
    }
}

__global__ void counter_add_kernel(unsigned* counter, unsigned n) { atomicAdd(counter, n); }

__global__ void bn_eval_coeffs_kernel(const float* gamma, const float* beta, const float* mm, const float* mv, float eps,
                                      int C, float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double inv = 1.0 / sqrt((double)mv[c] + (double)eps);
    const double a = (double)gamma[c] * inv;
    scale[c] = (float)a; shift[c] = (float)((double)beta[c] - (double)mm[c] * a);
}

// ---- BatchNorm finalize merged into its consumer's launch (round 6: the small-launch A/B) --------------------------------------------
// The statistics finalize (rows of partial sums -> mean / invstd / scale / shift / moving statistics) is a <= 7 us kernel between the conv
// that leaves the partials and the apply pass that needs the coefficients: two launch boundaries on the forward chain per layer.  Merged:
// workgroup b of the APPLY grid first finalizes channels b, b + grid, ... (the same arithmetic, thread for thread, as
// bn_train_finalize_partials_kernel -- results are bit-identical), publishes them (release fence + one atomic add on a device counter) and
// every workgroup waits until the counter shows all C channels before it reads its coefficients.  Finalizing workgroups never wait for a
// non-finalizing one and have the lowest indices (dispatched first), so the wait cannot deadlock even if the grid is not fully resident.
// The counter only grows: the host passes the value it must reach (previous target + C), compared modulo 2^32.
struct BnFin {
    const float* part; int rows; long P; const float* gamma; const float* beta; float eps, momentum; int unbiased;
    float* moving_mean; float* moving_var; float* mean; float* invstd; unsigned* counter; unsigned target;
};

__device__ __forceinline__ void bn_finalize_then_wait(const BnFin& f, int C, float* scale, float* shift) {
    __shared__ double sh[4][2];
    int done = 0;
    for (int c = blockIdx.x; c < C; c += gridDim.x, ++done) {
        const float* base = f.part + ((size_t)(c >> 6) * f.rows * 64 + (c & 63)) * 2;
        double s = 0.0, ss = 0.0;
        for (int k = threadIdx.x; k < f.rows; k += 256) { const float2 v = *reinterpret_cast<const float2*>(base + (size_t)k * 128); s += (double)v.x; ss += (double)v.y; }
        s = wave_sum(s); ss = wave_sum(ss);
        __syncthreads();                                   // (sh is reused per channel)
        if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6][0] = s; sh[threadIdx.x >> 6][1] = ss; }
        __syncthreads();
        if (threadIdx.x == 0) {
            s = ((sh[0][0] + sh[1][0]) + sh[2][0]) + sh[3][0]; ss = ((sh[0][1] + sh[1][1]) + sh[2][1]) + sh[3][1];
            const double m = s / (double)f.P;
            double var = ss / (double)f.P - m * m;
            if (var < 0.0) var = 0.0;
            const double inv = 1.0 / sqrt(var + (double)f.eps);
            f.mean[c] = (float)m; f.invstd[c] = (float)inv;
            const double a = (double)f.gamma[c] * inv;
            // the two values other workgroups of THIS launch read: device-scope stores (written through to where every XCD sees them)
            __hip_atomic_store(scale + c, (float)a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(shift + c, (float)((double)f.beta[c] - m * a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (f.moving_mean) {
                const double uv = (f.unbiased && f.P > 1) ? var * ((double)f.P / (double)(f.P - 1)) : var;
                f.moving_mean[c] = (float)((double)f.moving_mean[c] * f.momentum + m * (1.0 - (double)f.momentum));
                f.moving_var[c] = (float)((double)f.moving_var[c] * f.momentum + uv * (1.0 - (double)f.momentum));
            }
        }
    }
    // No agent-scope fences: a release fence writes back this XCD's whole L2 (the conv output of a moment ago is still dirty in it) and an
    // acquire in the spin loop invalidates it -- the first cut of this function did both and cost +190 us per launch.  The published values
    // are device-scope stores, complete (s_waitcnt) before the count moves; readers take them with device-scope loads (bn_coef_load).
    if (threadIdx.x == 0) {
        if (done) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __hip_atomic_fetch_add(f.counter, (unsigned)done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        while ((int)(__hip_atomic_load(f.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - f.target) < 0) __builtin_amdgcn_s_sleep(4);
    }
    __syncthreads();
}

// coefficients of a lane: plain loads, or -- after a merged finalize in the same launch -- device-scope loads
template <int VEC> __device__ __forceinline__ void bn_coef_load(float (&v)[VEC], const float* p, bool coherent) {
    if (!coherent) { vload<VEC>(v, p); return; }
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] = __hip_atomic_load(p + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// y = scale * r + shift.  A lane owns one channel group (its scale / shift live in registers) and walks pixels with a grid-wide
// stride, four pixels per step so that four 16-byte loads are in flight per lane: with one load in flight (the round-1 form, a flat
// index loop) these passes were latency-bound at ~3.2 TB/s.  flags: bit 0 = y stored as bf16, bit 1 = r stored as bf16.
template <int VEC>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ r, int ldr, const float* scale,
        const float* shift, float* __restrict__ y, int ldy, long P, int C, int tpp, int out16, BnFin fin) {
    if (fin.part) bn_finalize_then_wait(fin, C, const_cast<float*>(scale), const_cast<float*>(shift));
    const Lay l = make_lay<VEC>(C, tpp);
    if (!l.active) return;
    float a[VEC], b[VEC];
    bn_coef_load<VEC>(a, scale + l.c0, fin.part != nullptr); bn_coef_load<VEC>(b, shift + l.c0, fin.part != nullptr);
    const long S = (long)gridDim.x * l.npl;
    long pix = (long)blockIdx.x * l.npl + l.pl;
    for (; pix + 3 * S < P; pix += 4 * S) {
        float v[4][VEC];
#pragma unroll
        for (int u = 0; u < 4; ++u) vload_dt<VEC, kApplyNT>(v[u], r, (size_t)(pix + u * S) * ldr + l.c0, out16 & 2);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[u][e] = fmaf(a[e], v[u][e], b[e]);
            vstore_dt<VEC>(y, (size_t)(pix + u * S) * ldy + l.c0, v[u], out16 & 1);
        }
    }
    for (; pix < P; pix += S) {
        float v[VEC];
        vload_dt<VEC, kApplyNT>(v, r, (size_t)pix * ldr + l.c0, out16 & 2);
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[e] = fmaf(a[e], v[e], b[e]);
        vstore_dt<VEC>(y, (size_t)pix * ldy + l.c0, v, out16 & 1);
    }
}

// the same for channel counts the lane layout does not cover (C / VEC lanes per pixel must divide 256): flat index loop
template <int VEC>
__global__ __launch_bounds__(256) void bn_apply_flat_kernel(const float* __restrict__ r, int ldr, const float* __restrict__ scale,
        const float* __restrict__ shift, float* __restrict__ y, int ldy, long P, int C, int out16) {
    const int nq = C / VEC;
    const long total = P * nq, stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const long pix = i / nq; const int c0 = (int)(i - pix * nq) * VEC;
        float v[VEC], a[VEC], b[VEC];
        vload_dt<VEC>(v, r, (size_t)pix * ldr + c0, out16 & 2); vload<VEC>(a, scale + c0); vload<VEC>(b, shift + c0);
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[e] = fmaf(a[e], v[e], b[e]);
        vstore_dt<VEC>(y, (size_t)pix * ldy + c0, v, out16 & 1);
    }
}

// BN apply fused with the 2x2 max pool that follows the encoder's second conv of a level (UNet/model.py:36,50-53): one thread =
// one pooled pixel x channel quad: reads the 4 r values, writes the 4 normalised values (the skip tensor) and their first-max
// (row-major window order, the reference's tie rule) + its index -- the skip tensor is not read back for pooling.
template <int VEC>
__global__ __launch_bounds__(256) void bn_apply_pool_kernel(const float* __restrict__ r, int ldr, const float* scale,
        const float* shift, float* __restrict__ y, int ldy, float* __restrict__ pooled, int ldp, uint8_t* __restrict__ idx,
        int N, int H, int W, int C, int tpp, int out16, BnFin fin) {
    if (fin.part) bn_finalize_then_wait(fin, C, const_cast<float*>(scale), const_cast<float*>(shift));
    const Lay l = make_lay<VEC>(C, tpp);
    if (!l.active) return;
    const int H2 = H / 2, W2 = W / 2;
    const long total = (long)N * H2 * W2, S = (long)gridDim.x * l.npl;
    float a[VEC], b[VEC];
    bn_coef_load<VEC>(a, scale + l.c0, fin.part != nullptr); bn_coef_load<VEC>(b, shift + l.c0, fin.part != nullptr);
    for (long opix = (long)blockIdx.x * l.npl + l.pl; opix < total; opix += S) {
        long t = opix; const int ox = (int)(t % W2); t /= W2; const int oy = (int)(t % H2); const int n = (int)(t / H2);
        float best[VEC]; uint8_t bi[VEC];
        float v[4][VEC];
#pragma unroll
        for (int pos = 0; pos < 4; ++pos) {
            const size_t pix = (size_t)((long)n * H + 2 * oy + (pos >> 1)) * W + 2 * ox + (pos & 1);
            vload_dt<VEC, kApplyNT>(v[pos], r, pix * ldr + l.c0, out16 & 2);
        }
#pragma unroll
        for (int pos = 0; pos < 4; ++pos) {
            const size_t pix = (size_t)((long)n * H + 2 * oy + (pos >> 1)) * W + 2 * ox + (pos & 1);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                v[pos][e] = fmaf(a[e], v[pos][e], b[e]);
                if (pos == 0 || v[pos][e] > best[e]) { best[e] = v[pos][e]; bi[e] = (uint8_t)pos; }
            }
            vstore_dt<VEC>(y, pix * ldy + l.c0, v[pos], out16 & 1);
        }
        vstore_dt<VEC>(pooled, (size_t)opix * ldp + l.c0, best, out16 & 1);
        const uint32_t w0 = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
        if constexpr (VEC == 8) {
            const uint32_t w1 = (uint32_t)bi[4] | ((uint32_t)bi[5] << 8) | ((uint32_t)bi[6] << 16) | ((uint32_t)bi[7] << 24);
            *reinterpret_cast<uint2*>(idx + (size_t)opix * C + l.c0) = make_uint2(w0, w1);
        } else {
            *reinterpret_cast<uint32_t*>(idx + (size_t)opix * C + l.c0) = w0;
        }
    }
}

// backward pass 1: per-channel sum(dy) and sum(dy * xhat)
// The per-lane sums are taken in fp32 over the four pixels of a step and added to fp64 running sums once per step (a quarter of
// the fp64 conversions / additions of the element-wise form, which made these passes instruction-bound once their tensors shrank
// to bf16); sum(dy * xhat) is accumulated as sum(dy * (r - mean)) and multiplied by invstd in the finalize kernel.
template <int VEC>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ r,
        int ldr, const float* __restrict__ mean, const float* __restrict__ invstd, long P, int C, int tpp, long ppb,
        double* __restrict__ part, PoolGrad pg, int dt) {
    extern __shared__ __attribute__((aligned(16))) double sRd[];
    const Lay l = make_lay<VEC>(C, tpp);
    const long p0 = (long)blockIdx.x * ppb; long p1 = p0 + ppb; if (p1 > P) p1 = P;
    double acc[2][VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { acc[0][e] = 0.0; acc[1][e] = 0.0; }
    if (l.active) {
        float mu[VEC]; vload<VEC>(mu, mean + l.c0);
        long pix = p0 + l.pl;
        const long st = l.npl;
        for (; pix + 3 * st < p1; pix += 4 * st) {
            float g[4][VEC], v[4][VEC];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                vload_dt<VEC>(g[u], dy, (size_t)(pix + u * st) * lddy + l.c0, dt & 4); vload_dt<VEC>(v[u], r, (size_t)(pix + u * st) * ldr + l.c0, dt & 2);
                add_pool_grad<VEC>(g[u], pg, pix + u * st, l.c0);
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float s0 = (g[0][e] + g[1][e]) + (g[2][e] + g[3][e]);
                float s1 = g[0][e] * (v[0][e] - mu[e]);
                s1 = fmaf(g[1][e], v[1][e] - mu[e], s1); s1 = fmaf(g[2][e], v[2][e] - mu[e], s1); s1 = fmaf(g[3][e], v[3][e] - mu[e], s1);
                acc[0][e] += (double)s0; acc[1][e] += (double)s1;
            }
        }
        for (; pix < p1; pix += st) {
            float g[VEC], v[VEC];
            vload_dt<VEC>(g, dy, (size_t)pix * lddy + l.c0, dt & 4); vload_dt<VEC>(v, r, (size_t)pix * ldr + l.c0, dt & 2);
            add_pool_grad<VEC>(g, pg, pix, l.c0);
#pragma unroll
            for (int e = 0; e < VEC; ++e) { acc[0][e] += (double)g[e]; acc[1][e] += (double)(g[e] * (v[e] - mu[e])); }
        }
    }
    block_combine<VEC, 2>(acc, l, C, part, gridDim.x, sRd);
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double* __restrict__ part, int nblk, int C, const float* __restrict__ invstd,
                                                              float* dgamma, float* dbeta) {
    __shared__ double sh[2][4];
    const int c = blockIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int k = threadIdx.x; k < nblk; k += 256) { s1 += part[(size_t)k * C + c]; s2 += part[((size_t)nblk + k) * C + c]; }
    s1 = block256_sum(s1, sh[0]); s2 = block256_sum(s2, sh[1]);
    if (threadIdx.x == 0) { dbeta[c] = (float)s1; dgamma[c] = (float)(s2 * (double)invstd[c]); }       // sum dy (r - mean) * invstd
}

// backward pass 2: dz = relu'(r) * gamma*invstd * (dy - mean(dy) - xhat * mean(dy*xhat)); also per-channel sum(dz).
// With a = gamma * invstd, c1 = dbeta / P, c2 = dgamma / P and xhat = (r - mean) * invstd this is
//     dz = relu'(r) * (A * dy + B * r + K),  A = a,  B = -a * c2 * invstd,  K = a * (c2 * invstd * mean - c1):
// two FMAs per element with three per-channel constants in registers.
template <int VEC>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ r,
        int ldr, const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ invstd,
        const float* __restrict__ dgamma, const float* __restrict__ dbeta, long P, int C, int tpp, long ppb, int relu,
        float* __restrict__ dz, int lddz, double* __restrict__ part, PoolGrad pg, int dt) {
    extern __shared__ __attribute__((aligned(16))) double sRd[];
    const Lay l = make_lay<VEC>(C, tpp);
    const long p0 = (long)blockIdx.x * ppb; long p1 = p0 + ppb; if (p1 > P) p1 = P;
    double acc[1][VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[0][e] = 0.0;
    if (l.active) {
        float A[VEC], Bc[VEC], K[VEC];
        {
            float mu[VEC], is[VEC], ga[VEC], dg[VEC], db[VEC];
            vload<VEC>(mu, mean + l.c0); vload<VEC>(is, invstd + l.c0); vload<VEC>(ga, gamma + l.c0);
            vload<VEC>(dg, dgamma + l.c0); vload<VEC>(db, dbeta + l.c0);
            const float invP = 1.0f / (float)P;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float a = ga[e] * is[e], c1 = db[e] * invP, c2 = dg[e] * invP;
                A[e] = a; Bc[e] = -(a * (c2 * is[e])); K[e] = a * (c2 * is[e] * mu[e] - c1);
            }
        }
        long pix = p0 + l.pl;
        const long st = l.npl;
        for (; pix + 3 * st < p1; pix += 4 * st) {
            float g[4][VEC], v[4][VEC];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                vload_dt<VEC>(g[u], dy, (size_t)(pix + u * st) * lddy + l.c0, dt & 4); vload_dt<VEC>(v[u], r, (size_t)(pix + u * st) * ldr + l.c0, dt & 2);
                add_pool_grad<VEC>(g[u], pg, pix + u * st, l.c0);
            }
            float sum4[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) sum4[e] = 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float o[VEC];
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    float d = fmaf(A[e], g[u][e], fmaf(Bc[e], v[u][e], K[e]));
                    if (relu && !(v[u][e] > 0.f)) d = 0.f;
                    o[e] = d; sum4[e] += d;
                }
                vstore_dt<VEC>(dz, (size_t)(pix + u * st) * lddz + l.c0, o, dt & 1);
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[0][e] += (double)sum4[e];
        }
        for (; pix < p1; pix += st) {
            float g[VEC], v[VEC], o[VEC];
            vload_dt<VEC>(g, dy, (size_t)pix * lddy + l.c0, dt & 4); vload_dt<VEC>(v, r, (size_t)pix * ldr + l.c0, dt & 2);
            add_pool_grad<VEC>(g, pg, pix, l.c0);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                float d = fmaf(A[e], g[e], fmaf(Bc[e], v[e], K[e]));
                if (relu && !(v[e] > 0.f)) d = 0.f;
                o[e] = d; acc[0][e] += (double)d;
            }
            vstore_dt<VEC>(dz, (size_t)pix * lddz + l.c0, o, dt & 1);
        }
    }
    block_combine<VEC, 1>(acc, l, C, part, gridDim.x, sRd);
}

// backward pass 2 with dy, r AND dz stored as bf16 (the mixed-precision training step: every wide layer), 8 channels = 16 bytes per lane.
// Same arithmetic, same order of additions as bn_bwd_apply_kernel<8> -- dz is bit-identical and the per-lane sums are formed over the same
// groups of four pixels -- but the loaded words stay PACKED until they are used: 8 pixels in flight cost 64 registers instead of the 128
// unpacked floats (+ the fp32-storage code path) of the generic kernel, which sits at 200 VGPRs = 2 waves per SIMD and streams at
// 3.8 TB/s; this one keeps 4+ waves per SIMD.
__device__ __forceinline__ void bwd16_pixel(const uint4& g, const uint4& v, const float (&A)[8], const float (&Bc)[8], const float (&K)[8],
                                            int relu, float (&sum)[8], uint4& o) {
    const unsigned gw[4] = {g.x, g.y, g.z, g.w}, vw[4] = {v.x, v.y, v.z, v.w};
    unsigned ow[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float v0 = bf_lo(vw[j]), v1 = bf_hi(vw[j]);
        float d0 = fmaf(A[2 * j], bf_lo(gw[j]), fmaf(Bc[2 * j], v0, K[2 * j]));
        float d1 = fmaf(A[2 * j + 1], bf_hi(gw[j]), fmaf(Bc[2 * j + 1], v1, K[2 * j + 1]));
        if (relu && !(v0 > 0.f)) d0 = 0.f;
        if (relu && !(v1 > 0.f)) d1 = 0.f;
        sum[2 * j] += d0; sum[2 * j + 1] += d1;
        ow[j] = bn_pack2(d0, d1);
    }
    o = make_uint4(ow[0], ow[1], ow[2], ow[3]);
}

// Addressing: buffer loads / stores with 32-bit per-lane offsets (the lane's part + a uniform step, one add per access) -- 64-bit
// per-access addresses would cost 2 registers for each of the 24 accesses in flight.  (The uniform part went into the instruction's SCALAR
// offset at first: a few stores per launch then landed wrong at full size -- 0.05 % of the elements, sums intact -- in a pattern that
// pointed at the scalar-offset path and was not understood further; the end-to-end training test on the reference's tiles caught it.)
__global__ __launch_bounds__(256) void bn_bwd_apply16_kernel(const uint16_t* __restrict__ dy, int lddy, const uint16_t* __restrict__ r,
        int ldr, const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ invstd,
        const float* __restrict__ dgamma, const float* __restrict__ dbeta, long P, int C, int tpp, long ppb, int relu,
        uint16_t* __restrict__ dz, int lddz, double* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) double sRd[];
    constexpr int VEC = 8;
    const Lay l = make_lay<VEC>(C, tpp);
    const long p0 = (long)blockIdx.x * ppb; long p1 = p0 + ppb; if (p1 > P) p1 = P;
    double acc[1][VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[0][e] = 0.0;
    if (l.active) {
        float A[VEC], Bc[VEC], K[VEC];
        {
            float mu[VEC], is[VEC], ga[VEC], dg[VEC], db[VEC];
            vload<VEC>(mu, mean + l.c0); vload<VEC>(is, invstd + l.c0); vload<VEC>(ga, gamma + l.c0);
            vload<VEC>(dg, dgamma + l.c0); vload<VEC>(db, dbeta + l.c0);
            const float invP = 1.0f / (float)P;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float a = ga[e] * is[e], c1 = db[e] * invP, c2 = dg[e] * invP;
                A[e] = a; Bc[e] = -(a * (c2 * is[e])); K[e] = a * (c2 * is[e] * mu[e] - c1);
            }
        }
        const __amdgpu_buffer_rsrc_t sy = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, (int)((size_t)P * lddy * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc((void*)r, 0, (int)((size_t)P * ldr * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t sz = __builtin_amdgcn_make_buffer_rsrc((void*)dz, 0, (int)((size_t)P * lddz * 2), 0x00020000);
        const int vy = (l.pl * lddy + l.c0) * 2, vr = (l.pl * ldr + l.c0) * 2, vz = (l.pl * lddz + l.c0) * 2;
        const int st = l.npl;
        long base = p0;                                           // uniform: pixel of pixel-lane 0 in this step; the lane's pixel = base + l.pl
        // main loop: 8 steps while EVERY pixel lane of the block is inside the range (uniform trip count, scalar offsets)
        for (; base + (st - 1) + 7L * st < p1; base += 8L * st) {
            i32x4v g[8], v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                g[u] = __builtin_amdgcn_raw_buffer_load_b128(sy, vy + (int)((base + (long)u * st) * lddy * 2), 0, UNET_NT_AUX(UNET_NT_LDBN16));
                v[u] = __builtin_amdgcn_raw_buffer_load_b128(sr, vr + (int)((base + (long)u * st) * ldr * 2), 0, UNET_NT_AUX(UNET_NT_LDBN16));
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {                        // sums over groups of four pixels, as the generic kernel forms them
                float sum4[VEC];
#pragma unroll
                for (int e = 0; e < VEC; ++e) sum4[e] = 0.f;
#pragma unroll
                for (int u = 4 * h; u < 4 * h + 4; ++u) {
                    uint4 o;
                    bwd16_pixel(make_uint4(g[u][0], g[u][1], g[u][2], g[u][3]), make_uint4(v[u][0], v[u][1], v[u][2], v[u][3]), A, Bc, K, relu, sum4, o);
                    const i32x4v ov = {(int)o.x, (int)o.y, (int)o.z, (int)o.w};
                    __builtin_amdgcn_raw_buffer_store_b128(ov, sz, vz + (int)((base + (long)u * st) * lddz * 2), 0, UNET_NT_AUX(UNET_NT_BN16));
                }
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[0][e] += (double)sum4[e];
            }
        }
        // the end of the range: per lane, groups of four and then single pixels (the same grouping as the generic kernel's)
        const uint16_t* dyc = dy + l.c0; const uint16_t* rc = r + l.c0; uint16_t* dzc = dz + l.c0;
        long pix = base + l.pl;
        for (; pix + 3L * st < p1; pix += 4L * st) {
            uint4 g[4], v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                g[u] = *reinterpret_cast<const uint4*>(dyc + (size_t)(pix + (long)u * st) * lddy);
                v[u] = *reinterpret_cast<const uint4*>(rc + (size_t)(pix + (long)u * st) * ldr);
            }
            float sum4[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) sum4[e] = 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                uint4 o;
                bwd16_pixel(g[u], v[u], A, Bc, K, relu, sum4, o);
                *reinterpret_cast<uint4*>(dzc + (size_t)(pix + (long)u * st) * lddz) = o;
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[0][e] += (double)sum4[e];
        }
        for (; pix < p1; pix += st) {
            const uint4 g = *reinterpret_cast<const uint4*>(dyc + (size_t)pix * lddy), v = *reinterpret_cast<const uint4*>(rc + (size_t)pix * ldr);
            float s1[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) s1[e] = 0.f;
            uint4 o;
            bwd16_pixel(g, v, A, Bc, K, relu, s1, o);
            *reinterpret_cast<uint4*>(dzc + (size_t)pix * lddz) = o;
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[0][e] += (double)s1[e];
        }
    }
    block_combine<VEC, 1>(acc, l, C, part, gridDim.x, sRd);
}

// ---- the two backward passes for a layer whose output also went through MaxPool2D(2) (PoolGrad): a lane owns one POOLED pixel
// (and its channel group) per step -- it loads the pooled gradient and the first-max indices once and walks the 2x2 window, instead
// of every one of the four window pixels fetching them again through the cache (the per-pixel form ran at 2.1-2.6 TB/s).
template <int VEC> __device__ __forceinline__ void load_idx(uint8_t (&ix)[VEC], const uint8_t* p) {
    if constexpr (VEC == 8) { const uint2 w = *reinterpret_cast<const uint2*>(p);
#pragma unroll
        for (int e = 0; e < 4; ++e) { ix[e] = (uint8_t)(w.x >> (8 * e)); ix[4 + e] = (uint8_t)(w.y >> (8 * e)); } }
    else if constexpr (VEC == 4) { const uint32_t w = *reinterpret_cast<const uint32_t*>(p);
#pragma unroll
        for (int e = 0; e < 4; ++e) ix[e] = (uint8_t)(w >> (8 * e)); }
    else ix[0] = *p;
}

template <int VEC>
__global__ __launch_bounds__(256) void bn_bwd_reduce_pool_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ r,
        int ldr, const float* __restrict__ mean, long P2, int C, int tpp, long ppb, double* __restrict__ part, PoolGrad pg, int dt) {
    extern __shared__ __attribute__((aligned(16))) double sRd[];
    const Lay l = make_lay<VEC>(C, tpp);
    const long p0 = (long)blockIdx.x * ppb; long p1 = p0 + ppb; if (p1 > P2) p1 = P2;
    const int H2 = pg.H >> 1, W2 = pg.W >> 1;
    double acc[2][VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { acc[0][e] = 0.0; acc[1][e] = 0.0; }
    if (l.active) {
        float mu[VEC]; vload<VEC>(mu, mean + l.c0);
        for (long op = p0 + l.pl; op < p1; op += l.npl) {
            long t = op; const int ox = (int)(t % W2); t /= W2; const int oy = (int)(t % H2); const long n = t / H2;
            float pd[VEC]; uint8_t ix[VEC];
            vload_dt<VEC>(pd, pg.pdy, (size_t)op * pg.ldp + l.c0, pg.p16);
            load_idx<VEC>(ix, pg.idx + (size_t)op * pg.C + l.c0);
            float g[4][VEC], v[4][VEC];
#pragma unroll
            for (int pos = 0; pos < 4; ++pos) {
                const size_t pix = (size_t)((n * pg.H + 2 * oy + (pos >> 1)) * pg.W + 2 * ox + (pos & 1));
                vload_dt<VEC>(g[pos], dy, pix * lddy + l.c0, dt & 4); vload_dt<VEC>(v[pos], r, pix * ldr + l.c0, dt & 2);
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
#pragma unroll
                for (int pos = 0; pos < 4; ++pos) if (ix[e] == pos) g[pos][e] += pd[e];
                const float s0 = (g[0][e] + g[1][e]) + (g[2][e] + g[3][e]);
                float s1 = g[0][e] * (v[0][e] - mu[e]);
                s1 = fmaf(g[1][e], v[1][e] - mu[e], s1); s1 = fmaf(g[2][e], v[2][e] - mu[e], s1); s1 = fmaf(g[3][e], v[3][e] - mu[e], s1);
                acc[0][e] += (double)s0; acc[1][e] += (double)s1;
            }
        }
    }
    block_combine<VEC, 2>(acc, l, C, part, gridDim.x, sRd);
}

template <int VEC>
__global__ __launch_bounds__(256) void bn_bwd_apply_pool_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ r,
        int ldr, const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ invstd,
        const float* __restrict__ dgamma, const float* __restrict__ dbeta, long P, long P2, int C, int tpp, long ppb, int relu,
        float* __restrict__ dz, int lddz, double* __restrict__ part, PoolGrad pg, int dt) {
    extern __shared__ __attribute__((aligned(16))) double sRd[];
    const Lay l = make_lay<VEC>(C, tpp);
    const long p0 = (long)blockIdx.x * ppb; long p1 = p0 + ppb; if (p1 > P2) p1 = P2;
    const int H2 = pg.H >> 1, W2 = pg.W >> 1;
    double acc[1][VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[0][e] = 0.0;
    if (l.active) {
        float A[VEC], Bc[VEC], K[VEC];
        {
            float mu[VEC], is[VEC], ga[VEC], dg[VEC], db[VEC];
            vload<VEC>(mu, mean + l.c0); vload<VEC>(is, invstd + l.c0); vload<VEC>(ga, gamma + l.c0);
            vload<VEC>(dg, dgamma + l.c0); vload<VEC>(db, dbeta + l.c0);
            const float invP = 1.0f / (float)P;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float a = ga[e] * is[e], c1 = db[e] * invP, c2 = dg[e] * invP;
                A[e] = a; Bc[e] = -(a * (c2 * is[e])); K[e] = a * (c2 * is[e] * mu[e] - c1);
            }
        }
        for (long op = p0 + l.pl; op < p1; op += l.npl) {
            long t = op; const int ox = (int)(t % W2); t /= W2; const int oy = (int)(t % H2); const long n = t / H2;
            float pd[VEC]; uint8_t ix[VEC];
            vload_dt<VEC>(pd, pg.pdy, (size_t)op * pg.ldp + l.c0, pg.p16);
            load_idx<VEC>(ix, pg.idx + (size_t)op * pg.C + l.c0);
            float g[4][VEC], v[4][VEC];
#pragma unroll
            for (int pos = 0; pos < 4; ++pos) {
                const size_t pix = (size_t)((n * pg.H + 2 * oy + (pos >> 1)) * pg.W + 2 * ox + (pos & 1));
                vload_dt<VEC>(g[pos], dy, pix * lddy + l.c0, dt & 4); vload_dt<VEC>(v[pos], r, pix * ldr + l.c0, dt & 2);
            }
            float sum4[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) sum4[e] = 0.f;
#pragma unroll
            for (int pos = 0; pos < 4; ++pos) {
                const size_t pix = (size_t)((n * pg.H + 2 * oy + (pos >> 1)) * pg.W + 2 * ox + (pos & 1));
                float o[VEC];
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const float ge = g[pos][e] + (ix[e] == pos ? pd[e] : 0.f);
                    float d = fmaf(A[e], ge, fmaf(Bc[e], v[pos][e], K[e]));
                    if (relu && !(v[pos][e] > 0.f)) d = 0.f;
                    o[e] = d; sum4[e] += d;
                }
                vstore_dt<VEC>(dz, pix * lddz + l.c0, o, dt & 1);
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[0][e] += (double)sum4[e];
        }
    }
    block_combine<VEC, 1>(acc, l, C, part, gridDim.x, sRd);
}

// The two pooled passes with dy, r, the pooled gradient AND dz stored as bf16 (encoder levels 1-3 of the mixed-precision step): packed
// registers and buffer addressing as in bn_bwd_apply16_kernel; same arithmetic and the same order of additions as the generic
// window-per-lane kernels above (which sit at 138-146 VGPRs, 3 waves per SIMD, ~3.8 TB/s).
struct Pool16 { i32x4v g[4], v[4], pd; unsigned ixlo, ixhi; };

template <int LAST>               // LAST: the apply pass, which reads dy and r for the last time (streaming policy, common.h)
__device__ __forceinline__ void pool16_load(Pool16& w, const __amdgpu_buffer_rsrc_t& sy, const __amdgpu_buffer_rsrc_t& sr,
                                            const __amdgpu_buffer_rsrc_t& sp, const uint8_t* __restrict__ idx, long op, int c0, int C,
                                            int lddy, int ldr, int ldp, int H, int W) {
    const int W2 = W >> 1, H2 = H >> 1;
    long t = op; const int ox = (int)(t % W2); t /= W2; const int oy = (int)(t % H2); const long n = t / H2;
    w.pd = __builtin_amdgcn_raw_buffer_load_b128(sp, (int)((op * ldp + c0) * 2), 0, 0);
    const uint2 ix = *reinterpret_cast<const uint2*>(idx + (size_t)op * C + c0);
    w.ixlo = ix.x; w.ixhi = ix.y;
#pragma unroll
    for (int pos = 0; pos < 4; ++pos) {
        const long pix = (n * H + 2 * oy + (pos >> 1)) * W + 2 * ox + (pos & 1);
        w.g[pos] = __builtin_amdgcn_raw_buffer_load_b128(sy, (int)((pix * lddy + c0) * 2), 0, LAST ? UNET_NT_AUX(UNET_NT_LDBN16) : 0);
        w.v[pos] = __builtin_amdgcn_raw_buffer_load_b128(sr, (int)((pix * ldr + c0) * 2), 0, LAST ? UNET_NT_AUX(UNET_NT_LDBN16) : 0);
    }
}
__device__ __forceinline__ float bf_elem(const i32x4v& q, int e) { const unsigned wd = (unsigned)q[e >> 1]; return (e & 1) ? bf_hi(wd) : bf_lo(wd); }
__device__ __forceinline__ int pool16_winner(const Pool16& w, int e) { return (int)(((e < 4 ? w.ixlo : w.ixhi) >> (8 * (e & 3))) & 0xffu); }

__global__ __launch_bounds__(256) void bn_bwd_reduce_pool16_kernel(const uint16_t* __restrict__ dy, int lddy, const uint16_t* __restrict__ r,
        int ldr, const float* __restrict__ mean, long P, long P2, int C, int tpp, long ppb, double* __restrict__ part, PoolGrad pg) {
    extern __shared__ __attribute__((aligned(16))) double sRd[];
    constexpr int VEC = 8;
    const Lay l = make_lay<VEC>(C, tpp);
    const long p0 = (long)blockIdx.x * ppb; long p1 = p0 + ppb; if (p1 > P2) p1 = P2;
    double acc[2][VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { acc[0][e] = 0.0; acc[1][e] = 0.0; }
    if (l.active) {
        float mu[VEC]; vload<VEC>(mu, mean + l.c0);
        const __amdgpu_buffer_rsrc_t sy = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, (int)((size_t)P * lddy * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc((void*)r, 0, (int)((size_t)P * ldr * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t sp = __builtin_amdgcn_make_buffer_rsrc((void*)pg.pdy, 0, (int)((size_t)P2 * pg.ldp * 2), 0x00020000);
        for (long op = p0 + l.pl; op < p1; op += l.npl) {
            Pool16 w;
            pool16_load<0>(w, sy, sr, sp, pg.idx, op, l.c0, C, lddy, ldr, pg.ldp, pg.H, pg.W);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const int win = pool16_winner(w, e);
                const float pd = bf_elem(w.pd, e);
                float g[4];
#pragma unroll
                for (int pos = 0; pos < 4; ++pos) { g[pos] = bf_elem(w.g[pos], e); if (win == pos) g[pos] += pd; }
                const float s0 = (g[0] + g[1]) + (g[2] + g[3]);
                float s1 = g[0] * (bf_elem(w.v[0], e) - mu[e]);
                s1 = fmaf(g[1], bf_elem(w.v[1], e) - mu[e], s1); s1 = fmaf(g[2], bf_elem(w.v[2], e) - mu[e], s1); s1 = fmaf(g[3], bf_elem(w.v[3], e) - mu[e], s1);
                acc[0][e] += (double)s0; acc[1][e] += (double)s1;
            }
        }
    }
    block_combine<VEC, 2>(acc, l, C, part, gridDim.x, sRd);
}

__global__ __launch_bounds__(256) void bn_bwd_apply_pool16_kernel(const uint16_t* __restrict__ dy, int lddy, const uint16_t* __restrict__ r,
        int ldr, const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ invstd,
        const float* __restrict__ dgamma, const float* __restrict__ dbeta, long P, long P2, int C, int tpp, long ppb, int relu,
        uint16_t* __restrict__ dz, int lddz, double* __restrict__ part, PoolGrad pg) {
    extern __shared__ __attribute__((aligned(16))) double sRd[];
    constexpr int VEC = 8;
    const Lay l = make_lay<VEC>(C, tpp);
    const long p0 = (long)blockIdx.x * ppb; long p1 = p0 + ppb; if (p1 > P2) p1 = P2;
    const int H2 = pg.H >> 1, W2 = pg.W >> 1;
    double acc[1][VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[0][e] = 0.0;
    if (l.active) {
        float A[VEC], Bc[VEC], K[VEC];
        {
            float mu[VEC], is[VEC], ga[VEC], dg[VEC], db[VEC];
            vload<VEC>(mu, mean + l.c0); vload<VEC>(is, invstd + l.c0); vload<VEC>(ga, gamma + l.c0);
            vload<VEC>(dg, dgamma + l.c0); vload<VEC>(db, dbeta + l.c0);
            const float invP = 1.0f / (float)P;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float a = ga[e] * is[e], c1 = db[e] * invP, c2 = dg[e] * invP;
                A[e] = a; Bc[e] = -(a * (c2 * is[e])); K[e] = a * (c2 * is[e] * mu[e] - c1);
            }
        }
        const __amdgpu_buffer_rsrc_t sy = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, (int)((size_t)P * lddy * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc((void*)r, 0, (int)((size_t)P * ldr * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t sp = __builtin_amdgcn_make_buffer_rsrc((void*)pg.pdy, 0, (int)((size_t)P2 * pg.ldp * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t sz = __builtin_amdgcn_make_buffer_rsrc((void*)dz, 0, (int)((size_t)P * lddz * 2), 0x00020000);
        for (long op = p0 + l.pl; op < p1; op += l.npl) {
            Pool16 w;
            pool16_load<1>(w, sy, sr, sp, pg.idx, op, l.c0, C, lddy, ldr, pg.ldp, pg.H, pg.W);
            long t = op; const int ox = (int)(t % W2); t /= W2; const int oy = (int)(t % H2); const long n = t / H2;
            float sum4[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) sum4[e] = 0.f;
#pragma unroll
            for (int pos = 0; pos < 4; ++pos) {
                const long pix = (n * pg.H + 2 * oy + (pos >> 1)) * pg.W + 2 * ox + (pos & 1);
                i32x4v ov;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float d[2];
#pragma unroll
                    for (int k2 = 0; k2 < 2; ++k2) {
                        const int e = 2 * j + k2;
                        const float ge = bf_elem(w.g[pos], e) + (pool16_winner(w, e) == pos ? bf_elem(w.pd, e) : 0.f);
                        const float ve = bf_elem(w.v[pos], e);
                        float dd = fmaf(A[e], ge, fmaf(Bc[e], ve, K[e]));
                        if (relu && !(ve > 0.f)) dd = 0.f;
                        d[k2] = dd; sum4[e] += dd;
                    }
                    ov[j] = (int)bn_pack2(d[0], d[1]);
                }
                __builtin_amdgcn_raw_buffer_store_b128(ov, sz, (int)((pix * lddz + l.c0) * 2), 0, UNET_NT_AUX(UNET_NT_BN16));
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[0][e] += (double)sum4[e];
        }
    }
    block_combine<VEC, 1>(acc, l, C, part, gridDim.x, sRd);
}

__global__ __launch_bounds__(256) void colsum_finalize_kernel(const double* __restrict__ part, int nblk, int C, float* out) {
    __shared__ double sh[4];
    const int c = blockIdx.x;
    double s = 0.0;
    for (int k = threadIdx.x; k < nblk; k += 256) s += part[(size_t)k * C + c];
    s = block256_sum(s, sh);
    if (threadIdx.x == 0) out[c] = (float)s;
}

constexpr long MAX_BLOCKS = 2048;
struct Plan { int vec, tpp, nblk; long ppb; size_t smem2, smem1; };

bool make_plan(long P, int C, int ld_a, int ld_b, int ld_c, bool aligned, Plan* pl, bool wide = false) {
    int vec = 0, tpp = 0;
    // wide: 8 channels (16 bytes of a bf16 tensor) per lane -- chosen whenever one of the tensors is stored as bf16
    if (wide && C % 8 == 0 && (C / 8) <= 256 && 256 % (C / 8) == 0 && ld_a % 8 == 0 && ld_b % 8 == 0 && ld_c % 8 == 0 && aligned) { vec = 8; tpp = C / 8; }
    else if (C % 4 == 0 && (C / 4) <= 256 && 256 % (C / 4) == 0 && ld_a % 4 == 0 && ld_b % 4 == 0 && ld_c % 4 == 0 && aligned) { vec = 4; tpp = C / 4; }
    else if (C <= 256) { vec = 1; tpp = 1; while (tpp < C) tpp <<= 1; }
    else return false;
    // a block covers 256/tpp pixels per pass; give every lane ~8 passes, up to 2048 blocks (8 per CU)
    const long npl = 256 / tpp;
    long nblk = (P + npl * 8 - 1) / (npl * 8); if (nblk > MAX_BLOCKS) nblk = MAX_BLOCKS; if (nblk < 1) nblk = 1;
    pl->vec = vec; pl->tpp = tpp; pl->nblk = (int)nblk; pl->ppb = (P + nblk - 1) / nblk;
    pl->smem2 = (size_t)256 * vec * 2 * sizeof(double); pl->smem1 = (size_t)256 * vec * sizeof(double);
    return true;
}

int nblk_for(long P) { (void)P; return (int)MAX_BLOCKS; }        // workspace is sized for the largest grid

// 8 channels per lane for every tensor mix the shapes allow: with one lane layout for both storages the sums are added in one order,
// so bf16 storage of dz / y stays bit-identical to fp32 storage.
constexpr bool bn_wide_always() { return true; }

int bn_cus() {
    static int cus = 0;
    if (!cus) { int dev = 0; hipDeviceProp_t pr; (void)hipGetDevice(&dev); cus = (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256; }
    return cus;
}

// Largest grid <= MAX_BLOCKS that is a whole number of resident waves of workgroups for this kernel (its registers decide how many
// 256-thread workgroups a CU holds): a 2048-block launch of a kernel that fits 5 per CU runs 5 + 3 per CU -- the second wave leaves
// 3/8 of the machine idle while it finishes.
template <class K> int resident_grid(K kernel, size_t smem) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, smem) != hipSuccess || per_cu < 1) per_cu = 4;
    long g = (long)per_cu * bn_cus();
    return (int)(g > MAX_BLOCKS ? MAX_BLOCKS / bn_cus() * bn_cus() : g);
}
template <int VEC> int reduce_grid(size_t smem) { static int g = 0; if (!g) g = resident_grid(bn_bwd_reduce_kernel<VEC>, smem); return g; }
template <int VEC> int apply_grid(size_t smem) { static int g = 0; if (!g) g = resident_grid(bn_bwd_apply_kernel<VEC>, smem); return g; }
int pool16_grid(int which, size_t smem) {
    static int g[2] = {0, 0};
    if (!g[which]) g[which] = which ? resident_grid(bn_bwd_apply_pool16_kernel, smem) : resident_grid(bn_bwd_reduce_pool16_kernel, smem);
    return g[which];
}
int apply16_grid(size_t smem) { static int g = 0; if (!g) g = resident_grid(bn_bwd_apply16_kernel, smem); return g; }

// BatchNorm apply (+ 2x2 max pool when pooled != null) for any storage mix; picks the lane layout
int launch_bn_apply(const float* r, int ldr, int r16, const float* scale, const float* shift, float* y, int ldy, int y16,
                    float* pooled, int ldp, uint8_t* idx, int N, int H, int W, int C, hipStream_t st, const BnFin* finp = nullptr) {
    BnFin fin{}; if (finp) fin = *finp;
    const int flags = (y16 ? 1 : 0) | (r16 ? 2 : 0);
    const long P = (long)N * H * W;
    const bool al = unet_aligned16(r) && unet_aligned16(y) && unet_aligned16(scale) && unet_aligned16(shift) && (!pooled || unet_aligned16(pooled));
    int vec = 1;
    if ((flags || bn_wide_always()) && al && C % 8 == 0 && ldr % 8 == 0 && ldy % 8 == 0 && (!pooled || ldp % 8 == 0) && 256 % (C / 8) == 0) vec = 8;
    else if (al && C % 4 == 0 && ldr % 4 == 0 && ldy % 4 == 0 && (!pooled || ldp % 4 == 0)) vec = 4;
    if (flags && vec < 4) return UNET_EINVAL;                             // bf16 tensors need the vector forms
    int tpp = C / vec;
    bool lanes = (256 % tpp == 0);                                        // C / vec lanes per pixel tile a 256-thread block
    if (vec == 1) { tpp = 1; while (tpp < C) tpp <<= 1; lanes = tpp <= 256; }
    if (pooled) {
        if (!lanes || vec < 4) return UNET_EINVAL;
        const long total = (long)N * (H / 2) * (W / 2), npl = 256 / tpp;
        long blocks = (total + npl * 2 - 1) / (npl * 2); if (blocks > MAX_BLOCKS) blocks = MAX_BLOCKS; if (blocks < 1) blocks = 1;
        if (vec == 8) bn_apply_pool_kernel<8><<<(int)blocks, 256, 0, st>>>(r, ldr, scale, shift, y, ldy, pooled, ldp, idx, N, H, W, C, tpp, flags, fin);
        else          bn_apply_pool_kernel<4><<<(int)blocks, 256, 0, st>>>(r, ldr, scale, shift, y, ldy, pooled, ldp, idx, N, H, W, C, tpp, flags, fin);
        return UNET_LAUNCH_STATUS();
    }
    if (lanes) {
        const long npl = 256 / tpp;
        long blocks = (P + npl * 4 - 1) / (npl * 4); if (blocks > MAX_BLOCKS) blocks = MAX_BLOCKS; if (blocks < 1) blocks = 1;
        if (vec == 8)      bn_apply_kernel<8><<<(int)blocks, 256, 0, st>>>(r, ldr, scale, shift, y, ldy, P, C, tpp, flags, fin);
        else if (vec == 4) bn_apply_kernel<4><<<(int)blocks, 256, 0, st>>>(r, ldr, scale, shift, y, ldy, P, C, tpp, flags, fin);
        else               bn_apply_kernel<1><<<(int)blocks, 256, 0, st>>>(r, ldr, scale, shift, y, ldy, P, C, tpp, flags, fin);
    } else {
        if (fin.part) return UNET_EINVAL;                                 // (the flat form has no merged finalize: callers check bn_apply_lanes)
        const long total = P * (C / vec);
        long blocks = (total + 255) / 256; if (blocks > 8192) blocks = 8192;
        if (vec == 8)      bn_apply_flat_kernel<8><<<(int)blocks, 256, 0, st>>>(r, ldr, scale, shift, y, ldy, P, C, flags);
        else if (vec == 4) bn_apply_flat_kernel<4><<<(int)blocks, 256, 0, st>>>(r, ldr, scale, shift, y, ldy, P, C, flags);
        else               bn_apply_flat_kernel<1><<<(int)blocks, 256, 0, st>>>(r, ldr, scale, shift, y, ldy, P, C, flags);
    }
    return UNET_LAUNCH_STATUS();
}

}  // namespace

extern "C" size_t unet_bn_workspace(long P, int C) { return (size_t)3 * nblk_for(P) * C * sizeof(double); }

extern "C" int unet_bn_train_stats(const float* r, int ldr, long P, int C, const float* gamma, const float* beta,
        float eps, float momentum, int unbiased_moving_var, float* moving_mean, float* moving_var,
        float* mean, float* invstd, float* scale, float* shift, void* ws, size_t ws_bytes, void* stream) {
    UNET_CHECK_ARG(r && gamma && beta && mean && invstd && scale && shift && ws && P > 0 && C > 0 && ldr >= C);
    UNET_CHECK_ARG((moving_mean == nullptr) == (moving_var == nullptr));
    Plan pl;
    UNET_CHECK_ARG(make_plan(P, C, ldr, 4, 4, unet_aligned16(r), &pl));
    if (ws_bytes < unet_bn_workspace(P, C)) return UNET_ENOSPC;
    hipStream_t st = (hipStream_t)stream;
    double* part = (double*)ws;
    if (pl.vec == 4) bn_stats_kernel<4><<<pl.nblk, 256, pl.smem2, st>>>(r, ldr, P, C, pl.tpp, pl.ppb, part);
    else             bn_stats_kernel<1><<<pl.nblk, 256, pl.smem2, st>>>(r, ldr, P, C, pl.tpp, pl.ppb, part);
    int rc = UNET_LAUNCH_STATUS(); if (rc) return rc;
    bn_train_finalize_kernel<<<C, 64, 0, st>>>(part, pl.nblk, P, C, gamma, beta, eps, momentum,
        unbiased_moving_var, moving_mean, moving_var, mean, invstd, scale, shift);
    return UNET_LAUNCH_STATUS();
}

extern "C" int unet_bn_train_finalize_partials(const float* part, int rows, long P, int C, const float* gamma, const float* beta,
        float eps, float momentum, int unbiased_moving_var, float* moving_mean, float* moving_var,
        float* mean, float* invstd, float* scale, float* shift, void* stream) {
    UNET_CHECK_ARG(part && rows > 0 && P > 0 && C > 0 && C % 64 == 0 && gamma && beta && mean && invstd && scale && shift);
    UNET_CHECK_ARG((moving_mean == nullptr) == (moving_var == nullptr));
    bn_train_finalize_partials_kernel<<<C, 256, 0, (hipStream_t)stream>>>(part, rows, P, C, gamma, beta, eps, momentum,
        unbiased_moving_var, moving_mean, moving_var, mean, invstd, scale, shift);
    return UNET_LAUNCH_STATUS();
}

extern "C" int unet_bn_eval_coeffs(const float* gamma, const float* beta, const float* moving_mean, const float* moving_var,
                                   float eps, int C, float* scale, float* shift, void* stream) {
    UNET_CHECK_ARG(gamma && beta && moving_mean && moving_var && scale && shift && C > 0);
    bn_eval_coeffs_kernel<<<unet_cdiv(C, 128), 128, 0, (hipStream_t)stream>>>(gamma, beta, moving_mean, moving_var, eps, C, scale, shift);
    return UNET_LAUNCH_STATUS();
}

extern "C" int unet_bn_apply(const float* r, int ldr, const float* scale, const float* shift, float* y, int ldy,
                             long P, int C, void* stream) {
    UNET_CHECK_ARG(r && scale && shift && y && P > 0 && P < ((long)1 << 31) && C > 0 && ldr >= C && ldy >= C);
    return launch_bn_apply(r, ldr, 0, scale, shift, y, ldy, 0, nullptr, 0, nullptr, 1, 1, (int)P, C, (hipStream_t)stream);
}

// y = scale * r + shift (as unet_bn_apply) and, in the same pass, pooled = MaxPool2D(2)(y) with the first-max index
extern "C" int unet_bn_apply_maxpool(const float* r, int ldr, const float* scale, const float* shift, float* y, int ldy,
                                     float* pooled, int ldp, uint8_t* idx, int N, int H, int W, int C, void* stream) {
    UNET_CHECK_ARG(r && scale && shift && y && pooled && idx && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && C % 4 == 0);
    UNET_CHECK_ARG(ldr >= C && ldy >= C && ldp >= C && ldr % 4 == 0 && ldy % 4 == 0 && ldp % 4 == 0);
    UNET_CHECK_ARG(unet_aligned16(r) && unet_aligned16(y) && unet_aligned16(pooled) && unet_aligned16(scale) && unet_aligned16(shift) &&
                   (reinterpret_cast<uintptr_t>(idx) & 3u) == 0);
    return launch_bn_apply(r, ldr, 0, scale, shift, y, ldy, 0, pooled, ldp, idx, N, H, W, C, (hipStream_t)stream);
}

static int bn_bwd_launch(const float* dy, int lddy, const float* r, int ldr, const float* gamma, const float* mean,
        const float* invstd, long P, int C, int relu, float* dz, int lddz, float* dgamma, float* dbeta, float* dbias,
        const float* part_sums, int rows, PoolGrad pg, void* ws, size_t ws_bytes, void* stream, int dt = 0, int* host_bias_rows = nullptr) {
    // dt: bit 0 = dz stored as bf16, bit 1 = r stored as bf16, bit 2 = dy stored as bf16 (leading dimensions in elements)
    UNET_CHECK_ARG(dy && r && gamma && mean && invstd && dz && dgamma && dbeta && (dbias || host_bias_rows) && ws && P > 0 && C > 0);
    UNET_CHECK_ARG(lddy >= C && ldr >= C && lddz >= C);
    Plan pl;
    const bool al = unet_aligned16(dy) && unet_aligned16(r) && unet_aligned16(dz) && unet_aligned16(gamma) && unet_aligned16(mean) &&
                    unet_aligned16(invstd) && unet_aligned16(dgamma) && unet_aligned16(dbeta) && (!pg.pdy || (unet_aligned16(pg.pdy) && pg.ldp % 4 == 0));
    UNET_CHECK_ARG(make_plan(P, C, lddy, ldr, lddz, al && (!pg.pdy || pg.ldp % 8 == 0 || !(dt || pg.p16)), &pl, dt != 0 || pg.p16 != 0 || bn_wide_always()));
    if (ws_bytes < unet_bn_workspace(P, C)) return UNET_ENOSPC;
    hipStream_t st = (hipStream_t)stream;
    double* part = (double*)ws;
    int rc;
    if ((dt || pg.p16) && pl.vec < 4) return UNET_EINVAL;
    // grids: the plan's block count, capped to one resident wave of workgroups of the kernel that runs
    int nb_r = pl.nblk, nb_a = pl.nblk;
    {
        const int cr = pl.vec == 8 ? reduce_grid<8>(pl.smem2) : pl.vec == 4 ? reduce_grid<4>(pl.smem2) : reduce_grid<1>(pl.smem2);
        const int ca = pl.vec == 8 ? apply_grid<8>(pl.smem1) : pl.vec == 4 ? apply_grid<4>(pl.smem1) : apply_grid<1>(pl.smem1);
        if (nb_r > cr) nb_r = cr;
        if (nb_a > ca) nb_a = ca;
    }
    const bool pooled_form = pg.pdy != nullptr && pl.vec >= 4;       // window-per-lane kernels: grids over the POOLED pixels
    const long Pw = pooled_form ? P / 4 : P;
    if (pooled_form) {
        const long npl = 256 / pl.tpp;
        long nb = (Pw + npl * 4 - 1) / (npl * 4); if (nb < 1) nb = 1;
        if (nb_r > nb) nb_r = (int)nb;
        if (nb_a > nb) nb_a = (int)nb;
    }
    // all four tensors of a pooled layer stored as bf16 (and below 2 GiB: 32-bit buffer offsets): the packed-register kernels, on grids
    // of their own resident size
    const size_t ldmax = (size_t)(lddy > ldr ? (lddy > lddz ? lddy : lddz) : (ldr > lddz ? ldr : lddz));
    const bool small16 = (size_t)P * ldmax * 2 < ((size_t)1 << 31);
    const bool pool16 = pooled_form && pl.vec == 8 && dt == 7 && pg.p16 && small16 && pg.ldp % 8 == 0;
    if (pool16) {
        const long npl = 256 / pl.tpp;
        long nb = (Pw + npl * 4 - 1) / (npl * 4); if (nb < 1) nb = 1;
        nb_r = pl.nblk; nb_a = pl.nblk;
        const int cr = pool16_grid(0, pl.smem2), ca = pool16_grid(1, pl.smem1);
        if (nb_r > cr) nb_r = cr;
        if (nb_a > ca) nb_a = ca;
        if (nb_r > nb) nb_r = (int)nb;
        if (nb_a > nb) nb_a = (int)nb;
    }
    const long ppb_r = (Pw + nb_r - 1) / nb_r, ppb_a = (Pw + nb_a - 1) / nb_a;
    if (part_sums) {
        bn_bwd_finalize_partials_kernel<<<C, 256, 0, st>>>(part_sums, rows, C, mean, invstd, dgamma, dbeta);
    } else {
        if (pool16) bn_bwd_reduce_pool16_kernel<<<nb_r, 256, pl.smem2, st>>>((const uint16_t*)dy, lddy, (const uint16_t*)r, ldr, mean, P, Pw, C, pl.tpp, ppb_r, part, pg);
        else if (pooled_form && pl.vec == 8) bn_bwd_reduce_pool_kernel<8><<<nb_r, 256, pl.smem2, st>>>(dy, lddy, r, ldr, mean, Pw, C, pl.tpp, ppb_r, part, pg, dt);
        else if (pooled_form)      bn_bwd_reduce_pool_kernel<4><<<nb_r, 256, pl.smem2, st>>>(dy, lddy, r, ldr, mean, Pw, C, pl.tpp, ppb_r, part, pg, dt);
        else if (pl.vec == 8) bn_bwd_reduce_kernel<8><<<nb_r, 256, pl.smem2, st>>>(dy, lddy, r, ldr, mean, invstd, P, C, pl.tpp, ppb_r, part, pg, dt);
        else if (pl.vec == 4) bn_bwd_reduce_kernel<4><<<nb_r, 256, pl.smem2, st>>>(dy, lddy, r, ldr, mean, invstd, P, C, pl.tpp, ppb_r, part, pg, dt);
        else                  bn_bwd_reduce_kernel<1><<<nb_r, 256, pl.smem2, st>>>(dy, lddy, r, ldr, mean, invstd, P, C, pl.tpp, ppb_r, part, pg, 0);
        rc = UNET_LAUNCH_STATUS(); if (rc) return rc;
        bn_bwd_finalize_kernel<<<C, 256, 0, st>>>(part, nb_r, C, invstd, dgamma, dbeta);
    }
    rc = UNET_LAUNCH_STATUS(); if (rc) return rc;
    double* part2 = part + (size_t)2 * MAX_BLOCKS * C;
    if (pool16) {
        bn_bwd_apply_pool16_kernel<<<nb_a, 256, pl.smem1, st>>>((const uint16_t*)dy, lddy, (const uint16_t*)r, ldr, gamma, mean, invstd, dgamma, dbeta,
                                                                P, Pw, C, pl.tpp, ppb_a, relu, (uint16_t*)dz, lddz, part2, pg);
    } else if (dt == 7 && pl.vec == 8 && !pg.pdy && small16) {
        // dy, r and dz all stored as bf16: the packed-register kernel, on a grid of its own resident size
        int nb16 = pl.nblk; const int c16 = apply16_grid(pl.smem1); if (nb16 > c16) nb16 = c16;
        bn_bwd_apply16_kernel<<<nb16, 256, pl.smem1, st>>>((const uint16_t*)dy, lddy, (const uint16_t*)r, ldr, gamma, mean, invstd, dgamma, dbeta,
                                                           P, C, pl.tpp, (P + nb16 - 1) / nb16, relu, (uint16_t*)dz, lddz, part2);
        nb_a = nb16;
    } else
    if (pooled_form && pl.vec == 8) bn_bwd_apply_pool_kernel<8><<<nb_a, 256, pl.smem1, st>>>(dy, lddy, r, ldr, gamma, mean, invstd, dgamma, dbeta, P, Pw, C, pl.tpp, ppb_a, relu, dz, lddz, part2, pg, dt);
    else if (pooled_form)      bn_bwd_apply_pool_kernel<4><<<nb_a, 256, pl.smem1, st>>>(dy, lddy, r, ldr, gamma, mean, invstd, dgamma, dbeta, P, Pw, C, pl.tpp, ppb_a, relu, dz, lddz, part2, pg, dt);
    else if (pl.vec == 8) bn_bwd_apply_kernel<8><<<nb_a, 256, pl.smem1, st>>>(dy, lddy, r, ldr, gamma, mean, invstd, dgamma, dbeta, P, C, pl.tpp, ppb_a, relu, dz, lddz, part2, pg, dt);
    else if (pl.vec == 4) bn_bwd_apply_kernel<4><<<nb_a, 256, pl.smem1, st>>>(dy, lddy, r, ldr, gamma, mean, invstd, dgamma, dbeta, P, C, pl.tpp, ppb_a, relu, dz, lddz, part2, pg, dt);
    else                  bn_bwd_apply_kernel<1><<<nb_a, 256, pl.smem1, st>>>(dy, lddy, r, ldr, gamma, mean, invstd, dgamma, dbeta, P, C, pl.tpp, ppb_a, relu, dz, lddz, part2, pg, 0);
    rc = UNET_LAUNCH_STATUS(); if (rc) return rc;
    if (host_bias_rows) { *host_bias_rows = nb_a; return UNET_OK; }      // the caller finishes the bias gradient (unet_bn_bwd_bias), off the critical chain
    colsum_finalize_kernel<<<C, 256, 0, st>>>(part2, nb_a, C, dbias);
    return UNET_LAUNCH_STATUS();
}

extern "C" int unet_bn_bwd(const float* dy, int lddy, const float* r, int ldr, const float* gamma, const float* mean,
        const float* invstd, long P, int C, int relu, float* dz, int lddz, float* dgamma, float* dbeta, float* dbias,
        void* ws, size_t ws_bytes, void* stream) {
    return bn_bwd_launch(dy, lddy, r, ldr, gamma, mean, invstd, P, C, relu, dz, lddz, dgamma, dbeta, dbias, nullptr, 0,
                         PoolGrad{nullptr, 0, nullptr, 0, 0, 0, 0}, ws, ws_bytes, stream);
}

// unet_bn_bwd for a layer whose output also feeds MaxPool2D(2): the gradient is dy_skip + unpool(pooled_dy) (first-max indices
// idx from the forward pool), formed on the fly -- no separate pool-backward pass.  dy_skip / r / dz are [N,H,W,C].
extern "C" int unet_bn_bwd_pooled(const float* dy_skip, int lddy, const float* pooled_dy, int ldp, const uint8_t* idx, int N, int H, int W,
        const float* r, int ldr, const float* gamma, const float* mean, const float* invstd, int C, int relu,
        float* dz, int lddz, float* dgamma, float* dbeta, float* dbias, void* ws, size_t ws_bytes, void* stream) {
    UNET_CHECK_ARG(pooled_dy && idx && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && ldp >= C);
    return bn_bwd_launch(dy_skip, lddy, r, ldr, gamma, mean, invstd, (long)N * H * W, C, relu, dz, lddz, dgamma, dbeta, dbias, nullptr, 0,
                         PoolGrad{pooled_dy, ldp, idx, H, W, C, 0}, ws, ws_bytes, stream);
}

// unet_bn_bwd with the reduction pass replaced by the partial sums of a fused data-gradient kernel
// (unet_conv3x3_dgrad_winograd_fused_bnstats): part holds (C/64) * rows * 128 floats for exactly these C channels.
extern "C" int unet_bn_bwd_from_partials(const float* dy, int lddy, const float* r, int ldr, const float* gamma, const float* mean,
        const float* invstd, long P, int C, int relu, float* dz, int lddz, float* dgamma, float* dbeta, float* dbias,
        const float* part_sums, int rows, void* ws, size_t ws_bytes, void* stream) {
    UNET_CHECK_ARG(part_sums && rows > 0 && C % 64 == 0);
    return bn_bwd_launch(dy, lddy, r, ldr, gamma, mean, invstd, P, C, relu, dz, lddz, dgamma, dbeta, dbias, part_sums, rows,
                         PoolGrad{nullptr, 0, nullptr, 0, 0, 0, 0}, ws, ws_bytes, stream);
}

// All three forms of the BatchNorm backward in one call, with the choice of storing dz as bf16 (dz_bf16 != 0: `dz` is a bf16
// tensor [P][lddz]; its consumers are the bf16 data / weight gradient kernels, which would round the fp32 values the same way):
// pooled_dy / idx nullable (unet_bn_bwd_pooled when given), part_sums nullable (unet_bn_bwd_from_partials when given).
extern "C" int unet_bn_bwd_any(const void* dy, int lddy, const void* pooled_dy, int ldp, const uint8_t* idx, int N, int H, int W,
        const void* r, int ldr, const float* gamma, const float* mean, const float* invstd, int C, int relu,
        void* dz, int lddz, int dz_bf16, float* dgamma, float* dbeta, float* dbias, const float* part_sums, int rows,
        void* ws, size_t ws_bytes, void* stream, int r_bf16, int dy_bf16, int pooled_dy_bf16, int* host_bias_rows) {
    UNET_CHECK_ARG(N > 0 && H > 0 && W > 0 && (pooled_dy == nullptr) == (idx == nullptr));
    UNET_CHECK_ARG(!pooled_dy || (H % 2 == 0 && W % 2 == 0 && ldp >= C));
    UNET_CHECK_ARG(!part_sums || (rows > 0 && C % 64 == 0));
    const PoolGrad pg = pooled_dy ? PoolGrad{(const float*)pooled_dy, ldp, idx, H, W, C, pooled_dy_bf16 ? 1 : 0} : PoolGrad{nullptr, 0, nullptr, 0, 0, 0, 0};
    return bn_bwd_launch((const float*)dy, lddy, (const float*)r, ldr, gamma, mean, invstd, (long)N * H * W, C, relu, (float*)dz, lddz, dgamma, dbeta, dbias,
                         part_sums, part_sums ? rows : 0, pg, ws, ws_bytes, stream, (dz_bf16 ? 1 : 0) | (r_bf16 ? 2 : 0) | (dy_bf16 ? 4 : 0), host_bias_rows);
}

// The bias gradient sum(dz) of a unet_bn_bwd_any call that was given host_bias_rows: `ws` is that call's workspace (untouched since),
// `rows` the value it returned.  Only the optimizer / the gradient exchange need it, so a caller runs this beside the layer's weight
// gradient instead of in front of its data gradient.
extern "C" int unet_bn_bwd_bias(const void* ws, int rows, int C, float* dbias, void* stream) {
    UNET_CHECK_ARG(ws && dbias && rows > 0 && rows <= MAX_BLOCKS && C > 0);
    colsum_finalize_kernel<<<C, 256, 0, (hipStream_t)stream>>>((const double*)ws + (size_t)2 * MAX_BLOCKS * C, rows, C, dbias);
    return UNET_LAUNCH_STATUS();
}

// BatchNorm apply (+ optional 2x2 max pool: pooled / idx non-null) with either side stored as bf16: r_bf16 (the conv output it
// reads), y_bf16 (what it writes, y and pooled alike)
// unet_bn_train_finalize_partials + unet_bn_apply_any in ONE launch (bn_finalize_then_wait above): same results bit for bit, one launch
// boundary fewer on the forward chain.  `counter`: a device word that only this call sequence touches, zero at first use; `counter_target` =
// the value it must reach = (sum of C over every earlier call on this counter) + C, modulo 2^32 (the caller keeps the running sum).
// Layouts the lane form of the apply kernel does not cover fall back to the two launches (the counter is then advanced by the host's
// bookkeeping only: the kernel adds C through the finalize kernel's stand-in below, so the running sum stays valid).
extern "C" int unet_bn_finalize_apply_any(const float* part, int rows, const float* gamma, const float* beta, float eps, float momentum,
        int unbiased_moving_var, float* moving_mean, float* moving_var, float* mean, float* invstd, uint32_t* counter, uint32_t counter_target,
        const void* r, int ldr, int r_bf16, float* scale, float* shift, void* y, int ldy, int y_bf16,
        void* pooled, int ldp, uint8_t* idx, int N, int H, int W, int C, void* stream) {
    UNET_CHECK_ARG(part && rows > 0 && C > 0 && C % 64 == 0 && gamma && beta && mean && invstd && scale && shift && counter);
    UNET_CHECK_ARG((moving_mean == nullptr) == (moving_var == nullptr));
    UNET_CHECK_ARG(r && y && N > 0 && H > 0 && W > 0 && ldr >= C && ldy >= C && ldr % 4 == 0 && ldy % 4 == 0);
    UNET_CHECK_ARG(unet_aligned16(r) && unet_aligned16(y) && unet_aligned16(scale) && unet_aligned16(shift) && (pooled == nullptr) == (idx == nullptr));
    if (pooled) UNET_CHECK_ARG(H % 2 == 0 && W % 2 == 0 && ldp >= C && ldp % 4 == 0 && unet_aligned16(pooled) && (reinterpret_cast<uintptr_t>(idx) & 7u) == 0);
    const long P = (long)N * H * W;
    const int vec = (C % 8 == 0 && ldr % 8 == 0 && ldy % 8 == 0 && (!pooled || ldp % 8 == 0) && 256 % (C / 8) == 0) ? 8 : 4;
    const bool lanes = 256 % (C / vec) == 0;
    if (!lanes) {        // two launches; the counter still advances by C so that the caller's running sum holds
        bn_train_finalize_partials_kernel<<<C, 256, 0, (hipStream_t)stream>>>(part, rows, P, C, gamma, beta, eps, momentum,
            unbiased_moving_var, moving_mean, moving_var, mean, invstd, scale, shift);
        int rc = UNET_LAUNCH_STATUS(); if (rc) return rc;
        counter_add_kernel<<<1, 1, 0, (hipStream_t)stream>>>(counter, (unsigned)C);
        rc = UNET_LAUNCH_STATUS(); if (rc) return rc;
        return launch_bn_apply((const float*)r, ldr, r_bf16 ? 1 : 0, scale, shift, (float*)y, ldy, y_bf16 ? 1 : 0, (float*)pooled, ldp, idx, N, H, W, C, (hipStream_t)stream);
    }
    const BnFin fin{part, rows, P, gamma, beta, eps, momentum, unbiased_moving_var, moving_mean, moving_var, mean, invstd, counter, counter_target};
    return launch_bn_apply((const float*)r, ldr, r_bf16 ? 1 : 0, scale, shift, (float*)y, ldy, y_bf16 ? 1 : 0, (float*)pooled, ldp, idx,
                           N, H, W, C, (hipStream_t)stream, &fin);
}

extern "C" int unet_bn_apply_any(const void* r, int ldr, int r_bf16, const float* scale, const float* shift, void* y, int ldy, int y_bf16,
                                 void* pooled, int ldp, uint8_t* idx, int N, int H, int W, int C, void* stream) {
    UNET_CHECK_ARG(r && scale && shift && y && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && ldr >= C && ldy >= C && ldr % 4 == 0 && ldy % 4 == 0);
    UNET_CHECK_ARG(unet_aligned16(r) && unet_aligned16(y) && unet_aligned16(scale) && unet_aligned16(shift) && (pooled == nullptr) == (idx == nullptr));
    if (pooled) UNET_CHECK_ARG(H % 2 == 0 && W % 2 == 0 && ldp >= C && ldp % 4 == 0 && unet_aligned16(pooled) && (reinterpret_cast<uintptr_t>(idx) & 7u) == 0);
    return launch_bn_apply((const float*)r, ldr, r_bf16 ? 1 : 0, scale, shift, (float*)y, ldy, y_bf16 ? 1 : 0, (float*)pooled, ldp, idx,
                           N, H, W, C, (hipStream_t)stream);
}

