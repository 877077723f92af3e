// Microbenchmark for DESIGN.md lead (g): an fp32 product block on the bf16 matrix pipe with three-piece operands.
//   x = hi + mid + lo, three bf16 pieces (nearest-even each, remainders exact in fp32); six piece products hh, hm, mh, hl, lh, mm, exact in the
//   fp32 accumulator of v_mfma_f32_32x32x16_bf16.  Compared with v_mfma_f32_32x32x2_f32 on the same operands against an fp64 host reference
//   (C = A[32 x K] * B[K x 32]), plus the three-product form (hh, hm, mh: "bf16x3") to show what is NOT enough.  Second part: matrix-pipe
//   cycles per 32 x 32 x 16 block of multiply-adds, operands in registers (8 fp32 MFMAs vs 6 bf16 MFMAs), one wave per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 -o bf16x6_gemm bf16x6_gemm.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short rne_bf16(float x) {
    unsigned r; asm("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(r) : "v"(x)); return (unsigned short)(r & 0xffffu);
}
__device__ __forceinline__ float bf16_f32(unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
__device__ __forceinline__ void split3(float x, unsigned short& hi, unsigned short& mid, unsigned short& lo) {
    hi = rne_bf16(x); const float r1 = x - bf16_f32(hi);
    mid = rne_bf16(r1); const float r2 = r1 - bf16_f32(mid);
    lo = rne_bf16(r2);
}

// mode 0: fp32 MFMA; 1: six products; 2: three products (hh, hm, mh)
__global__ void gemm32(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int K, int mode) {
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    f32x16 acc; for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    if (mode == 0) {
        for (int s = 0; s < K / 2; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + 2 * s + h], B[(2 * s + h) * 32 + i], acc, 0, 0, 0);
    } else {
        for (int s = 0; s < K / 16; ++s) {
            bf16x8 a[3], b[3];
            for (int e = 0; e < 8; ++e) {
                unsigned short p0, p1, p2;
                split3(A[i * K + 16 * s + 8 * h + e], p0, p1, p2); a[0][e] = (short)p0; a[1][e] = (short)p1; a[2][e] = (short)p2;
                split3(B[(16 * s + 8 * h + e) * 32 + i], p0, p1, p2); b[0][e] = (short)p0; b[1][e] = (short)p1; b[2][e] = (short)p2;
            }
            // smallest terms first
            if (mode == 1) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
        }
    }
    for (int e = 0; e < 16; ++e) C[((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + i] = acc[e];
}

// matrix-pipe time: `blocks` 32 x 32 x 16 multiply-add blocks per wave and iteration into 4 independent accumulators
template <int MODE>
__global__ __launch_bounds__(256, 1) void rate(float* out, long long* clk, int iters, float a0) {
    f32x16 acc[4]; for (int x = 0; x < 4; ++x) for (int e = 0; e < 16; ++e) acc[x][e] = 0.f;
    bf16x8 a, b; for (int e = 0; e < 8; ++e) { a[e] = (short)(0x3f80 + threadIdx.x); b[e] = (short)(0x3f00 + e); }
    const float fa = a0 + threadIdx.x, fb = a0 * 0.5f;
    const long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            if (MODE == 0) {
#pragma unroll
                for (int r = 0; r < 8; ++r) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[x]) : "v"(fa), "v"(fb));
            } else {
#pragma unroll
                for (int r = 0; r < 6; ++r) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[x]) : "v"(a), "v"(b));
            }
        }
    }
    const long long c1 = __builtin_readcyclecounter();
    float s = 0.f; for (int x = 0; x < 4; ++x) for (int e = 0; e < 16; ++e) s += acc[x][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = c1 - c0;
}

int main() {
    const int K = 1024;
    std::vector<float> A(32 * K), B(K * 32);
    srand(1);
    for (auto& v : A) v = (float)((rand() / (double)RAND_MAX - 0.5) * 4.0);
    for (auto& v : B) v = (float)((rand() / (double)RAND_MAX - 0.5) * 4.0);
    std::vector<double> ref(1024, 0.0);
    double scale = 0.0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double s = 0, m = 0; for (int k = 0; k < K; ++k) { s += (double)A[i * K + k] * B[k * 32 + j]; m += fabs((double)A[i * K + k] * B[k * 32 + j]); } ref[i * 32 + j] = s; scale = fmax(scale, m); }
    float *dA, *dB, *dC; hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 4096);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    const char* names[3] = {"v_mfma_f32_32x32x2_f32 (fp32 operands)", "bf16 pieces, six products (hh hm mh hl lh mm)", "bf16 pieces, three products (hh hm mh)"};
    printf("C = A[32 x %d] * B[%d x 32], operands uniform in (-2, 2); error against the fp64 product, relative to sum |a b| of the element (%.1f at most)\n", K, K, scale);
    for (int mode = 0; mode < 3; ++mode) {
        gemm32<<<1, 64>>>(dA, dB, dC, K, mode);
        std::vector<float> C(1024); hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
        double worst = 0, rms = 0;
        for (int e = 0; e < 1024; ++e) { const double d = fabs((double)C[e] - ref[e]) / scale; worst = fmax(worst, d); rms += d * d; }
        printf("  %-48s max %.3e  rms %.3e   (2^-24 = %.3e)\n", names[mode], worst, sqrt(rms / 1024), ldexp(1.0, -24));
    }
    float* dOut; long long* dClk; hipMalloc(&dOut, 1024 * 256 * 4); hipMalloc(&dClk, 1024 * 8);
    int dev = 0; hipDeviceProp_t pr; hipGetDeviceProperties(&pr, dev);
    const int grid = pr.multiProcessorCount, iters = 2000;
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) { if (mode == 0) rate<0><<<grid, 256>>>(dOut, dClk, iters, 1.f); else rate<1><<<grid, 256>>>(dOut, dClk, iters, 1.f); }
        hipDeviceSynchronize();
        std::vector<long long> c(grid); hipMemcpy(c.data(), dClk, grid * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : c) s += (double)v;
        printf("  %-28s %.1f shader-clock-counter ticks per 32 x 32 x 16 block and wave (every CU busy, one wave per SIMD)\n", mode ? "6 x v_mfma_f32_32x32x16_bf16" : "8 x v_mfma_f32_32x32x2_f32", s / grid / iters / 4);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        hipEventRecord(e0);
        if (mode == 0) rate<0><<<grid, 256>>>(dOut, dClk, iters, 1.f); else rate<1><<<grid, 256>>>(dOut, dClk, iters, 1.f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double blocks = (double)grid * 4 * iters * 4;            // waves x iterations x accumulators
        printf("  %-28s %.3f ms for %.0f blocks: %.1f T multiply-adds/s = %.0f TFLOP/s fp32-grade\n", mode ? "6 x v_mfma_f32_32x32x16_bf16" : "8 x v_mfma_f32_32x32x2_f32", ms, blocks,
               blocks * 32 * 32 * 16 / ms / 1e9, 2 * blocks * 32 * 32 * 16 / ms / 1e9);
    }
    return 0;
}
