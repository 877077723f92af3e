// Microbenchmark: v_mfma_f32_32x32x2_f32 issue rate vs accumulator access pattern (one wave per SIMD, operands in VGPRs).
//   pattern 0: 16 accumulators, round robin (each revisited after 16 MFMAs)      -- fused wgrad kernel's inner loop
//   pattern 1: 16 accumulators, 4 consecutive MFMAs per accumulator              -- fused forward kernel's inner loop
//   pattern 2:  4 accumulators, round robin                                       -- classic GEMM tile
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int PATTERN>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[16];
    for (int x = 0; x < 16; ++x) for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;
    float a[4], b[4];
    for (int s = 0; s < 4; ++s) { a[s] = a0 + threadIdx.x * 0.001f + s; b[s] = b0 - threadIdx.x * 0.002f + s; }
    for (int it = 0; it < iters; ++it) {
        if (PATTERN == 0) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int x = 0; x < 16; ++x) acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc[x], 0, 0, 0);
        } else if (PATTERN == 1) {
#pragma unroll
            for (int x = 0; x < 16; ++x)
#pragma unroll
                for (int s = 0; s < 4; ++s) acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc[x], 0, 0, 0);
        } else {
#pragma unroll
            for (int rep = 0; rep < 16; ++rep)
#pragma unroll
                for (int x = 0; x < 4; ++x) acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rep & 3], b[x], acc[x], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int x = 0; x < 16; ++x) for (int r = 0; r < 16; ++r) s += acc[x][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int P> void run(const char* name) {
    float* d; hipMalloc(&d, 256 * 256 * 4);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<P><<<256, 256>>>(d, 10, 1.f, 2.f); hipDeviceSynchronize();
    hipEventRecord(e0); k<P><<<256, 256>>>(d, iters, 1.f, 2.f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * 4 * iters * 64 * 4096.0;
    printf("%-48s %8.3f ms  %7.1f TFLOP/s\n", name, ms, flops / ms / 1e9);
    hipFree(d);
}
int main() {
    run<0>("16 accumulators round robin");
    run<1>("16 accumulators, 4 consecutive MFMAs each");
    run<2>("4 accumulators round robin");
    run<0>("16 accumulators round robin (again)");
    return 0;
}
