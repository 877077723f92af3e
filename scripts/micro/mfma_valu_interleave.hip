// Microbenchmark: does VALU / LDS work placed between v_mfma_f32_32x32x2_f32 instructions slow the matrix pipe?
//   WPS = 1: one wave per SIMD with 16 accumulators (256 AGPRs, as in the fused Winograd kernels)
//   WPS = 2: two waves per SIMD with 8 accumulators each (128 AGPRs)
// Every instruction is inline asm so the order is exactly what is written.  Reports shader cycles per MFMA per SIMD from
// s_memtime next to the wall-clock rate, so clock changes under load are not mistaken for issue stalls.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(i) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b))
#define VADD(r) asm volatile("v_add_f32 %0, %0, %1" : "+v"(t[r]) : "v"(u))
#define DSRD(r) asm volatile("ds_read_b32 %0, %1" : "=v"(l[r]) : "v"(laddr))
template <int WPS, int NV, int NL>
__global__ __launch_bounds__(256 * WPS, 1) void k(float* out, long long* clk, int iters, float a0, float b0) {
    constexpr int NACC = 16 / WPS;
    __shared__ float lds[1024];
    lds[threadIdx.x] = a0; __syncthreads();
    f32x16 acc[NACC];
    for (int x = 0; x < NACC; ++x) for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;
    float a = a0 + threadIdx.x * 0.001f, b = b0 - threadIdx.x * 0.002f, u = a0 * 1e-9f;
    float t[8] = {1, 2, 3, 4, 5, 6, 7, 8}; float l[2] = {0, 0};
    const int laddr = (threadIdx.x & 63) * 4;
    const long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int x = 0; x < NACC; ++x) {
            MFMA(x);
#pragma unroll
            for (int v = 0; v < NV; ++v) VADD(v & 7);
#pragma unroll
            for (int q = 0; q < NL; ++q) DSRD(q & 1);
        }
        if (NL) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float s = l[0] + l[1];
    for (int v = 0; v < 8; ++v) s += t[v];
    for (int x = 0; x < NACC; ++x) for (int r = 0; r < 16; ++r) s += acc[x][r];
    out[blockIdx.x * 256 * WPS + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}
template <int WPS, int NV, int NL> void run() {
    float* d; long long* c; long long h[2];
    (void)hipMalloc(&d, 256 * 512 * 4); (void)hipMalloc(&c, 16);
    const int iters = 8000 ;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<WPS, NV, NL><<<256, 256 * WPS>>>(d, c, 10, 1.f, 2.f); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k<WPS, NV, NL><<<256, 256 * WPS>>>(d, c, iters, 1.f, 2.f); (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
    const double flops = 256.0 * 4 * iters * 16 * 4096.0;
    // s_memtime ticks per MFMA issued on one SIMD; wall_clock64 is 100 MHz
    printf("%d wave/SIMD, per MFMA: %2d VALU %d ds_read  %7.3f ms %6.1f TFLOP/s   memtime ticks/MFMA/SIMD %6.1f  (ticks at %.0f MHz)\n",
           WPS, NV, NL, ms, flops / ms / 1e9, (double)h[0] / (iters * 16.0), (double)h[0] / ((double)h[1] / 100.0));
    (void)hipFree(d); (void)hipFree(c);
}
int main() {
    run<1, 0, 0>(); run<1, 0, 0>(); run<1, 2, 0>(); run<1, 4, 0>(); run<1, 8, 0>(); run<1, 16, 0>(); run<1, 0, 1>(); run<1, 4, 1>();
    run<2, 0, 0>(); run<2, 2, 0>(); run<2, 4, 0>(); run<2, 8, 0>(); run<2, 16, 0>(); run<2, 4, 1>(); run<1, 0, 0>();
    return 0;
}
