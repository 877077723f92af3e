// Microbenchmark: cost, in matrix-pipe cycles, of one instruction of another class placed between v_mfma_f32_32x32x2_f32
// instructions of the same wave (one wave per SIMD, 256 accumulator AGPRs - the fused Winograd kernels' regime).
// Everything is inline asm so the order is exactly what is written; cycles are s_memtime ticks (shader clock).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
enum { OP_VADD, OP_PKADD, OP_DSB32, OP_DSB128, OP_SALU, OP_VMOV, OP_DS2ST64, OP_NOP1, OP_DMA };
__device__ float g_src[64 * 4 * 16];
template <int OP, int N>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* clk, int iters, float a0, float b0) {
    __shared__ __attribute__((aligned(1024))) float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = a0;
    __syncthreads();
    f32x16 acc[16];
    for (int x = 0; x < 16; ++x) for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;
    float a = a0 + threadIdx.x * 0.001f, b = b0 - threadIdx.x * 0.002f, u = a0 * 1e-9f;
    float t[8] = {1, 2, 3, 4, 5, 6, 7, 8}; f32x2 t2[4] = {{1, 2}, {3, 4}, {5, 6}, {7, 8}}, u2 = {u, u};
    float l[4] = {0, 0, 0, 0}; f32x4 l4[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}}; f32x2 l2[2] = {{0, 0}, {0, 0}};
    int sc[4] = {0, 0, 0, 0};
    const int laddr = (threadIdx.x & 63) * 4, laddr4 = (threadIdx.x & 63) * 16;
    const float* gsrc = g_src + (threadIdx.x & 63) * 4;
    const int wv = threadIdx.x >> 6;
    const long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int x = 0; x < 16; ++x) {
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[x]) : "v"(a), "v"(b));
#pragma unroll
            for (int v = 0; v < N; ++v) {
                if (OP == OP_VADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(t[v & 7]) : "v"(u));
                if (OP == OP_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(t2[v & 3]) : "v"(u2));
                if (OP == OP_DSB32) asm volatile("ds_read_b32 %0, %1" : "=v"(l[v & 3]) : "v"(laddr));
                if (OP == OP_DSB128) asm volatile("ds_read_b128 %0, %1" : "=v"(l4[v & 1]) : "v"(laddr4));
                if (OP == OP_DS2ST64) asm volatile("ds_read2st64_b32 %0, %1 offset0:1 offset1:2" : "=v"(l2[v & 1]) : "v"(laddr));
                if (OP == OP_SALU) asm volatile("s_add_i32 %0, %0, 1" : "+s"(sc[v & 3]));
                if (OP == OP_VMOV) asm volatile("v_mov_b32 %0, %1" : "=v"(t[v & 7]) : "v"(u));
                if (OP == OP_NOP1) asm volatile("s_nop 1");
                if (OP == OP_DMA) __builtin_amdgcn_global_load_lds(gsrc, (__attribute__((address_space(3))) void*)(lds + wv * 1024 + (v & 3) * 256), 16, 0, 0);
            }
            if (OP == OP_DMA) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
        if (OP == OP_DSB32 || OP == OP_DSB128 || OP == OP_DS2ST64) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long c1 = __builtin_readcyclecounter();
    float s = l[0] + l[1] + l[2] + l[3] + l4[0][0] + l4[1][3] + l2[0][0] + l2[1][1] + (float)(sc[0] + sc[1] + sc[2] + sc[3]);
    for (int v = 0; v < 8; ++v) s += t[v];
    for (int v = 0; v < 4; ++v) s += t2[v][0] + t2[v][1];
    for (int x = 0; x < 16; ++x) for (int r = 0; r < 16; ++r) s += acc[x][r];
    out[blockIdx.x * 256 + threadIdx.x] = s + lds[threadIdx.x];
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = c1 - c0;
}
template <int OP, int N> double run1() {
    float* d; long long* c; long long h;
    (void)hipMalloc(&d, 256 * 256 * 4); (void)hipMalloc(&c, 16);
    const int iters = 4000;
    k<OP, N><<<256, 256>>>(d, c, 10, 1.f, 2.f); (void)hipDeviceSynchronize();
    k<OP, N><<<256, 256>>>(d, c, iters, 1.f, 2.f); (void)hipDeviceSynchronize();
    (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    (void)hipFree(d); (void)hipFree(c);
    return (double)h / (iters * 16.0);
}
template <int OP> void run(const char* name) {
    const double c1 = run1<OP, 1>(), c2 = run1<OP, 2>(), c4 = run1<OP, 4>(), c8 = run1<OP, 8>();
    printf("%-22s cycles per MFMA with 1/2/4/8 of them after each MFMA: %6.1f %6.1f %6.1f %6.1f   (marginal %.1f per instruction)\n",
           name, c1, c2, c4, c8, (c8 - c4) / 4.0);
}
int main() {
    printf("bare MFMA stream: %.1f cycles per MFMA\n", run1<OP_NOP1, 0>());
    run<OP_VADD>("v_add_f32"); run<OP_PKADD>("v_pk_add_f32"); run<OP_VMOV>("v_mov_b32"); run<OP_SALU>("s_add_i32");
    run<OP_NOP1>("s_nop 1"); run<OP_DSB32>("ds_read_b32"); run<OP_DS2ST64>("ds_read2st64_b32"); run<OP_DSB128>("ds_read_b128");
    run<OP_DMA>("global_load_lds x4");
    return 0;
}
