// Does a raw buffer access with the uniform part of its address in the SCALAR offset behave at large offsets?  (The first packed BatchNorm
// kernel of round 3 stored 0.05 % of its elements wrong at full size with such addressing; conv_bf16.hip's epilogue uses the same mode for
// the saved-activation loads.)  Copy `n` 16-byte items with voffset = lane part, soffset = block part, in blocks of 8 steps like that kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void copy_soff(const int* src, int* dst, long items_per_block, long total_items, int mode) {
    const __amdgpu_buffer_rsrc_t ss = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)(total_items * 16), 0x00020000);
    const __amdgpu_buffer_rsrc_t sd = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, (int)(total_items * 16), 0x00020000);
    const long b0 = (long)blockIdx.x * items_per_block;
    const int v = threadIdx.x * 16;
    for (long base = b0; base + 8 * 256 <= b0 + items_per_block; base += 8 * 256) {
        i32x4 r[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int so = (int)((base + (long)u * 256) * 16);
            r[u] = mode ? __builtin_amdgcn_raw_buffer_load_b128(ss, v + so, 0, 0) : __builtin_amdgcn_raw_buffer_load_b128(ss, v, so, 0);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int so = (int)((base + (long)u * 256) * 16);
            if (mode) __builtin_amdgcn_raw_buffer_store_b128(r[u], sd, v + so, 0, 0); else __builtin_amdgcn_raw_buffer_store_b128(r[u], sd, v, so, 0);
        }
    }
}
int main() {
    for (int mode = 0; mode < 2; ++mode)
        for (long mb : {32L, 256L, 1024L, 2040L}) {
            const long items = mb * 1024 * 1024 / 16, blocks = 1024, ipb = items / blocks;
            int *s, *d;
            (void)hipMalloc(&s, items * 16); (void)hipMalloc(&d, items * 16);
            std::vector<int> h(items * 4);
            for (long i = 0; i < items * 4; ++i) h[i] = (int)(i * 2654435761u);
            (void)hipMemcpy(s, h.data(), items * 16, hipMemcpyHostToDevice); (void)hipMemset(d, 0, items * 16);
            copy_soff<<<blocks, 256>>>(s, d, ipb, items, mode); (void)hipDeviceSynchronize();
            std::vector<int> o(items * 4);
            (void)hipMemcpy(o.data(), d, items * 16, hipMemcpyDeviceToHost);
            long bad = 0, covered = blocks * (ipb / 2048) * 2048 * 4L;
            for (long b = 0; b < blocks; ++b) for (long i = 0; i < (ipb / 2048) * 2048 * 4; ++i) { const long k = b * ipb * 4 + i; bad += o[k] != h[k]; }
            printf("%s  %5ld MB: %ld wrong of %ld ints\n", mode ? "voffset only  " : "scalar offset ", mb, bad, covered);
            (void)hipFree(s); (void)hipFree(d);
        }
    return 0;
}
