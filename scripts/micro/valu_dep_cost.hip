// Microbenchmark: what EIGHT vector-ALU instructions between two v_mfma_f32_32x32x16_bf16 of the same wave cost (one wave per SIMD),
// as a function of how they depend on each other.  Everything is inline asm with named registers, so the pattern is exactly what is written.
//   A  eight independent adds, eight sources, eight destinations          B  the same into four destinations (write-after-write four apart)
//   C  four producer -> consumer pairs back to back (read-after-write distance 1)      D  the pairs interleaved two deep (distance 2)
//   E  interleaved four deep (distance 4)       F  eight in-place chains (each instruction reads its own result of the previous MFMA gap)
//   G  the BF16x6 split of a value pair as the kernel has it (and, sub, and, sub, perm)  H  the same with the two values interleaved
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
template <int PAT, int REP>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* clk, int iters, float a0) {
    f32x16 acc[8];
    for (int x = 0; x < 8; ++x) for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (short)(0x3f80 + threadIdx.x); b[e] = (short)(0x3f00 + e); }
    float u = a0 * 1e-9f;
    asm volatile("v_mov_b32 v40, %0\n v_mov_b32 v41, %0\n v_mov_b32 v42, %0\n v_mov_b32 v43, %0\n v_mov_b32 v44, %0\n v_mov_b32 v45, %0\n v_mov_b32 v46, %0\n v_mov_b32 v47, %0\n"
                 "v_mov_b32 v48, %0\n v_mov_b32 v49, %0\n v_mov_b32 v50, %0\n v_mov_b32 v51, %0\n v_mov_b32 v52, %0\n v_mov_b32 v53, %0\n v_mov_b32 v54, %0\n v_mov_b32 v55, %0\n v_mov_b32 v56, %1\n"
                 :: "v"(a0), "v"(u) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56");
    const unsigned mask = 0xffff0000u, sel = 0x07060302u;
    const long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[x]) : "v"(a), "v"(b));
#pragma unroll
            for (int r = 0; r < REP; ++r) {
            if (PAT == 0) asm volatile("v_add_f32 v48, v40, v56\n v_add_f32 v49, v41, v56\n v_add_f32 v50, v42, v56\n v_add_f32 v51, v43, v56\n v_add_f32 v52, v44, v56\n v_add_f32 v53, v45, v56\n v_add_f32 v54, v46, v56\n v_add_f32 v55, v47, v56" ::: "memory");
            if (PAT == 1) asm volatile("v_add_f32 v48, v40, v56\n v_add_f32 v49, v41, v56\n v_add_f32 v50, v42, v56\n v_add_f32 v51, v43, v56\n v_add_f32 v48, v44, v56\n v_add_f32 v49, v45, v56\n v_add_f32 v50, v46, v56\n v_add_f32 v51, v47, v56" ::: "memory");
            if (PAT == 2) asm volatile("v_add_f32 v48, v40, v56\n v_add_f32 v49, v48, v56\n v_add_f32 v50, v41, v56\n v_add_f32 v51, v50, v56\n v_add_f32 v52, v42, v56\n v_add_f32 v53, v52, v56\n v_add_f32 v54, v43, v56\n v_add_f32 v55, v54, v56" ::: "memory");
            if (PAT == 3) asm volatile("v_add_f32 v48, v40, v56\n v_add_f32 v50, v41, v56\n v_add_f32 v49, v48, v56\n v_add_f32 v51, v50, v56\n v_add_f32 v52, v42, v56\n v_add_f32 v54, v43, v56\n v_add_f32 v53, v52, v56\n v_add_f32 v55, v54, v56" ::: "memory");
            if (PAT == 4) asm volatile("v_add_f32 v48, v40, v56\n v_add_f32 v50, v41, v56\n v_add_f32 v52, v42, v56\n v_add_f32 v54, v43, v56\n v_add_f32 v49, v48, v56\n v_add_f32 v51, v50, v56\n v_add_f32 v53, v52, v56\n v_add_f32 v55, v54, v56" ::: "memory");
            if (PAT == 5) asm volatile("v_add_f32 v40, v40, v56\n v_add_f32 v41, v41, v56\n v_add_f32 v42, v42, v56\n v_add_f32 v43, v43, v56\n v_add_f32 v44, v44, v56\n v_add_f32 v45, v45, v56\n v_add_f32 v46, v46, v56\n v_add_f32 v47, v47, v56" ::: "memory");
            if (PAT == 6) asm volatile("v_and_b32 v48, %0, v40\n v_sub_f32 v40, v40, v48\n v_and_b32 v49, %0, v41\n v_sub_f32 v41, v41, v49\n v_perm_b32 v52, v49, v48, %1\n"
                                       "v_and_b32 v50, %0, v42\n v_sub_f32 v42, v42, v50\n v_and_b32 v51, %0, v43\n v_sub_f32 v43, v43, v51\n v_perm_b32 v53, v51, v50, %1" :: "s"(mask), "s"(sel) : "memory");
            if (PAT == 8) asm volatile("v_add_f32 v48, v40, v56\n v_add_f32 v49, v41, v56\n v_add_f32 v50, v42, v56\n v_add_f32 v51, v43, v56\n v_add_f32 v52, v44, v56" ::: "memory");
            if (PAT == 9) asm volatile("v_add_f32 v48, v40, v56\n s_nop 0\n v_add_f32 v49, v41, v56\n v_add_f32 v50, v42, v56\n s_nop 0\n v_add_f32 v51, v43, v56\n v_add_f32 v52, v44, v56" ::: "memory");
            if (PAT == 10) asm volatile("v_add_f32 v48, v40, v56\n s_waitcnt lgkmcnt(2)\n v_add_f32 v49, v41, v56\n v_add_f32 v50, v42, v56\n s_waitcnt vmcnt(8)\n v_add_f32 v51, v43, v56\n v_add_f32 v52, v44, v56" ::: "memory");
            if (PAT == 11) asm volatile("v_add_f32 v48, v40, v56\n s_add_u32 s40, s41, 0x1000\n v_add_f32 v49, v41, v56\n v_add_f32 v50, v42, v56\n s_add_u32 s42, s41, 0x2000\n v_add_f32 v51, v43, v56\n v_add_f32 v52, v44, v56" ::: "memory", "s40", "s41", "s42", "scc");
            if (PAT == 12) asm volatile("v_add_f32 v48, v40, v56\n v_add_f32 v49, v41, v56\n v_add_f32 v50, v42, v56\n v_add_f32 v51, v43, v56" ::: "memory");
            if (PAT == 13) asm volatile("v_add_f32 v48, v40, v56\n v_add_f32 v49, v41, v56\n v_add_f32 v50, v42, v56\n v_add_f32 v51, v43, v56\n v_add_f32 v52, v44, v56\n v_add_f32 v53, v45, v56" ::: "memory");
            if (PAT == 14) asm volatile("v_accvgpr_read_b32 v48, a240\n v_accvgpr_read_b32 v49, a241\n v_accvgpr_read_b32 v50, a242\n v_accvgpr_read_b32 v51, a243\n v_accvgpr_read_b32 v52, a244\n v_accvgpr_read_b32 v53, a245\n v_accvgpr_read_b32 v54, a246\n v_accvgpr_read_b32 v55, a247" ::: "memory");
            if (PAT == 7) asm volatile("v_and_b32 v48, %0, v40\n v_and_b32 v49, %0, v41\n v_and_b32 v50, %0, v42\n v_and_b32 v51, %0, v43\n v_sub_f32 v40, v40, v48\n v_sub_f32 v41, v41, v49\n"
                                       "v_sub_f32 v42, v42, v50\n v_sub_f32 v43, v43, v51\n v_perm_b32 v52, v49, v48, %1\n v_perm_b32 v53, v51, v50, %1" :: "s"(mask), "s"(sel) : "memory");
            }
        }
    }
    const long long c1 = __builtin_readcyclecounter();
    float s = 0.f, tmp;
    asm volatile("v_add_f32 %0, v48, v49\n v_add_f32 %0, %0, v50\n v_add_f32 %0, %0, v51\n v_add_f32 %0, %0, v52\n v_add_f32 %0, %0, v53\n v_add_f32 %0, %0, v54\n v_add_f32 %0, %0, v55\n v_add_f32 %0, %0, v40" : "=v"(tmp));
    s += tmp;
    for (int x = 0; x < 8; ++x) for (int r = 0; r < 16; ++r) s += acc[x][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = c1 - c0;
}
template <int PAT, int REP> double run1() {
    float* d; long long* c; long long h;
    (void)hipMalloc(&d, 256 * 256 * 4); (void)hipMalloc(&c, 16);
    const int iters = 2000;
    k<PAT, REP><<<256, 256>>>(d, c, 10, 1.f); (void)hipDeviceSynchronize();
    k<PAT, REP><<<256, 256>>>(d, c, iters, 1.f); (void)hipDeviceSynchronize();
    (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    (void)hipFree(d); (void)hipFree(c);
    return (double)h / (iters * 8.0);
}
template <int PAT> void run(const char* name) { printf("%-66s %6.1f  %6.1f cycles per MFMA with the group once / twice per gap\n", name, run1<PAT, 1>(), run1<PAT, 2>()); }
int main() {
    run<0>("A  8 independent adds (8 sources, 8 destinations)");
    run<1>("B  8 adds into 4 destinations (write-after-write 4 apart)");
    run<2>("C  4 producer->consumer pairs back to back (distance 1)");
    run<3>("D  pairs interleaved two deep (distance 2)");
    run<4>("E  pairs interleaved four deep (distance 4)");
    run<5>("F  8 in-place chains (distance 8 + one MFMA)");
    run<6>("G  split of two value pairs, as compiled (10 instructions)");
    run<7>("H  the same, ands first then subs then perms");
    run<14>("M  8 v_accvgpr_read_b32 (registers no MFMA of the loop writes)");
    run<12>("I4 4 independent adds");
    run<8>("I  5 independent adds");
    run<13>("I6 6 independent adds");
    run<9>("J  5 adds + 2 s_nop 0");
    run<10>("K  5 adds + 2 s_waitcnt (satisfied)");
    run<11>("L  5 adds + 2 s_add_u32");
    return 0;
}
