// Microbenchmark: cost, in matrix-pipe cycles, of instructions of another class placed between v_mfma_f32_32x32x16_bf16 instructions of
// the SAME wave (one wave per SIMD, 128 accumulator AGPRs in use + 128 idle: the regime of a double-accumulator 64-channel conv kernel).
// Question: does the epilogue's VALU / store work of the previous tile hide in the shadow of the next tile's bf16 MFMAs?
// (fp32 MFMAs share the VALU datapath -- scripts/micro/mfma_issue_cost.hip: ~4 cycles per VALU instruction, never hidden.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
enum { OP_NONE, OP_VADD, OP_PKADD, OP_ACCREAD, OP_CVT, OP_STORE, OP_DSB128, OP_EPI, OP_AND, OP_PERM, OP_PKFMA, OP_MIX5, OP_MIX4PK, OP_GLOAD, OP_ANDS, OP_FMAC, OP_FMA, OP_MIX5S, OP_AND8, OP_OR8, OP_ADD4D, OP_SUBNEW };
template <int OP, int N>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* clk, int iters, float a0) {
    f32x16 acc[8], old[8];
    for (int x = 0; x < 8; ++x) for (int r = 0; r < 16; ++r) { acc[x][r] = 0.f; old[x][r] = a0 * r; }
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (short)(0x3f80 + threadIdx.x); b[e] = (short)(0x3f00 + e); }
    float t[8] = {1, 2, 3, 4, 5, 6, 7, 8}; f32x2 t2[4] = {{1, 2}, {3, 4}, {5, 6}, {7, 8}}, u2 = {a0, a0};
    float u = a0 * 1e-9f, rd[4] = {0, 0, 0, 0};
    unsigned pk[4] = {0, 0, 0, 0};
    __shared__ __attribute__((aligned(1024))) float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = a0;
    __syncthreads();
    const int laddr4 = (threadIdx.x & 63) * 16;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 l4[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    float* dst = out + 65536 + (size_t)blockIdx.x * 4096 + threadIdx.x;
    const char* gsrc = reinterpret_cast<const char*>(out) + (size_t)(blockIdx.x & 63) * 16384 + (threadIdx.x & 63) * 16;
    const long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[x]) : "v"(a), "v"(b));
#pragma unroll
            for (int v = 0; v < N; ++v) {
                if (OP == OP_VADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(t[v & 7]) : "v"(u));
                if (OP == OP_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(t2[v & 3]) : "v"(u2));
                if (OP == OP_ACCREAD) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(rd[v & 3]) : "a"(old[x][v & 15]));
                if (OP == OP_CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[v & 3]) : "v"(t[v & 7]), "v"(t[(v + 1) & 7]));
                if (OP == OP_STORE) asm volatile("global_store_dword %0, %1, off" :: "v"(dst + 256 * (v & 7)), "v"(t[v & 7]) : "memory");
                if (OP == OP_DSB128) asm volatile("ds_read_b128 %0, %1" : "=v"(l4[v & 1]) : "v"(laddr4));
                if (OP == OP_AND) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(pk[v & 3]) : "v"(t[v & 7]));
                if (OP == OP_PERM) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(pk[v & 3]) : "v"(t[v & 7]), "v"(t[(v + 1) & 7]), "s"(0x07060302));
                if (OP == OP_PKFMA) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(t2[v & 3]) : "v"(u2));
                if (OP == OP_MIX5) {     // the BF16x6 split step of one value pair: 2 and, 2 sub, 1 perm
                    float h0, h1;
                    asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(h0) : "v"(t[(2 * v) & 7]));
                    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(t[(2 * v) & 7]) : "v"(t[(2 * v) & 7]), "v"(h0));
                    asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(h1) : "v"(t[(2 * v + 1) & 7]));
                    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(t[(2 * v + 1) & 7]) : "v"(t[(2 * v + 1) & 7]), "v"(h1));
                    asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(pk[v & 3]) : "v"(h1), "v"(h0), "s"(0x07060302));
                }
                if (OP == OP_MIX4PK) {   // the same with one packed subtraction: 2 and, 1 pk_add (neg), 1 perm
                    f32x2 hh;
                    asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(hh[0]) : "v"(t2[v & 3][0]));
                    asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(hh[1]) : "v"(t2[v & 3][1]));
                    asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(t2[v & 3]) : "v"(hh));
                    asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(pk[v & 3]) : "v"(hh[1]), "v"(hh[0]), "s"(0x07060302));
                }
                if (OP == OP_ANDS) asm volatile("v_and_b32_e32 %0, %1, %2" : "=v"(pk[v & 3]) : "s"(0xffff0000u), "v"(t[v & 7]));
                if (OP == OP_FMAC) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(t[v & 7]) : "s"(a0), "v"(u));
                if (OP == OP_FMA) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(t[v & 7]) : "v"(a0), "v"(u));
                if (OP == OP_MIX5S) {    // the split step with the mask in a scalar register (4-byte encodings except the perm)
                    float h0, h1;
                    asm volatile("v_and_b32_e32 %0, %1, %2" : "=v"(h0) : "s"(0xffff0000u), "v"(t[(2 * v) & 7]));
                    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(t[(2 * v) & 7]) : "v"(t[(2 * v) & 7]), "v"(h0));
                    asm volatile("v_and_b32_e32 %0, %1, %2" : "=v"(h1) : "s"(0xffff0000u), "v"(t[(2 * v + 1) & 7]));
                    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(t[(2 * v + 1) & 7]) : "v"(t[(2 * v + 1) & 7]), "v"(h1));
                    asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(pk[v & 3]) : "v"(h1), "v"(h0), "s"(0x07060302));
                }
                if (OP == OP_AND8) asm volatile("v_and_b32_e32 %0, %1, %0" : "+v"(t[v & 7]) : "s"(0xffffff00u));          // eight independent chains, like v_add_f32 above
                if (OP == OP_OR8) asm volatile("v_or_b32_e32 %0, %1, %0" : "+v"(t[v & 7]) : "s"(0x1u));
                if (OP == OP_ADD4D) asm volatile("v_add_f32 %0, %1, %2" : "=v"(rd[v & 3]) : "v"(t[v & 7]), "v"(u));        // four destinations, sources never written: like v_and_b32 above
                if (OP == OP_SUBNEW) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(rd[v & 3]) : "v"(t[v & 7]), "v"(t[(v + 1) & 7]));
                if (OP == OP_GLOAD) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(l4[v & 1]) : "v"(gsrc + 1024 * (v & 7)) : "memory");
                if (OP == OP_EPI) {      // one epilogue item: 2 acc reads, pk add bias, pk max, 2 pk fma-ish stats, cvt, (store every item)
                    float v0, v1;
                    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v0) : "a"(old[x][(2 * v) & 15]));
                    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v1) : "a"(old[x][(2 * v + 1) & 15]));
                    f32x2 vv = {v0, v1};
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(vv) : "v"(u2));
                    asm volatile("v_max_f32 %0, %0, %1" : "+v"(vv[0]) : "v"(u));         // (no packed max on gfx950)
                    asm volatile("v_max_f32 %0, %0, %1" : "+v"(vv[1]) : "v"(u));
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(t2[0]) : "v"(vv));
                    asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(t2[1]) : "v"(vv));
                    unsigned p;
                    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p) : "v"(vv[0]), "v"(vv[1]));
                    asm volatile("global_store_dword %0, %1, off" :: "v"(dst + 256 * (v & 7)), "v"(p) : "memory");
                }
            }
            if (OP == OP_STORE || OP == OP_EPI || OP == OP_GLOAD) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        }
        if (OP == OP_DSB128) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long c1 = __builtin_readcyclecounter();
    float s = rd[0] + rd[1] + rd[2] + rd[3] + (float)(pk[0] + pk[1] + pk[2] + pk[3]) + l4[0][0] + l4[1][3];
    for (int v = 0; v < 8; ++v) s += t[v];
    for (int v = 0; v < 4; ++v) s += t2[v][0] + t2[v][1];
    for (int x = 0; x < 8; ++x) for (int r = 0; r < 16; ++r) s += acc[x][r] + old[x][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = c1 - c0;
}
template <int OP, int N> double run1() {
    float* d; long long* c; long long h;
    (void)hipMalloc(&d, (65536 + 256 * 4096 + 4096) * 4); (void)hipMalloc(&c, 16);
    const int iters = 2000;
    k<OP, N><<<256, 256>>>(d, c, 10, 1.f); (void)hipDeviceSynchronize();
    k<OP, N><<<256, 256>>>(d, c, iters, 1.f); (void)hipDeviceSynchronize();
    (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    (void)hipFree(d); (void)hipFree(c);
    return (double)h / (iters * 8.0);
}
template <int OP> void run(const char* name) {
    const double c1 = run1<OP, 1>(), c2 = run1<OP, 2>(), c4 = run1<OP, 4>(), c8 = run1<OP, 8>();
    printf("%-28s cycles per MFMA with 1/2/4/8 of them after each MFMA: %6.1f %6.1f %6.1f %6.1f\n", name, c1, c2, c4, c8);
}
int main() {
    printf("bare bf16 MFMA stream (32x32x16): %.1f cycles per MFMA\n", run1<OP_NONE, 0>());
    run<OP_VADD>("v_add_f32"); run<OP_PKADD>("v_pk_add_f32"); run<OP_ACCREAD>("v_accvgpr_read_b32"); run<OP_CVT>("v_cvt_pk_bf16_f32");
    run<OP_DSB128>("ds_read_b128"); run<OP_STORE>("global_store_dword");
    run<OP_AND>("v_and_b32"); run<OP_PERM>("v_perm_b32"); run<OP_PKFMA>("v_pk_fma_f32"); run<OP_GLOAD>("global_load_dwordx4");
    run<OP_AND8>("v_and_b32 in place, 8 chains"); run<OP_OR8>("v_or_b32 in place, 8 chains"); run<OP_ADD4D>("v_add_f32 into 4 registers"); run<OP_SUBNEW>("v_sub_f32 into 4 registers");
    run<OP_ANDS>("v_and_b32_e32 (scalar mask)"); run<OP_FMA>("v_fma_f32 (VOP3)"); run<OP_FMAC>("v_fmac_f32_e32"); run<OP_MIX5S>("split pair, 4-byte ands");
    run<OP_MIX5>("split pair: 2 and 2 sub 1 perm"); run<OP_MIX4PK>("split pair: 2 and 1 pk_sub 1 perm");
    printf("epilogue item (2 accvgpr_read + 3 packed VALU + 2 max + cvt + 4-byte store):\n");
    run<OP_EPI>("  items per MFMA");
    return 0;
}
