// Microbenchmark + check: the BF16x6 remainder step  a = v - bf16_trunc(v)  as ONE v_dot2c_f32_bf16 on the already PACKED pieces
// (a += hp.lo * (-1) + hp.hi * 0) instead of v_and_b32 + v_sub_f32.  (1) bit-exact against and + sub over random and edge-case values;
// (2) cycles per v_mfma_f32_32x32x16_bf16 with the split of two value pairs behind each MFMA, both forms (named registers, exact order).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
__global__ void check(const float* v, unsigned* out, int n) {
    const int i = 2 * (blockIdx.x * blockDim.x + threadIdx.x);
    if (i + 1 >= n) return;
    float v0 = v[i], v1 = v[i + 1];
    const unsigned sel = 0x07060302u, clo = 0x0000bf80u, chi = 0xbf800000u;
    unsigned hp, mp, lp; float a0 = v0, a1 = v1;
    asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(hp) : "v"(v1), "v"(v0), "s"(sel));
    asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a0) : "v"(hp), "v"(clo));
    asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a1) : "v"(hp), "v"(chi));
    asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(mp) : "v"(a1), "v"(a0), "s"(sel));
    float b0 = a0, b1 = a1;
    asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(b0) : "v"(mp), "v"(clo));
    asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(b1) : "v"(mp), "v"(chi));
    asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(lp) : "v"(b1), "v"(b0), "s"(sel));
    out[3 * (i / 2)] = hp; out[3 * (i / 2) + 1] = mp; out[3 * (i / 2) + 2] = lp;
}
template <int PAT, int REP>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* clk, int iters, float a0) {
    f32x16 acc[8];
    for (int x = 0; x < 8; ++x) for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (short)(0x3f80 + threadIdx.x); b[e] = (short)(0x3f00 + e); }
    asm volatile("v_mov_b32 v40, %0\n v_mov_b32 v41, %0\n v_mov_b32 v42, %0\n v_mov_b32 v43, %0\n v_mov_b32 v44, %0\n v_mov_b32 v45, %0\n v_mov_b32 v46, %0\n v_mov_b32 v47, %0\n"
                 "v_mov_b32 v48, %0\n v_mov_b32 v49, %0\n v_mov_b32 v50, %0\n v_mov_b32 v51, %0\n v_mov_b32 v52, %0\n v_mov_b32 v53, %0\n v_mov_b32 v54, %0\n v_mov_b32 v55, %0\n v_mov_b32 v56, 0xbf80\n v_mov_b32 v57, 0xbf800000\n"
                 :: "v"(a0) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57");
    const unsigned mask = 0xffff0000u, sel = 0x07060302u;
    const long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[x]) : "v"(a), "v"(b));
#pragma unroll
            for (int r = 0; r < REP; ++r) {
                // one full split of a value pair (v40, v41): h, m, l packs
                if (PAT == 0) asm volatile("v_perm_b32 v52, v41, v40, %1\n v_and_b32 v48, %0, v40\n v_and_b32 v49, %0, v41\n v_sub_f32 v42, v40, v48\n v_sub_f32 v43, v41, v49\n v_perm_b32 v53, v43, v42, %1\n"
                                           "v_and_b32 v48, %0, v42\n v_and_b32 v49, %0, v43\n v_sub_f32 v44, v42, v48\n v_sub_f32 v45, v43, v49\n v_perm_b32 v54, v45, v44, %1" :: "s"(mask), "s"(sel) : "memory");
                if (PAT == 1) asm volatile("v_perm_b32 v52, v41, v40, %0\n v_mov_b32 v42, v40\n v_mov_b32 v43, v41\n v_dot2c_f32_bf16 v42, v52, v56\n v_dot2c_f32_bf16 v43, v52, v57\n v_perm_b32 v53, v43, v42, %0\n"
                                           "v_dot2c_f32_bf16 v42, v53, v56\n v_dot2c_f32_bf16 v43, v53, v57\n v_perm_b32 v54, v43, v42, %0" :: "s"(sel) : "memory");
                if (PAT == 2) asm volatile("v_perm_b32 v52, v41, v40, %0\n v_dot2c_f32_bf16 v40, v52, v56\n v_dot2c_f32_bf16 v41, v52, v57\n v_perm_b32 v53, v41, v40, %0\n"
                                           "v_dot2c_f32_bf16 v40, v53, v56\n v_dot2c_f32_bf16 v41, v53, v57\n v_perm_b32 v54, v41, v40, %0" :: "s"(sel) : "memory");
            }
        }
    }
    const long long c1 = __builtin_readcyclecounter();
    float s = 0.f, tmp;
    asm volatile("v_add_f32 %0, v52, v53\n v_add_f32 %0, %0, v54\n v_add_f32 %0, %0, v40" : "=v"(tmp));
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
static unsigned fbits(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static float bitsf(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
int main() {
    const int n = 1 << 22;
    float* hv = (float*)malloc(n * 4);
    srand(1);
    for (int i = 0; i < n; ++i) {
        unsigned u = ((unsigned)rand() << 16) ^ (unsigned)rand();
        if (i % 5 == 0) u = (u & 0x807fffffu) | ((unsigned)(rand() % 40) << 23);          // tiny magnitudes incl. denormals
        if (i % 7 == 0) u = (u & 0x807fffffu) | ((unsigned)(200 + rand() % 55) << 23);    // huge magnitudes
        if (((u >> 23) & 0xff) == 0xff) u &= 0xbfffffffu;                                 // no inf / nan
        hv[i] = bitsf(u);
    }
    hv[0] = 0.f; hv[1] = -0.f; hv[2] = 1.f; hv[3] = -1.f; hv[4] = bitsf(0x00000001u); hv[5] = bitsf(0x007fffffu); hv[6] = bitsf(0x7f7fffffu); hv[7] = bitsf(0x00800000u);
    float* dv; unsigned* dout; (void)hipMalloc(&dv, n * 4); (void)hipMalloc(&dout, (size_t)n / 2 * 3 * 4);
    (void)hipMemcpy(dv, hv, n * 4, hipMemcpyHostToDevice);
    check<<<n / 2 / 256, 256>>>(dv, dout, n); (void)hipDeviceSynchronize();
    unsigned* ho = (unsigned*)malloc((size_t)n / 2 * 3 * 4);
    (void)hipMemcpy(ho, dout, (size_t)n / 2 * 3 * 4, hipMemcpyDeviceToHost);
    long bad = 0, bad_normal = 0;
    for (int i = 0; i < n; i += 2) {
        unsigned want[3];
        float a[2] = {hv[i], hv[i + 1]};
        for (int piece = 0; piece < 3; ++piece) {
            unsigned t0 = fbits(a[0]) & 0xffff0000u, t1 = fbits(a[1]) & 0xffff0000u;
            want[piece] = t1 | (t0 >> 16);
            a[0] = a[0] - bitsf(t0); a[1] = a[1] - bitsf(t1);
        }
        for (int piece = 0; piece < 3; ++piece)
            if (ho[3 * (i / 2) + piece] != want[piece]) {
                ++bad;
                const unsigned e0 = (fbits(hv[i]) >> 23) & 0xff, e1 = (fbits(hv[i + 1]) >> 23) & 0xff;
                if (e0 > 40 && e1 > 40) { if (bad_normal < 5) printf("  mismatch piece %d: v = %08x %08x got %08x want %08x\n", piece, fbits(hv[i]), fbits(hv[i + 1]), ho[3 * (i / 2) + piece], want[piece]); ++bad_normal; }
            }
    }
    printf("packed pieces of %d value pairs, dot2c form against and + sub (host): %ld words differ, %ld of them with both exponents above 2^-87\n", n / 2, bad, bad_normal);
    printf("cycles per MFMA, split of one / two value pairs per gap:\n");
    printf("  and + sub form (11 instructions per pair)            %6.1f %6.1f\n", run1<0, 1>(), run1<0, 2>());
    printf("  dot2c form with copies (9 instructions per pair)     %6.1f %6.1f\n", run1<1, 1>(), run1<1, 2>());
    printf("  dot2c form in place (7 instructions per pair)        %6.1f %6.1f\n", run1<2, 1>(), run1<2, 2>());
    return 0;
}
