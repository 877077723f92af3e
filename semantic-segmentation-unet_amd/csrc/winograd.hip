// Winograd F(2x2, 3x3) for the 3x3 'same' convolutions with many channels (reference layer: UNet/model.py:28-35).
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A        per 2x2 output tile, 4x4 input patch d, 3x3 kernel g
//
// Exact fp32 arithmetic, 16 multiplies per tile and channel pair instead of 36 (2.25x fewer MACs on the matrix cores).
// Unfused pipeline: input transform (HBM-bound, writes 16 planes [tile][Cin] = 4x the input bytes) -> 16 independent
// fp32-MFMA GEMMs [T x Cin] x [Cin x Cout] (conv_igemm.hip, MODE 3) -> output transform (+bias, ReLU).  The transforms
// move ~14x the activation bytes, so the engine uses this path only where channels are wide enough (>= 256) for the
// saved matrix work to dominate.  dgrad is the same pipeline on dz with the 180-degree rotated, in/out-swapped kernel.
//
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]   A^T = [1 1 1 0; 0 1 -1 -1]
#include <utility>
#include "common.h"
#include "wino_epilogue.h"
#include <cstdlib>
#ifndef UNET_ABLATE
#define UNET_ABLATE 0
#endif

namespace {

// U[xi][k][n], xi = 4*i + j.  mode 0 (forward): k = ci, n = co, g[a][b] = w[a][b][ci][co];
// mode 1 (dgrad): k = co, n = ci, g[a][b] = w[2-a][2-b][ci][co].
__global__ __launch_bounds__(256) void wino_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int Ci, int Co, int mode) {
    const long total = (long)Ci * Co, stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const int ci = (int)(i / Co), co = (int)(i % Co);
        float g[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int aa = (mode & 1) ? 2 - a : a, bb = (mode & 1) ? 2 - b : b;
                g[a][b] = w[((size_t)(aa * 3 + bb) * Ci + ci) * Co + co];
            }
        float s[4][3];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            s[0][b] = g[0][b];
            s[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
            s[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
            s[3][b] = g[2][b];
        }
        const size_t plane = (size_t)Ci * Co;
        size_t off;
        if (mode == 0)      off = (size_t)ci * Co + co;                                     // [k=ci][n=co]
        else if (mode == 1) off = (size_t)co * Ci + ci;                                     // [k=co][n=ci]
        else if (mode == 2) off = ((size_t)(ci >> 3) * Co + co) * 8 + (ci & 7);             // [k/8][n=co][k%8]  (fused kernel)
        else                off = ((size_t)(co >> 3) * Ci + ci) * 8 + (co & 7);             // [k/8][n=ci][k%8]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            U[(size_t)(4 * r + 0) * plane + off] = s[r][0];
            U[(size_t)(4 * r + 1) * plane + off] = 0.5f * (s[r][0] + s[r][1] + s[r][2]);
            U[(size_t)(4 * r + 2) * plane + off] = 0.5f * (s[r][0] - s[r][1] + s[r][2]);
            U[(size_t)(4 * r + 3) * plane + off] = s[r][2];
        }
    }
}

// All fused-route weight transforms of a step in ONE launch (34 tiny launches otherwise: ~0.5 ms of launch bubbles per
// step).  jobs[j] = { w, Uc forward (mode 2), Uc dgrad (mode 3), Ci | Co << 32, first block, unused }.  The data-gradient
// kernel uses the 180-degree rotated filter, whose transform is the forward one with points 0 and 3 swapped in both
// directions (G flip(g) G^T = P (G g G^T) P, P = (3,1,2,0)): computed once, written twice.
__global__ __launch_bounds__(256) void wino_weight_batch_kernel(const long long* __restrict__ jobs, int njobs) {
    int j = 0;
    while (j + 1 < njobs && (int)jobs[(j + 1) * 6 + 4] <= (int)blockIdx.x) ++j;
    const float* __restrict__ w = reinterpret_cast<const float*>(jobs[j * 6 + 0]);
    float* __restrict__ uf = reinterpret_cast<float*>(jobs[j * 6 + 1]);
    float* __restrict__ ud = reinterpret_cast<float*>(jobs[j * 6 + 2]);
    const int Ci = (int)(jobs[j * 6 + 3] & 0xffffffffll), Co = (int)(jobs[j * 6 + 3] >> 32);
    const size_t plane = (size_t)Ci * Co;
    // one thread = 8 consecutive input channels x one output channel: the forward layout [ci/8][co][8] gets 32-byte runs
    // (adjacent threads = adjacent co = adjacent runs), the data-gradient layout [co/8][ci][8] gets 8 lanes x 4 bytes
    const long items = (long)(Ci >> 3) * Co;
    const long it = ((long)blockIdx.x - (int)jobs[j * 6 + 4]) * 256 + threadIdx.x;
    if (it >= items) return;
    const int c8 = (int)(it / Co), co = (int)(it % Co);
    float t[16][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ci = 8 * c8 + e;
        float g[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) g[a][b] = w[((size_t)(a * 3 + b) * Ci + ci) * Co + co];
        float s[4][3];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            s[0][b] = g[0][b];
            s[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
            s[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
            s[3][b] = g[2][b];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            t[4 * r + 0][e] = s[r][0]; t[4 * r + 1][e] = 0.5f * (s[r][0] + s[r][1] + s[r][2]);
            t[4 * r + 2][e] = 0.5f * (s[r][0] - s[r][1] + s[r][2]); t[4 * r + 3][e] = s[r][2];
        }
    }
    const size_t offf = ((size_t)c8 * Co + co) * 8;                                   // [k/8][n=co][k%8]
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) {
        float* o = uf + (size_t)xi * plane + offf;
        *reinterpret_cast<f32x4*>(o) = f32x4{t[xi][0], t[xi][1], t[xi][2], t[xi][3]};
        *reinterpret_cast<f32x4*>(o + 4) = f32x4{t[xi][4], t[xi][5], t[xi][6], t[xi][7]};
        const int r = xi >> 2, c = xi & 3;
        const int xp = 4 * (r == 0 ? 3 : (r == 3 ? 0 : r)) + (c == 0 ? 3 : (c == 3 ? 0 : c));
        float* od = ud + (size_t)xp * plane + ((size_t)(co >> 3) * Ci + 8 * c8) * 8 + (co & 7);     // [k/8][n=ci][k%8]
#pragma unroll
        for (int e = 0; e < 8; ++e) od[e * 8] = t[xi][e];
    }
}


// ---- fully fused Winograd F(2x2,3x3) forward / data-gradient convolution ------------------------------------------------
// One workgroup = 8x8 Winograd tiles (16x16 output pixels) x 64 output channels; wave (mi, ni) owns [32 tiles x 32 channels]
// for ALL 16 Winograd points (16 x f32x16 = 256 AGPRs, one wave per SIMD).  K is streamed in chunks of 8 input channels:
//   D  raw input patch 18x18 px x 8 ch   <- LDS-DMA from x (zero page outside the image)                 2 buffers x 12 KB
//   V  [xi][tile][8] = B^T d B           <- every lane transforms one (tile, channel pair) per chunk      2 buffers x 32 KB
//   U  [xi][n][8] transformed weights    <- LDS-DMA from Uc[xi][K/8][N][8] (L2-resident)                  2 buffers x 32 KB
// A lane's 4 consecutive k (lane half picks the quad) are one ds_read_b128 for both operands; the quad slot is XOR-ed with
// bit 3 of the row (on the DMA source side for U, on the transform's write side for V) so the 16-lane read groups hit 16
// distinct 16-B slots.  The output transform A^T m A is lane-local in the epilogue.  HBM sees the input once (x1.27 halo)
// and the output once; the 4x-expanded V / M planes of the unfused route never exist.
//
// Schedule.  With one wave per SIMD nothing but asynchronous work hides under the MFMAs: measured on gfx950
// (scripts/micro/mfma_issue_cost.hip) a VALU instruction between two MFMAs of the same wave costs its own ~4 cycles plus
// ~10 per matrix-pipe <-> VALU switch, while LDS reads and DMA issue cost ~1 and ~15.  So all four waves share the transform
// (32 packed-fp32 VALU instructions per lane per chunk, one block) and the DMA issue (11 per wave per chunk, addresses from
// scalar bases / running pointers), and the chunk loop is written as an explicit instruction stream (inline asm; the
// compiler would re-sink the LDS reads next to their uses and wait on each):
//   barrier | DMA U(c+1), D(c+3) | operand reads for points 0,1 | transform D(c+1) regs -> V(c+1) | 16 x { wait operands(xi);
//   4 MFMAs; operand reads for xi+2; one LDS read of D(c+2) into registers } | wait all | barrier
// D is pipelined DMA -> LDS -> registers -> V, one chunk per stage, so everything a chunk needs was issued a full chunk
// earlier.  Operand registers form a 3-deep ring.  lgkmcnt immediates below count the LDS instructions issued after the one
// waited for (LDS instructions retire in order).
#define WF_RD128(dst, base, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst) : "v"(base), "n"(off))
#define WF_RD64(dst, base, off) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=&v"(dst) : "v"(base), "n"(off))
#define WF_WR64(base, off, val) asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(base), "v"(val), "n"(off) : "memory")
#define WF_MFMA(accv, av, bv) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(accv) : "v"(av), "v"(bv) : "memory")
#define WF_ALL_D \
    "+v"(dd[0][0]), "+v"(dd[0][1]), "+v"(dd[0][2]), "+v"(dd[0][3]), "+v"(dd[1][0]), "+v"(dd[1][1]), "+v"(dd[1][2]), "+v"(dd[1][3]), \
    "+v"(dd[2][0]), "+v"(dd[2][1]), "+v"(dd[2][2]), "+v"(dd[2][3]), "+v"(dd[3][0]), "+v"(dd[3][1]), "+v"(dd[3][2]), "+v"(dd[3][3])

#if UNET_ABLATE == 8
__device__ long long g_wf_timeline[8];
#endif
constexpr int kWfDB = 12 * 256 * 4, kWfIB = 16 * 64 * 8 * 4;      // bytes of one D buffer (18x18 px x 8 ch in 12 1-KB pieces) / one V or U image

// (template functions rather than generic lambdas: clang rejects captured variables as asm operands inside generic lambdas)
// raw patch D buffer PAR -> registers: 16 LDS reads, not waited for
template <int PAR> __device__ __forceinline__ void wf_read_D(f32x2 (&dd)[4][4], unsigned d_base) {
#pragma unroll
    for (int j = 0; j < 16; ++j) WF_RD64(dd[j >> 2][j & 3], d_base, PAR * kWfDB + ((j >> 2) * 18 + (j & 3)) * 32);
}

// B^T d B of the raw patch in dd -> V image PAR (16 LDS writes).  In the chunk loop 4 operand reads are issued ahead of the
// writes: with 8 writes behind them lgkmcnt(8) means the reads are back (the counter is 4 bits, so 16 writes cannot be
// counted past in one go).
template <int PAR> __device__ __forceinline__ void wf_transform_to(const f32x2 (&dd)[4][4], unsigned v_base) {
    f32x2 tt[4][4], vv[16];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        tt[0][c] = dd[0][c] - dd[2][c]; tt[1][c] = dd[1][c] + dd[2][c];
        tt[2][c] = dd[2][c] - dd[1][c]; tt[3][c] = dd[1][c] - dd[3][c];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        vv[4 * r + 0] = tt[r][0] - tt[r][2]; vv[4 * r + 1] = tt[r][1] + tt[r][2];
        vv[4 * r + 2] = tt[r][2] - tt[r][1]; vv[4 * r + 3] = tt[r][1] - tt[r][3];
    }
#pragma unroll
    for (int xi = 0; xi < 8; ++xi) WF_WR64(v_base, PAR * kWfIB + xi * 2048, vv[xi]);
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
#pragma unroll
    for (int xi = 8; xi < 16; ++xi) WF_WR64(v_base, PAR * kWfIB + xi * 2048, vv[xi]);
}

// One chunk after its DMAs were issued; PAR = chunk parity = buffer of V(c), U(c), D(c+2); the other buffers take V(c+1).
#if UNET_ABLATE == 8
#define WF_STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory")
#define WF_TL_ARG , long long (&tl)[6]
#else
#define WF_STAMP(t)
#define WF_TL_ARG
#endif
// One chunk; PAR = chunk parity = buffer of V(c), U(c), D(c+2); the other buffers take V(c+1), U(c+1), D(c+3).  usrc / dsrc
// are this lane's DMA sources for U(c+1) (8 pieces) and D(c+3) (3 pieces); sUw / sDw the wave's first piece of buffer 0.
// A global_load_lds occupies the vector-memory issue path for ~64 cycles, during which the wave can run MFMAs but not issue
// another one: eleven in a row stall ~900 cycles, one per MFMA group costs ~15 each (measured, scripts/micro/mfma_issue_cost).
// FIRST: the chunk opens a new output tile - the first MFMA of every point starts from C = 0 instead of the accumulator.
template <int PAR, bool FIRST = false, class SRC> __device__ __forceinline__ void wf_chunk(f32x16 (&acc)[16], f32x2 (&dd)[4][4], unsigned a_base, unsigned b_base,
                                                            unsigned d_base, unsigned v_base, SRC&& sources, const float* (&usrc)[8],
                                                            const float* (&dsrc)[3], float* sUw, float* sDw WF_TL_ARG) {
    constexpr int IMG = 16 * 64 * 8, DFL = 12 * 256;
    f32x4 A[3], B[3];
#if UNET_ABLATE == 8
    long long s0, s1, s2, s3;
    WF_STAMP(s0);
    tl[0] += s0 - tl[5];          // address arithmetic (since the previous barrier)
#endif
    WF_RD128(A[0], a_base, PAR * kWfIB); WF_RD128(B[0], b_base, PAR * kWfIB);
    WF_RD128(A[1], a_base, PAR * kWfIB + 2048); WF_RD128(B[1], b_base, PAR * kWfIB + 2048);
    sources();                                     // DMA source addresses (VALU) while those four reads are in flight
    wf_transform_to<PAR ^ 1>(dd, v_base);
    asm volatile("" : "+v"(A[0]), "+v"(B[0]), "+v"(A[1]), "+v"(B[1]));
    WF_STAMP(s1);
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) {
        if (xi >= 2) {
            // issued after the reads for xi: D read of group xi-2, then group xi-1's operand reads (if any) and D read
            if (xi + 1 < 16) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(A[xi % 3]), "+v"(B[xi % 3]));
            else asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(A[xi % 3]), "+v"(B[xi % 3]));
        }
        // srcA = weights (rows of the result = output channels), srcB = data (columns = tiles): a lane then holds 16 channels of
        // ONE tile, in runs of 4 consecutive channels - the epilogue stores dwordx4 and does its arithmetic packed
        if (FIRST) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=a"(acc[xi]) : "v"(B[xi % 3][0]), "v"(A[xi % 3][0]) : "memory");
        else WF_MFMA(acc[xi], B[xi % 3][0], A[xi % 3][0]);
        // DMA number j of this chunk: U pieces 0..7, then D pieces 0..2; two per group (behind the 1st and 3rd MFMA), so all
        // eleven are on their way after 6 of the 16 groups and have the rest of the chunk to land
        if (2 * xi < 8) __builtin_amdgcn_global_load_lds(usrc[2 * xi], (lds_void_f*)(sUw + (PAR ^ 1) * IMG + 4 * (2 * xi) * 256), 16, 0, 0);
        else if (2 * xi < 11) __builtin_amdgcn_global_load_lds(dsrc[2 * xi - 8], (lds_void_f*)(sDw + (PAR ^ 1) * DFL + 4 * (2 * xi - 8) * 256), 16, 0, 0);
        WF_MFMA(acc[xi], B[xi % 3][1], A[xi % 3][1]);
        WF_MFMA(acc[xi], B[xi % 3][2], A[xi % 3][2]);
        if (2 * xi + 1 < 8) __builtin_amdgcn_global_load_lds(usrc[2 * xi + 1], (lds_void_f*)(sUw + (PAR ^ 1) * IMG + 4 * (2 * xi + 1) * 256), 16, 0, 0);
        else if (2 * xi + 1 < 11) __builtin_amdgcn_global_load_lds(dsrc[2 * xi + 1 - 8], (lds_void_f*)(sDw + (PAR ^ 1) * DFL + 4 * (2 * xi + 1 - 8) * 256), 16, 0, 0);
        WF_MFMA(acc[xi], B[xi % 3][3], A[xi % 3][3]);
        if (xi + 2 < 16) {
            WF_RD128(A[(xi + 2) % 3], a_base, PAR * kWfIB + (xi + 2) * 2048);
            WF_RD128(B[(xi + 2) % 3], b_base, PAR * kWfIB + (xi + 2) * 2048);
        }
        WF_RD64(dd[xi >> 2][xi & 3], d_base, PAR * kWfDB + ((xi >> 2) * 18 + (xi & 3)) * 32);      // D(c+2) -> registers
    }
    WF_STAMP(s2);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" : WF_ALL_D : : "memory");
#if UNET_ABLATE == 8
    WF_STAMP(s3);
    tl[1] += s1 - s0; tl[2] += s2 - s1; tl[3] += s3 - s2; tl[5] = s3;
#endif
}

__global__ __launch_bounds__(256, 1) void wino_fused_kernel(WinoFusedArgs p) {
    constexpr int DPIX = 18 * 18, DPIECES = 12, DFL = DPIECES * 256;       // D image padded to 3 1-KB DMA pieces per wave
    constexpr int IMG = 16 * 64 * 8;
    constexpr int DB = kWfDB, IB = kWfIB;
    static_assert(DB == DFL * 4 && IB == IMG * 4, "buffer sizes");
    __shared__ __attribute__((aligned(1024))) float smem[2 * DFL + 4 * IMG];
    float* sD = smem; float* sU = smem + 2 * DFL + 2 * IMG;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mi = wv & 1, ni = wv >> 1;
    const int li = lane & 31, lh = lane >> 5;
    int b = blockIdx.x;
    const int tn = b % p.nt; b /= p.nt;
    const int bx = b % p.tbx; b /= p.tbx;
    const int by = b % p.tby; const int img = b / p.tby;
    const int n0 = tn * 64;
    const int gy0 = 16 * by - 1, gx0 = 16 * bx - 1;           // image coords of patch pixel (0,0)
    const int nchunks = p.K / 8;

    f32x16 acc[16];
#pragma unroll
    for (int x = 0; x < 16; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;

    // --- DMA duty of this wave: D pieces wv, wv+4, wv+8 (32 pixels each, lane -> pixel lane/2, channel quad lane&1) and
    //     U pieces wv + 4k, k = 0..7 (point xi = 2k + wv/2, rows 32*(wv&1) + lane/2; quad slot swizzled by bit 3 of the row)
    const int drow = lane >> 1, dh = lane & 1;
    const float* dptr[3];                 // chunk 0 source of the lane's D pixels; advanced 8 channels per chunk
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int pix = 32 * (wv + 4 * k) + drow;
        const int py = pix / 18, px = pix - py * 18;
        const int gy = gy0 + py, gx = gx0 + px;
        const bool ok = pix < DPIX && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        dptr[k] = ok ? p.x + ((size_t)(img * p.H + gy) * p.W + gx) * p.ldx + 4 * dh : (p.pad ? p.pad : g_zero_page_f) + 4 * dh;
    }
    const int urow = 32 * (wv & 1) + drow;
    const unsigned uoff = (unsigned)((urow * 8 + 4 * (dh ^ ((urow >> 3) & 1))) * 4);        // bytes, same for all 8 pieces
    const size_t ustride_xi = (size_t)nchunks * p.Nout * 32;                                // bytes between points
    const char* ubase0 = reinterpret_cast<const char*>(p.Uc) + (size_t)n0 * 32 + (size_t)(wv >> 1) * ustride_xi;
    auto issue_D = [&](int chunk, int par) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
            __builtin_amdgcn_global_load_lds(dptr[k] + chunk * 8, (lds_void_f*)(sD + par * DFL + (wv + 4 * k) * 256), 16, 0, 0);
    };
    auto issue_U = [&](int chunk, int par) {
        const char* ub = ubase0 + (size_t)chunk * p.Nout * 32;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(ub + 2 * k * ustride_xi + uoff),
                                             (lds_void_f*)(sU + par * IMG + (wv + 4 * k) * 256), 16, 0, 0);
    };

    // --- LDS byte addresses (everything else is an immediate offset)
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_f*)smem;
    const unsigned ldsV = lds0 + 2 * DB, ldsU = ldsV + 2 * IB;
    const int arow = 32 * mi + li, brow = 32 * ni + li;
    const unsigned a_base = ldsV + 4u * (arow * 8 + 4 * (lh ^ ((arow >> 3) & 1)));
    const unsigned b_base = ldsU + 4u * (brow * 8 + 4 * (lh ^ ((brow >> 3) & 1)));
    // transform duty: lane -> (tile t_lt = 16*wv + lane/4, channel pair t_q = lane&3); V slot = quad ^ bit3(tile)
    const int t_lt = 16 * wv + (lane >> 2), t_q = lane & 3;
    const unsigned d_base = lds0 + 4u * (((2 * (t_lt >> 3)) * 18 + 2 * (t_lt & 7)) * 8 + 2 * t_q);
    const unsigned v_base = ldsV + 4u * (t_lt * 8 + 4 * ((t_q >> 1) ^ ((t_lt >> 3) & 1)) + 2 * (t_q & 1));

    f32x2 dd[4][4];
    f32x4 bias4[4];
    wf_load_bias(p, n0, ni, lh, bias4);
    const int last = nchunks - 1;
#if UNET_ABLATE == 8        /* diagnostics only: s_memtime at the phase boundaries of workgroup 0 */
    long long wf_t0, wf_t1, wf_t2;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(wf_t0) :: "memory");
#endif

    // prologue: D(0), D(1), U(0) -> LDS; V(0) from D(0); D(1) into registers; D(2) -> LDS
    issue_D(0, 0); issue_D(min(1, last), 1); issue_U(0, 0);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    wf_read_D<0>(dd, d_base);
    asm volatile("s_waitcnt lgkmcnt(0)" : WF_ALL_D);
    wf_transform_to<0>(dd, v_base);
    wf_read_D<1>(dd, d_base);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : WF_ALL_D : : "memory");
    issue_D(min(2, last), 0);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

#if UNET_ABLATE == 8
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(wf_t1) :: "memory");
    long long tl[6] = {0, 0, 0, 0, 0, wf_t1};
#define WF_TL , tl
#else
#define WF_TL
#endif
    // (prefetches past the last chunk re-read it; nothing consumes them)
    const float* usrc[8]; const float* dsrc[3];
    float* const sUw = sU + wv * 256; float* const sDw = sD + wv * 256;
    auto sources = [&](int cu, int cd) {          // this lane's DMA sources for U(cu) and D(cd): 11 64-bit adds, pinned here
        const char* ub = ubase0 + (size_t)cu * p.Nout * 32;
#pragma unroll
        for (int k = 0; k < 8; ++k) usrc[k] = reinterpret_cast<const float*>(ub + 2 * k * ustride_xi + uoff);
#pragma unroll
        for (int k = 0; k < 3; ++k) dsrc[k] = dptr[k] + cd * 8;
        asm volatile("" : "+v"(usrc[0]), "+v"(usrc[1]), "+v"(usrc[2]), "+v"(usrc[3]), "+v"(usrc[4]), "+v"(usrc[5]), "+v"(usrc[6]),
                          "+v"(usrc[7]), "+v"(dsrc[0]), "+v"(dsrc[1]), "+v"(dsrc[2]));
    };
    for (int c = 0; c + 1 < nchunks; c += 2) {
        wf_chunk<0>(acc, dd, a_base, b_base, d_base, v_base, [&] { sources(min(c + 1, last), min(c + 3, last)); }, usrc, dsrc, sUw, sDw WF_TL);
        wf_chunk<1>(acc, dd, a_base, b_base, d_base, v_base, [&] { sources(min(c + 2, last), min(c + 4, last)); }, usrc, dsrc, sUw, sDw WF_TL);
    }
    if (nchunks & 1) {                // odd tail outside the loop: inside it the extra control flow made the allocator spill
        wf_chunk<0>(acc, dd, a_base, b_base, d_base, v_base, [&] { sources(last, last); }, usrc, dsrc, sUw, sDw WF_TL);
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the inline-asm MFMAs are invisible to the compiler's hazard recogniser
#if UNET_ABLATE == 8
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(wf_t2) :: "memory");
#endif

    f32x2 s1u[8], s2u[8]; f32x4 rvu[4][4];
    wf_epilogue<0>(acc, p, img, by, bx, n0, mi, ni, li, lh, bias4, s1u, s2u, rvu);
#if UNET_ABLATE == 8
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (blockIdx.x == 0 && tid == 0) {
        long long t3; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t3) :: "memory");
        g_wf_timeline[0] = wf_t1 - wf_t0; g_wf_timeline[1] = wf_t2 - wf_t1; g_wf_timeline[2] = t3 - wf_t2; g_wf_timeline[3] = nchunks;
        for (int i = 0; i < 4; ++i) g_wf_timeline[4 + i] = tl[i];
    }
#endif
}

// Persistent form of the kernel above for K % 16 == 0, K >= 32: a workgroup walks output tiles blockIdx.x, +gridDim.x, ... and
// the chunk stream runs straight across tile boundaries - the last chunks of a tile prefetch U(0), D(0..2) of the next one
// instead of idling, so only the first tile of a workgroup pays the two DMA round trips of the prologue, and there is no
// launch gap between tiles (with 152 KB of LDS a CU holds one workgroup, so nothing else would hide either).  A tile's first
// chunk starts its accumulators from C = 0.
template <int STATS>
__device__ __forceinline__ void wino_fused_stream_body(const WinoFusedArgs& p, int ntiles) {
    constexpr int DPIX = 18 * 18, DPIECES = 12, DFL = DPIECES * 256;
    constexpr int IMG = 16 * 64 * 8;
    constexpr int DB = kWfDB, IB = kWfIB;
    __shared__ __attribute__((aligned(1024))) float smem[2 * DFL + 4 * IMG];
    float* sD = smem; float* sU = smem + 2 * DFL + 2 * IMG;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mi = wv & 1, ni = wv >> 1;
    const int li = lane & 31, lh = lane >> 5;
    const int nchunks = p.K / 8;
    const int drow = lane >> 1, dh = lane & 1;
    const int urow = 32 * (wv & 1) + drow;
    const unsigned uoff = (unsigned)((urow * 8 + 4 * (dh ^ ((urow >> 3) & 1))) * 4);
    const size_t ustride_xi = (size_t)nchunks * p.Nout * 32;
    const size_t ustep = (size_t)p.Nout * 32;                                               // bytes between chunks

    // Per-tile DMA sources: this lane's three D pixels at channel 0 and the wave's U base at chunk 0.  The lane's patch
    // pixels are tile-invariant; a tile contributes a scalar base and the image-border test.  Tile coordinates advance by
    // gridDim.x tiles with carries instead of being decoded by division each time (per-tile VALU work is matrix-pipe time).
    int ppy[3], ppx[3], poff[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int pix = 32 * (wv + 4 * k) + drow;
        ppy[k] = pix < DPIX ? pix / 18 : (1 << 20);                 // beyond the 18x18 patch: never in the image
        ppx[k] = pix % 18;
        poff[k] = ((pix / 18) * p.W + ppx[k]) * p.ldx + 4 * dh;
    }
    struct TileCoord { int tn, bx, by, img; };
    auto decode = [&](int t) { TileCoord c; c.tn = t % p.nt; t /= p.nt; c.bx = t % p.tbx; t /= p.tbx; c.by = t % p.tby; c.img = t / p.tby; return c; };
    const TileCoord dstep = decode((int)gridDim.x);
    auto advance = [&](TileCoord c) {
        c.tn += dstep.tn; int cy = c.tn >= p.nt; c.tn -= cy ? p.nt : 0;
        c.bx += dstep.bx + cy; cy = c.bx >= p.tbx; c.bx -= cy ? p.tbx : 0;
        c.by += dstep.by + cy; cy = c.by >= p.tby; c.by -= cy ? p.tby : 0;
        c.img += dstep.img + cy;
        return c;
    };
    const float* const padsrc = p.pad ? p.pad : g_zero_page_f;
    auto tile_sources = [&](const TileCoord& c, const float* (&dp)[3], const char*& ub0) {
        const int gy0 = 16 * c.by - 1, gx0 = 16 * c.bx - 1;
        const float* xb = p.x + ((long long)(c.img * p.H + gy0) * p.W + gx0) * p.ldx;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const bool ok = (unsigned)(gy0 + ppy[k]) < (unsigned)p.H && (unsigned)(gx0 + ppx[k]) < (unsigned)p.W;
            dp[k] = ok ? xb + poff[k] : padsrc + 4 * dh;
        }
        ub0 = reinterpret_cast<const char*>(p.Uc) + (size_t)c.tn * 64 * 32 + (size_t)(wv >> 1) * ustride_xi;
    };

    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_f*)smem;
    const unsigned ldsV = lds0 + 2 * DB, ldsU = ldsV + 2 * IB;
    const int arow = 32 * mi + li, brow = 32 * ni + li;
    const unsigned a_base = ldsV + 4u * (arow * 8 + 4 * (lh ^ ((arow >> 3) & 1)));
    const unsigned b_base = ldsU + 4u * (brow * 8 + 4 * (lh ^ ((brow >> 3) & 1)));
    const int t_lt = 16 * wv + (lane >> 2), t_q = lane & 3;
    const unsigned d_base = lds0 + 4u * (((2 * (t_lt >> 3)) * 18 + 2 * (t_lt & 7)) * 8 + 2 * t_q);
    const unsigned v_base = ldsV + 4u * (t_lt * 8 + 4 * ((t_q >> 1) ^ ((t_lt >> 3) & 1)) + 2 * (t_q & 1));
    float* const sUw = sU + wv * 256; float* const sDw = sD + wv * 256;

    f32x16 acc[16];
    f32x2 dd[4][4];
    const float* dcur[3]; const float* dnxt[3]; const char* ucur; const char* unxt;
    // Workgroups are dealt round-robin to the 8 XCDs (one L2 each).  With fewer than 8 n-tiles per spatial block, renumber
    // so that an XCD's workgroups hold CONSECUTIVE tile blocks: the n-tiles of one spatial block (same input patch) and
    // neighbouring blocks (shared halo) then meet in one L2 instead of crossing the fabric once per XCD (measured: -35 % HBM
    // reads on the 64..256-channel layers).  With 8 or 16 n-tiles the round-robin deal already pins one weight slice
    // (2-4 MB, the dominant stream there) to each XCD for the whole launch, and renumbering would triple the reads.
    int t = blockIdx.x;
#if UNET_ABLATE != 9
    if ((gridDim.x & 7) == 0 && (p.nt & 7) != 0) t = (t & 7) * (int)(gridDim.x >> 3) + (t >> 3);
#endif
    const int t_first = t;
    f32x2 s1[8], s2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { s1[i] = f32x2{0.f, 0.f}; s2[i] = f32x2{0.f, 0.f}; }
    TileCoord tc = decode(t);
    tile_sources(tc, dcur, ucur);

    // prologue of the workgroup's first tile: D(0), D(1), U(0) -> LDS; V(0) from D(0); D(1) into registers; D(2) -> LDS
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        __builtin_amdgcn_global_load_lds(dcur[k], (lds_void_f*)(sDw + 4 * k * 256), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(dcur[k] + 8, (lds_void_f*)(sDw + DFL + 4 * k * 256), 16, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(ucur + 2 * k * ustride_xi + uoff), (lds_void_f*)(sUw + 4 * k * 256), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    wf_read_D<0>(dd, d_base);
    asm volatile("s_waitcnt lgkmcnt(0)" : WF_ALL_D);
    wf_transform_to<0>(dd, v_base);
    wf_read_D<1>(dd, d_base);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : WF_ALL_D : : "memory");
#pragma unroll
    for (int k = 0; k < 3; ++k) __builtin_amdgcn_global_load_lds(dcur[k] + 16, (lds_void_f*)(sDw + 4 * k * 256), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

    const float* usrc[8]; const float* dsrc[3];
    // this lane's DMA sources for chunk c of the current tile: U(c+1) and D(c+3), continuing into the next tile
    auto sources = [&](int c) {
        const int cu = c + 1, cd = c + 3;
        const char* ub = cu < nchunks ? ucur + (size_t)cu * ustep : unxt;
#pragma unroll
        for (int k = 0; k < 8; ++k) usrc[k] = reinterpret_cast<const float*>(ub + 2 * k * ustride_xi + uoff);
        const bool same = cd < nchunks;
        const int doff = (same ? cd : cd - nchunks) * 8;
#pragma unroll
        for (int k = 0; k < 3; ++k) dsrc[k] = (same ? dcur[k] : dnxt[k]) + doff;
        asm volatile("" : "+v"(usrc[0]), "+v"(usrc[1]), "+v"(usrc[2]), "+v"(usrc[3]), "+v"(usrc[4]), "+v"(usrc[5]), "+v"(usrc[6]),
                          "+v"(usrc[7]), "+v"(dsrc[0]), "+v"(dsrc[1]), "+v"(dsrc[2]));
    };

#if UNET_ABLATE == 8
    long long q0, q1, q2, q3, qa[4] = {0, 0, 0, 0};
    long long tl[6] = {0, 0, 0, 0, 0, 0};
    WF_STAMP(q3);
#endif
    for (; t < ntiles; t += gridDim.x) {
#if UNET_ABLATE == 8
        WF_STAMP(q0); tl[5] = q0;
#endif
        const TileCoord tcn = t + (int)gridDim.x < ntiles ? advance(tc) : tc;            // the last tile prefetches itself again
        tile_sources(tcn, dnxt, unxt);
        f32x4 bias4[4];
        wf_load_bias(p, tc.tn * 64, ni, lh, bias4);
        f32x4 rv[4][4];
        if (STATS == 2) wf_load_r(p, tc.img, tc.by, tc.bx, tc.tn * 64, mi, ni, li, lh, rv);
        wf_chunk<0, true>(acc, dd, a_base, b_base, d_base, v_base, [&] { sources(0); }, usrc, dsrc, sUw, sDw WF_TL);
        wf_chunk<1>(acc, dd, a_base, b_base, d_base, v_base, [&] { sources(1); }, usrc, dsrc, sUw, sDw WF_TL);
        for (int c = 2; c < nchunks; c += 2) {
            wf_chunk<0>(acc, dd, a_base, b_base, d_base, v_base, [&] { sources(c); }, usrc, dsrc, sUw, sDw WF_TL);
            wf_chunk<1>(acc, dd, a_base, b_base, d_base, v_base, [&] { sources(c + 1); }, usrc, dsrc, sUw, sDw WF_TL);
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // inline-asm MFMAs are invisible to the compiler's hazard recogniser
#if UNET_ABLATE == 8
        WF_STAMP(q1);
#endif

        wf_epilogue<STATS>(acc, p, tc.img, tc.by, tc.bx, tc.tn * 64, mi, ni, li, lh, bias4, s1, s2, rv);
#pragma unroll
        for (int k = 0; k < 3; ++k) dcur[k] = dnxt[k];
        ucur = unxt; tc = tcn;
#if UNET_ABLATE == 8
        WF_STAMP(q2);
        qa[0] += q0 - q3; qa[1] += q1 - q0; qa[2] += q2 - q1; qa[3] += 1; q3 = q2;
#endif
    }
#if UNET_ABLATE == 8
    if (blockIdx.x == 0 && tid == 0) {
        g_wf_timeline[0] = qa[0]; g_wf_timeline[1] = qa[1]; g_wf_timeline[2] = qa[2]; g_wf_timeline[3] = nchunks * qa[3];
        for (int i = 0; i < 4; ++i) g_wf_timeline[4 + i] = tl[i];
    }
#endif
    if (STATS) wf_write_stats(p, t_first, 2 * ((int)gridDim.x / p.nt), mi, ni, li, lh, s1, s2);
}
// (plain kernels around the templated body: the host-side stub of a kernel TEMPLATE containing this inline asm is not emitted)
__global__ __launch_bounds__(256, 1) void wino_fused_stream_kernel(WinoFusedArgs p, int ntiles) { wino_fused_stream_body<0>(p, ntiles); }
__global__ __launch_bounds__(256, 1) void wino_fused_stream_stats_kernel(WinoFusedArgs p, int ntiles) { wino_fused_stream_body<1>(p, ntiles); }
__global__ __launch_bounds__(256, 1) void wino_fused_stream_bnbwd_kernel(WinoFusedArgs p, int ntiles) { wino_fused_stream_body<2>(p, ntiles); }

// ---- fully fused Winograd weight gradient ----------------------------------------------------------------------------
//   dW = G^T [ sum_tiles (B^T d B)[xi][ci] * (A dY A^T)[xi][co] ] G
// Workgroup = 64 input channels x 64 output channels x all 16 Winograd points (wave = [32 ci x 32 co] x 16 points in 256
// AGPRs); the MFMA reduce dimension is TILES: a chunk = 8 consecutive tiles of one tile row.  Raw rows only go through
// LDS -- x: 4 pixel rows x 18 px x 64 ci, dz: 2 rows x 16 px x 64 co, LDS-DMA'd two chunks ahead into a 3-deep ring --
// and every lane transforms ITS OWN 4 tiles in registers (lane = channel, lane half = tile quad): no transformed LDS
// images, no transform waves, one barrier per chunk.  Tile s of the chunk feeds 16 independent MFMAs (one per point).
// Epilogue: G^T dU G is lane-local; split partials go to the workspace [split][9][Ci][Co] and are reduced in fixed order.
template <int... I, class F> __device__ __forceinline__ void static_for_seq(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
// f(step) for step = 0 .. N-1 with decltype(step)::value a compile-time constant (an unrolled loop whose index can pick registers,
// immediates and `if constexpr` branches without depending on the unroller)
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) { static_for_seq(std::make_integer_sequence<int, N>{}, f); }
struct WinoWgradArgs {
    const float* x; const float* dz; float* ws;
    int ldx, lddz, N, H, W, Ci, Co;
    int mt, nt, splits, tbx, nchunks;
};
typedef __attribute__((address_space(3))) void lds_void_g;

__global__ __launch_bounds__(256, 1) void wino_wgrad_fused_kernel(WinoWgradArgs p) {
    constexpr int XP = 4 * 18, ZP = 2 * 16;                 // raw pixels per chunk
    constexpr int XPIECES = XP / 4, ZPIECES = ZP / 4;       // 1-KB DMA pieces (4 pixels x 64 channels): 18 + 8
    constexpr int KPW = 7;                                  // pieces per wave per chunk (4 x 7 = 28 slots, 2 of them dummies)
    constexpr int BUF = 4 * KPW * 256;                      // floats per ring slot (28 KB)
    constexpr int NSLOT = 5, LEAD = NSLOT - 1;              // chunks in flight: HBM latency under load exceeds one chunk time
    __shared__ __attribute__((aligned(1024))) float smem[NSLOT * BUF];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mi = wv & 1, ni = wv >> 1;
    const int li = lane & 31, lh = lane >> 5;
    // Workgroups are dealt round-robin to the 8 XCDs (each with its own L2).  Renumber so that one XCD holds consecutive
    // logical ids = the (ci, co) tiles of as few splits as possible: tiles of one split read the same x / dz chunks, so
    // they should hit one L2 rather than pull the chunk over the fabric once per XCD.
    int bid = blockIdx.x;
#if UNET_ABLATE != 6
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
#endif
    const int tmn = bid % (p.mt * p.nt);
    const int split = bid / (p.mt * p.nt);
    const int m0 = (tmn / p.nt) * 64, n0 = (tmn % p.nt) * 64;
    const int Th = p.H >> 1;

    f32x16 acc[16];
#pragma unroll
    for (int x = 0; x < 16; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;

    // DMA geometry, chunk-invariant: piece 4k+wv of wave wv; pieces 0..17 are x rows (pixels (-1,-1)..(2,16) relative to
    // the chunk's first output pixel), 18..25 dz rows, 26..27 dummies that re-read an always-valid x pixel into an unused LDS
    // piece (so every wave issues exactly KPW DMAs per chunk).  The loads are buffer loads (`buffer_load_dwordx4 ... offen lds`):
    // the chunk's base address is SCALAR (two descriptors per chunk, x and dz), a lane keeps one 32-bit byte offset per piece,
    // and a halo pixel outside the image is given an offset the descriptor's range check rejects -- the hardware then writes
    // zeros into LDS (checked on gfx950), so no zero page, no 64-bit per-lane address and no select between two pointers.
    // Per lane and piece that leaves a 4-bit border code (top / bottom / left / right pixel of the halo) and three VALU
    // instructions per chunk; with one wave per SIMD every VALU instruction is matrix-pipe time (round 5: 45 -> 22 here).
    const int dq = lane & 15, dp = lane >> 4;
    const int wlast = p.W - 16 * (p.tbx - 1);      // image columns covered by the last chunk of a tile row (16 unless W is ragged)
    unsigned g_off[KPW];                 // byte offset from the chunk base (x: pixel (-1,-1); dz: pixel (0,0))
    unsigned g_codes = 0;                // 4 bits per piece, from bit 4k+1: pixel lies in the top row / bottom row / left column / beyond the last image
                                         // column (2, 4, 8, 16) of a chunk at the corresponding image border.  (From bit 1, not 0: the scalar
                                         // side's `cond ? 1 : 0` is turned into a vector-ALU zero-extension by the compiler, `cond ? 2 : 0` is not.)
#pragma unroll
    for (int k = 0; k < KPW; ++k) {
        const int piece = 4 * k + wv;
        if (piece < XPIECES) {
            const int px = 4 * piece + dp, row = px / 18, col = px % 18;          // halo coordinates 0..3 x 0..17
            g_off[k] = (unsigned)(((row * p.W + col) * p.ldx + 4 * dq) * 4);
            g_codes |= (unsigned)((row == 0) | ((row == 3) << 1) | ((col == 0) << 2) | ((col > wlast) << 3)) << (4 * k + 1);
        } else if (piece < XPIECES + ZPIECES) {
            const int px = 4 * (piece - XPIECES) + dp;
            g_off[k] = (unsigned)((((px >> 4) * p.W + (px & 15)) * p.lddz + 4 * dq) * 4);
            g_codes |= (unsigned)(((px & 15) >= wlast) << 3) << (4 * k + 1);
        } else {
            g_off[k] = (unsigned)(((p.W + 1) * p.ldx + 4 * dq) * 4);              // halo pixel (1,1): never out of bounds
        }
    }
    // Chunk bases, all scalar.  A workgroup takes chunks split, split + splits, ...; the chunk's first output pixel is
    //     pix = (img H + 2 ty) W + 16 txb,
    // and going `splits` chunks on adds a constant number of pixels plus (2 W - 16 tbx) each time the column-block digit wraps
    // (the wrap of the tile-row digit into the image digit adds H W - 2 Th W = 0): one running pixel index, two 32 x 32 -> 64
    // bit multiplications per chunk for the byte offsets, no division.  Past the last chunk the bases fall back to chunk 0 with every border
    // code raised: only pixels certainly inside image 0 are read, and the data is never used.
    const int d_txb = p.splits % p.tbx, d_ty = (p.splits / p.tbx) % Th;
    const unsigned xpix = (unsigned)p.ldx * 4u, zpix = (unsigned)p.lddz * 4u;      // bytes per pixel
    const int dpix0 = ((p.splits / (p.tbx * Th)) * p.H + 2 * d_ty) * p.W + 16 * d_txb;
    const int dpix1 = 2 * p.W - 16 * p.tbx;
    const char* const x0 = reinterpret_cast<const char*>(p.x + m0) - (long long)(p.W + 1) * xpix;   // chunk 0's halo pixel (-1,-1)
    const char* const z0 = reinterpret_cast<const char*>(p.dz + n0);
    // (the initial digits come out of integer divisions, which the compiler expands on the vector ALU: readfirstlane keeps the
    // loop-carried state in scalar registers)
    int ic = split, i_txb, i_ty, pix;
    {
        int t = ic; i_txb = __builtin_amdgcn_readfirstlane(t % p.tbx); t /= p.tbx; i_ty = __builtin_amdgcn_readfirstlane(t % Th);
        const int img = __builtin_amdgcn_readfirstlane(t / Th);
        pix = (img * p.H + 2 * i_ty) * p.W + 16 * i_txb;          // < N H W, which the host keeps below 2^31
    }
    constexpr unsigned kRecords = 0x40000000u, kRejected = 0x80000000u;   // every real offset is far below 1 GB (host-checked)
    // what the NEXT batch needs: the descriptors of its x and dz rows and of the two pieces that are x rows in two waves and dz
    // rows in the other two, and its border mask.  A descriptor is four scalar registers: base[31:0], base[47:32] (stride 0),
    // the record count, the gfx9 raw-buffer format word.
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 r_x, r_z, r_4, r_6;
    unsigned b_mask;
    auto descriptor = [&](const char* base) {
        const unsigned long long a = (unsigned long long)(uintptr_t)base;
        return u32x4{(unsigned)a, (unsigned)(a >> 32) & 0xFFFFu, kRecords, 0x00020000u};
    };
    auto next_scalars = [&]() {
        const bool cv = ic < p.nchunks;
        const unsigned scode = (i_ty == 0 ? 2u : 0u) | (i_ty == Th - 1 ? 4u : 0u) | (i_txb == 0 ? 8u : 0u) | (i_txb == p.tbx - 1 ? 16u : 0u);
        b_mask = (cv ? scode : 0x1Eu) * 0x01111111u;
        const unsigned cp = cv ? (unsigned)pix : 0u;
        r_x = descriptor(x0 + (unsigned long long)cp * xpix);
        r_z = descriptor(z0 + (unsigned long long)cp * zpix);
        r_4 = wv < 2 ? r_x : r_z;                     // pieces 16, 17 are x rows, 18, 19 dz rows
        r_6 = wv < 2 ? r_z : r_x;                     // pieces 24, 25 are dz rows, 26, 27 the dummies
        ic += p.splits;
        i_txb += d_txb; const bool c1 = i_txb >= p.tbx; i_txb -= c1 ? p.tbx : 0;
        i_ty += d_ty + (c1 ? 1 : 0); i_ty -= i_ty >= Th ? Th : 0;
        pix += dpix0 + (c1 ? dpix1 : 0);
    };
    unsigned voff[KPW];                  // this lane's offsets for the next batch
    auto next_offsets = [&]() {
        unsigned hit;                                 // (asm: takes the mask as a scalar operand where it is; the compiler's own `and` first copies it to a vector register in the MFMA block)
        asm("v_and_b32 %0, %1, %2" : "=v"(hit) : "s"(b_mask), "v"(g_codes));
#pragma unroll
        for (int k = 0; k < KPW; ++k) voff[k] = (hit & (0xFu << (4 * k + 1))) ? kRejected : g_off[k];
        asm volatile("" : "+v"(voff[0]), "+v"(voff[1]), "+v"(voff[2]), "+v"(voff[3]), "+v"(voff[4]), "+v"(voff[5]), "+v"(voff[6]));
    };
    // piece k of this wave into the ring slot whose LDS byte address (+ this wave's 1 KB) is ldsw.  M0 is written in the same statement
    // that uses it (nothing else in this kernel needs M0); asm so that the descriptor quads stay exactly where the scalar stages below
    // left them -- the builtin rebuilds each descriptor from a pointer in front of the transforms (10 scalar instructions per chunk).
#define WG_DMA(k, ldsw) asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" \
        :: "v"(voff[k]), "s"((k) < 4 ? r_x : (k) == 4 ? r_4 : (k) == 5 ? r_z : r_6), "s"(ldsw), "n"(4096 * (k)) : "memory", "scc")
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_g*)smem;
    const unsigned lds_wv = lds0 + 1024u * (unsigned)wv;

    int c = split;
#pragma unroll
    for (int k = 0; k < LEAD; ++k) {
        next_scalars(); next_offsets();
        const unsigned ldsw = lds_wv + (unsigned)k * (BUF * 4u);
        WG_DMA(0, ldsw); WG_DMA(1, ldsw); WG_DMA(2, ldsw); WG_DMA(3, ldsw); WG_DMA(4, ldsw); WG_DMA(5, ldsw); WG_DMA(6, ldsw);
    }
    next_scalars();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int xa = (8 * lh) * 64 + 32 * mi + li;                    // + (row*18 + cc)*64
    const int za = XP * 64 + (8 * lh) * 64 + 32 * ni + li;          // + (row*16 + cc)*64
    // Raw rows of this lane's 4 tiles: x columns 8*lh .. 8*lh+9 (4 rows), dz columns 8*lh .. 8*lh+7 (2 rows), as column pairs
    // (one ds_read2st64_b32 each).  The LDS reads are inline asm on purpose: left to itself the compiler sinks each read next
    // to its first use and waits on it there, which with one wave per SIMD exposes the LDS latency ~16 times per chunk
    // (WAIT_INST_ANY 0.61, matrix pipe 50 % busy).  Here the whole batch for chunk i+1 is issued behind chunk i's last
    // transform and lands under its last 16 MFMAs; READ_PAIR ties one MFMA operand to each read so those MFMAs stay behind it,
    // and the single lgkmcnt(0) at the top of the next iteration carries every pair as an operand so no use can move above it.
    f32x2 xp[4][5], zp[2][4];
    const unsigned xa_b = lds0 + 4u * xa, za_b = lds0 + 4u * za;
#define READ_PAIR(dst, base, off) \
    asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=&v"(dst) : "v"(base), "n"(off), "n"((off) + 1))
#define ALL_PAIRS \
    "+v"(xp[0][0]), "+v"(xp[0][1]), "+v"(xp[0][2]), "+v"(xp[0][3]), "+v"(xp[0][4]), "+v"(xp[1][0]), "+v"(xp[1][1]), \
    "+v"(xp[1][2]), "+v"(xp[1][3]), "+v"(xp[1][4]), "+v"(xp[2][0]), "+v"(xp[2][1]), "+v"(xp[2][2]), "+v"(xp[2][3]), \
    "+v"(xp[2][4]), "+v"(xp[3][0]), "+v"(xp[3][1]), "+v"(xp[3][2]), "+v"(xp[3][3]), "+v"(xp[3][4]), "+v"(zp[0][0]), \
    "+v"(zp[0][1]), "+v"(zp[0][2]), "+v"(zp[0][3]), "+v"(zp[1][0]), "+v"(zp[1][1]), "+v"(zp[1][2]), "+v"(zp[1][3])
    float V[4][16], M[4][16];
#pragma unroll
    for (int i = 0; i < 32; ++i) { V[i >> 4][i & 15] = 0.f; M[i >> 4][i & 15] = 0.f; }
    // read number k (0..27) of a batch
#define READ_K(k, xb, zb) do { \
        if ((k) < 20) READ_PAIR(xp[(k) / 5][(k) % 5], xb, ((k) / 5) * 18 + 2 * ((k) % 5)); \
        else READ_PAIR(zp[((k) - 20) / 4][((k) - 20) % 4], zb, (((k) - 20) / 4) * 16 + 2 * (((k) - 20) % 4)); \
    } while (0)
    {
#pragma unroll
        for (int k = 0; k < 28; ++k) READ_K(k, xa_b, za_b);
    }
    // ring positions as scalar byte offsets: the slot the next chunk's raw rows are read from and the slot the next batch of DMAs
    // fills (LEAD - 1 slots further on), both advanced behind the MFMAs
    unsigned rd_off = 0, wr_ldsw = lds_wv + (unsigned)(LEAD - 1) * (BUF * 4u);
    auto ring_step = [&]() {
        rd_off = rd_off + BUF * 4u == NSLOT * BUF * 4u ? 0u : rd_off + BUF * 4u;
        wr_ldsw = wr_ldsw + BUF * 4u == lds_wv + NSLOT * BUF * 4u ? lds_wv : wr_ldsw + BUF * 4u;
    };
    ring_step();
#if UNET_ABLATE == 7        /* diagnostics only: s_memtime stamps around the phases of one iteration (block 0, wave 0) */
    long long tl[5] = {0, 0, 0, 0, 0}, ts0, ts1, ts2, ts3, ts4 = 0;
#define STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory")
#else
#define STAMP(t)
#endif
    // One wave per SIMD (the 16 point-accumulators take all 256 AGPRs), and measured on gfx950 (scripts/micro/mfma_issue_cost)
    // a VALU instruction between two MFMAs of the same wave is not hidden: it costs its own ~4 cycles plus ~10 each time the
    // stream switches from the matrix pipe to the VALU and back.  Only asynchronous work (LDS reads, DMA) hides under MFMAs.
    // So per chunk: all four tiles' transforms first, in one VALU block; then the LDS reads for the next chunk and this
    // wave's share of the DMA for the chunk LEAD ahead are issued; then 64 MFMAs back to back with both in flight.
    for (; c < p.nchunks; c += p.splits) {
        STAMP(ts0);
        asm volatile("s_waitcnt lgkmcnt(0)" : ALL_PAIRS);
        STAMP(ts1);
        // Packed-fp32 transforms: a register pair holds two adjacent columns.  Row 3 and column 3 of both B^T d B and
        // A dY A^T are produced with flipped sign (the products V*M are unchanged), which makes every step one v_pk_add_f32:
        // plain adds / subs for the row stage, two op_sel / neg forms for the column stage (inline asm: the compiler does
        // not form them; their distance to the MFMAs that read the results is the barrier below, so no hazard NOPs are owed).
        {
            f32x2 T[4][5];
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                T[0][j] = xp[0][j] - xp[2][j]; T[1][j] = xp[1][j] + xp[2][j];
                T[2][j] = xp[2][j] - xp[1][j]; T[3][j] = xp[3][j] - xp[1][j];
            }
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f32x2 v03, v12;
                    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[1,0]" : "=v"(v03) : "v"(T[i][sidx]), "v"(T[i][sidx + 1]));
                    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(v12) : "v"(T[i][sidx + 1]), "v"(T[i][sidx]));
                    V[sidx][4 * i + 0] = v03.x; V[sidx][4 * i + 3] = v03.y; V[sidx][4 * i + 1] = v12.x; V[sidx][4 * i + 2] = v12.y;
                }
                const f32x2 R[4] = {zp[0][sidx], zp[0][sidx] + zp[1][sidx], zp[0][sidx] - zp[1][sidx], zp[1][sidx]};
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    f32x2 m12;
                    asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(m12) : "v"(R[a]));
                    M[sidx][4 * a + 0] = R[a].x; M[sidx][4 * a + 1] = m12.x; M[sidx][4 * a + 2] = m12.y; M[sidx][4 * a + 3] = R[a].y;
                }
            }
        }
        next_offsets();                               // 7 DMA offsets for the chunk LEAD ahead: VALU work, so it belongs here
        unsigned xb = xa_b + rd_off, zb = za_b + rd_off;      // and the two read bases
        asm volatile("" : "+v"(xb), "+v"(zb));
        __builtin_amdgcn_sched_barrier(0);            // keep the VALU block out of the MFMA block
        STAMP(ts2);
        asm volatile("s_waitcnt vmcnt(14)\n\ts_barrier" ::: "memory");
        static_assert((LEAD - 2) * KPW == 14, "vmcnt immediate");
        STAMP(ts3);
#if UNET_ABLATE == 7
        if (ts4) tl[3] += ts0 - ts4;
        tl[0] += ts1 - ts0; tl[1] += ts2 - ts1; tl[2] += ts3 - ts2; tl[4] += 1; ts4 = ts3;
#endif
        // 64 MFMAs as an explicit stream.  Behind each of the first 28: one LDS read of the next chunk's raw rows (stale data
        // past the last chunk, never used; a burst of 28 would stall on the 15-deep LDS counter).  Behind every 4th from the
        // 31st on: one DMA (an LDS-DMA load holds the vector-memory issue path ~64 cycles: back to back they stall, spread out
        // they cost ~15 cycles each, scripts/micro/mfma_issue_cost).
        // Order: points 0, 3, 12, 15 of the dz transform ARE raw dz values (corners of A dY A^T), i.e. the registers the dz reads
        // overwrite; their 16 MFMAs go first and the dz reads behind them, so no register copy has to keep them alive (the
        // compiler otherwise puts 2-10 v_mov_b64 into this block at ~14 cycles each).  The other 48 follow tile by tile.
        static_for<64>([&](auto step) {
            constexpr int n = decltype(step)::value;
            constexpr int r16 = n & 3, r48 = (n + 32) % 12;        // (n + 32 = n - 16 mod 12, non-negative)
            constexpr int ms = n < 16 ? n >> 2 : (n - 16) / 12;
            constexpr int mp = n < 16 ? (r16 == 0 ? 0 : r16 == 1 ? 3 : r16 == 2 ? 12 : 15) : (r48 < 2 ? r48 + 1 : r48 < 10 ? r48 + 2 : r48 + 3);
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[mp]) : "v"(V[ms][mp]), "v"(M[ms][mp]) : "memory");
            if constexpr (n < 16) READ_K(n, xb, zb);                        // x pairs 0..15
            else if constexpr (n < 24) READ_K(n + 4, xb, zb);               // the 8 dz pairs
            else if constexpr (n < 28) READ_K(n - 8, xb, zb);               // x pairs 16..19
            if constexpr (n >= 30 && ((n - 30) & 3) == 0 && ((n - 30) >> 2) < KPW) WG_DMA((n - 30) >> 2, wr_ldsw);
            // The scalar bases of the batch after that one, in five gaps behind the last DMA: ~60 scalar instructions that a
            // single wave per SIMD would otherwise issue one at a time in front of the transforms (round 4: 105 scalar
            // instructions per chunk outside the MFMA block, ~400 cycles of 5 900).  Each stage is fenced by empty asm
            // statements on its inputs and outputs, which keeps it between the two neighbouring MFMAs.
            if constexpr (n == 55) {
                asm volatile("" : "+s"(ic), "+s"(i_txb), "+s"(i_ty));
                const unsigned scode = (i_ty == 0 ? 2u : 0u) | (i_ty == Th - 1 ? 4u : 0u) | (i_txb == 0 ? 8u : 0u) | (i_txb == p.tbx - 1 ? 16u : 0u);
                b_mask = (ic < p.nchunks ? scode : 0x1Eu) * 0x01111111u;
                asm volatile("" : "+s"(b_mask));
            }
            if constexpr (n == 56) {
                asm volatile("" : "+s"(ic), "+s"(pix));
                const unsigned cp = ic < p.nchunks ? (unsigned)pix : 0u;
                r_x = descriptor(x0 + (unsigned long long)cp * xpix);
                r_z = descriptor(z0 + (unsigned long long)cp * zpix);
                asm volatile("" : "+s"(r_x), "+s"(r_z));
            }
            if constexpr (n == 57) {
                asm volatile("" : "+s"(r_x), "+s"(r_z));
                r_4 = wv < 2 ? r_x : r_z;
                r_6 = wv < 2 ? r_z : r_x;
                asm volatile("" : "+s"(r_4), "+s"(r_6));
            }
            if constexpr (n == 58) {
                asm volatile("" : "+s"(ic), "+s"(i_txb), "+s"(i_ty), "+s"(pix));
                ic += p.splits;
                i_txb += d_txb; const bool c1 = i_txb >= p.tbx; i_txb -= c1 ? p.tbx : 0;
                i_ty += d_ty + (c1 ? 1 : 0); i_ty -= i_ty >= Th ? Th : 0;
                pix += dpix0 + (c1 ? dpix1 : 0);
                asm volatile("" : "+s"(ic), "+s"(i_txb), "+s"(i_ty), "+s"(pix));
            }
            if constexpr (n == 59) {
                asm volatile("" : "+s"(rd_off), "+s"(wr_ldsw));
                ring_step();
                asm volatile("" : "+s"(rd_off), "+s"(wr_ldsw));
            }
        });
    }
    // retire the last read batch (its registers are dead to the compiler, not to the LDS) and the dummy tail DMAs
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" : ALL_PAIRS :: "memory");       // (+ MFMA -> accumulator read distance:
#undef READ_PAIR                                                                                           //  inline-asm MFMAs are invisible to the hazard recogniser)
#undef READ_K
#undef ALL_PAIRS
#if UNET_ABLATE == 7
    if (blockIdx.x == 0 && tid == 0) {
        long long* o = reinterpret_cast<long long*>(p.ws + (size_t)p.splits * 9 * p.Ci * p.Co);
        for (int i = 0; i < 5; ++i) o[i] = tl[i];
    }
#endif

    // epilogue: dw[a][b] = (G^T dU G)[a][b], lane-local per (ci, co)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        float sv[3][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sv[0][j] = acc[0 + j][r] + 0.5f * (acc[4 + j][r] + acc[8 + j][r]);
            sv[1][j] = 0.5f * (acc[4 + j][r] - acc[8 + j][r]);
            sv[2][j] = 0.5f * (acc[4 + j][r] + acc[8 + j][r]) + acc[12 + j][r];
        }
        float* o = p.ws + (((size_t)split * 9) * p.Ci + m0 + 32 * mi + row) * p.Co + n0 + 32 * ni + li;
        const size_t tapstride = (size_t)p.Ci * p.Co;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            o[(3 * a + 0) * tapstride] = sv[a][0] + 0.5f * (sv[a][1] + sv[a][2]);
            o[(3 * a + 1) * tapstride] = 0.5f * (sv[a][1] - sv[a][2]);
            o[(3 * a + 2) * tapstride] = 0.5f * (sv[a][1] + sv[a][2]) + sv[a][3];
        }
    }
}

// dw = sum over the split partials in a fixed order: a block covers 256/SL consecutive float4 outputs x SL slices of the split
// range (SL a power of two <= 16, chosen on the host so that a small weight tensor with many splits -- 64 x 64 channels, 256
// splits: 36 blocks of serial 256-term sums before -- still fills the chip); each thread sums its slice front to back, the
// first thread of an output adds the slice sums front to back.
__global__ __launch_bounds__(256) void wino_partial_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, long n4, int splits, int sl) {
    __shared__ f32x4 part[256];
    const int per = 256 / sl;
    const int o = threadIdx.x % per, sj = threadIdx.x / per;
    const long i = (long)blockIdx.x * per + o;
    const int k0 = (int)((long)splits * sj / sl), k1 = (int)((long)splits * (sj + 1) / sl);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i < n4)
        for (int k = k0; k < k1; ++k) s += reinterpret_cast<const f32x4*>(ws)[(size_t)k * n4 + i];
    if (sl == 1) { if (i < n4) reinterpret_cast<f32x4*>(dw)[i] = s; return; }
    part[threadIdx.x] = s;
    __syncthreads();
    if (sj == 0 && i < n4) {
        for (int j = 1; j < sl; ++j) s += part[j * per + o];
        reinterpret_cast<f32x4*>(dw)[i] = s;
    }
}

int wgrad_fused_splits(int N, int H, int W, int Ci, int Co, int max_workgroups) {
    const int blocks = (Ci / 64) * (Co / 64);
    const long nchunks = (long)N * (H / 2) * ((W / 2 + 7) / 8);
    // Workgroups on the chip: one per CU by default (max_workgroups = 0).  A caller's cap of 224 leaves ~4 CUs per XCD to the OTHER
    // stream -- the main stream's BatchNorm passes and, data-parallel, RCCL's kernels cannot share a CU with these 512-register workgroups
    // and otherwise queue behind a chip-filling grid: the single-GPU step gains 0.3 ms of 44.1 (the passes then run beside the weight
    // gradient, but at 1.3-2.1 TB/s: DESIGN.md), while the kernel itself is 12 % slower when timed alone.
    const int want = (max_workgroups >= 32 && max_workgroups <= 256) ? max_workgroups : 256;
    long sp = 256 / blocks; if (sp < 1) sp = 1;
    if (blocks * sp > want && blocks * (want / blocks) >= 192) sp = want / blocks;
    if (sp > nchunks) sp = nchunks;
    return (int)sp;
}

int grid_for(long total, int cap) { long b = (total + 255) / 256; if (b > cap) b = cap; if (b < 1) b = 1; return (int)b; }


// ---- BatchNorm-apply on load for the fused forward kernel (SURVEY.md 7 "hard parts": the consumer applies the producer's
// BatchNorm on load and padded positions must contribute exactly 0) ---------------------------------------------------------------
// The producer's BatchNorm output y = s[c] * r + t[c] feeds this layer's convolution (UNet/model.py:36 -> :30), zero padded.  Per
// input channel the scale commutes with the Winograd transforms and the shift is a constant image, so
//     conv(y; W) = conv(r; s . W)  +  sum_taps t . W           at every pixel whose 3x3 window lies inside the image,
// and at the border the taps that fall outside must not see the shift.  Reading r with the per-channel padding value
// pad[c] = -t[c] / s[c] instead of 0 makes a padded tap contribute s W pad = -t W, which cancels the uniform shift term
// exactly where the zero padding of y would have contributed nothing.  So the unchanged kernel, given
//     Uc' = s . transform(W),   bias' = bias + sum_{taps, c} t[c] W[tap][c][:],   pad,
// computes the convolution of the BatchNorm OUTPUT without that tensor ever being written or read.
// Conditioning.  pad = -t / s grows as the BatchNorm scale s = gamma / sqrt(var + eps) shrinks (a dead / pruned channel: gamma ~ 0 with
// beta = O(1)), but every use of it is multiplied by s again: the input transform's rounding error is |pad| * 2^-24-ish, the products carry
// the factor s, so the error of the layer output is ~ |t| |W| 2^-24 per channel whatever |s| is -- relative to the shift term itself,
// never to the vanishing s * r term (tests: |gamma| in {0, 1e-30, 1e-6, 1e-4, 1e-2}, both signs, against the fp64 oracle at the usual
// 2e-5).  Only s == 0 has no padding value at all and |pad| > ~1e37 overflows the 4-term sums of the transform; both are fenced HERE,
// on the device, every time the fold runs (every step): |s| is floored at kFoldScaleFloor * max(1, |t|) -- the folded layer then
// computes with s' instead of s, an absolute change of the output below 1e-30 * sum |r W|, far under fp32 resolution of any output
// the shift term reaches -- so pad is always finite, |pad| <= 1e30, and no host-side gamma check or fallback route is needed.
//
// one thread = 8 consecutive input channels x one output channel (the batch kernel's work item): forward layout only
constexpr float kFoldScaleFloor = 1e-30f;
__device__ __forceinline__ float fold_scale(float sc, float sh) {
    const float smin = kFoldScaleFloor * fmaxf(1.f, fabsf(sh));
    return fabsf(sc) < smin ? copysignf(smin, sc) : sc;
}

// One launch: a workgroup owns kFoldCo output channels and ALL input-channel groups, so the folded bias (a sum over the input channels)
// is finished inside it -- thread = (output channel, one of 64 lanes over the 8-channel groups), fixed-order double sums.  (Round 3: the
// separate bias kernel was one of three tiny launches between two persistent convolutions, each worth 25 us of chain in the fp32 step.)
constexpr int kFoldCo = 4, kFoldLanes = 256 / kFoldCo;
__global__ __launch_bounds__(256) void wino_weight_fold_kernel(const float* __restrict__ w, const float* __restrict__ scale,
        const float* __restrict__ shift, const float* __restrict__ bias, float* __restrict__ uf, float* __restrict__ bias_out,
        float* __restrict__ pad, int Ci, int Co) {
    __shared__ double sPart[kFoldLanes][kFoldCo];
    const size_t plane = (size_t)Ci * Co;
    const int col = threadIdx.x % kFoldCo, gl = threadIdx.x / kFoldCo;
    const int co = blockIdx.x * kFoldCo + col;
    const bool live = co < Co;
    double bsum = 0.0;
    for (int c8 = gl; c8 < (Ci >> 3) && live; c8 += kFoldLanes) {
        float t[16][8];
        float tsum = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ci = 8 * c8 + e;
            const float sh = shift[ci], sc = fold_scale(scale[ci], sh);
            float g[3][3], gs = 0.f;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) { const float v = w[((size_t)(a * 3 + b) * Ci + ci) * Co + co]; gs += v; g[a][b] = sc * v; }
            tsum = fmaf(sh, gs, tsum);
            float s[4][3];
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                s[0][b] = g[0][b];
                s[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
                s[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
                s[3][b] = g[2][b];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                t[4 * r + 0][e] = s[r][0]; t[4 * r + 1][e] = 0.5f * (s[r][0] + s[r][1] + s[r][2]);
                t[4 * r + 2][e] = 0.5f * (s[r][0] - s[r][1] + s[r][2]); t[4 * r + 3][e] = s[r][2];
            }
            if (co == 0) pad[ci] = -sh / sc;
        }
        const size_t offf = ((size_t)c8 * Co + co) * 8;                                   // [k/8][n=co][k%8]
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) {
            float* o = uf + (size_t)xi * plane + offf;
            *reinterpret_cast<f32x4*>(o) = f32x4{t[xi][0], t[xi][1], t[xi][2], t[xi][3]};
            *reinterpret_cast<f32x4*>(o + 4) = f32x4{t[xi][4], t[xi][5], t[xi][6], t[xi][7]};
        }
        bsum += (double)tsum;                                           // shift term of this 8-channel group
    }
    sPart[gl][col] = bsum;
    __syncthreads();
    if (gl == 0 && live) {
        double sacc = 0.0;
#pragma unroll
        for (int l = 0; l < kFoldLanes; ++l) sacc += sPart[l][col];
        bias_out[co] = (float)((double)(bias ? bias[co] : 0.f) + sacc);
    }
    if (blockIdx.x == 0 && threadIdx.x < 8) pad[Ci + threadIdx.x] = 0.f;             // the 8 floats the halo pointers may walk past the end
}

// ---- weight gradient of a layer whose input was read through BatchNorm-apply on load ----------------------------------------------
// With x = s . r + t inside the image (0 outside), dW[a][b][ci][co] = sum_p x[p + (a,b) - 1][ci] dz[p][co]
//     = s[ci] * (the kernel's result on the raw r)  +  t[ci] * S[a][b][co],   S = sum of dz over the pixels p whose tap (a, b) lies
// inside the image = total - (row excluded by a) - (column excluded by b) + (their corner): a = 0 excludes dz row 0, a = 2 row H-1,
// b = 0 column 0, b = 2 column W-1.  dz_border_sums_kernel leaves the 4 line sums and 4 corner sums per channel (fp64 accumulation,
// fixed order); wgrad_fold_fix_kernel applies the two terms to dW in place.
constexpr int kBorderSegs = 32;            // segments a border line is split into (one workgroup each)

// partial sums: part[line][seg][Co]; line 0..3 = row 0, row H-1, column 0, column W-1 (over all images)
__global__ __launch_bounds__(256) void dz_border_sums_kernel(const float* __restrict__ dz, int lddz, int N, int H, int W, int Co,
                                                             float* __restrict__ part) {
    // block = (line, 64-channel group, segment); thread = (channel quad q, pixel lane pl of 16)
    const int line = blockIdx.x & 3, cg = (blockIdx.x >> 2) % ((Co + 63) / 64), seg = (blockIdx.x >> 2) / ((Co + 63) / 64);
    const int c0 = cg * 64, q = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int len = line < 2 ? W : H;
    const long total = (long)N * len, per = (total + kBorderSegs - 1) / kBorderSegs;
    const long i0 = seg * per; long i1 = i0 + per; if (i1 > total) i1 = total;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    const int c = c0 + 4 * q;
    if (c < Co) {
        for (long i = i0 + pl; i < i1; i += 16) {
            const int n = (int)(i / len), k = (int)(i % len);
            const int y = line == 0 ? 0 : (line == 1 ? H - 1 : k), x = line == 2 ? 0 : (line == 3 ? W - 1 : k);
            const f32x4 v = *reinterpret_cast<const f32x4*>(dz + ((size_t)(n * H + y) * W + x) * lddz + c);
            acc[0] += (double)v[0]; acc[1] += (double)v[1]; acc[2] += (double)v[2]; acc[3] += (double)v[3];
        }
    }
    __shared__ double red[16][64];
#pragma unroll
    for (int e = 0; e < 4; ++e) red[pl][4 * q + e] = acc[e];
    __syncthreads();
    if (threadIdx.x < 64 && c0 + threadIdx.x < Co) {
        double s = 0.0;
        for (int k = 0; k < 16; ++k) s += red[k][threadIdx.x];
        part[((size_t)line * kBorderSegs + seg) * Co + c0 + threadIdx.x] = (float)s;
    }
}

// sums[8][Co]: the four line sums (segments added in order) and the four corner sums
__global__ __launch_bounds__(256) void dz_border_finalize_kernel(const float* __restrict__ part, const float* __restrict__ dz, int lddz,
                                                                 int N, int H, int W, int Co, float* __restrict__ sums) {
    const int ch = blockIdx.x * 256 + threadIdx.x;
    if (ch >= Co) return;
#pragma unroll
    for (int line = 0; line < 4; ++line) {
        double s = 0.0;
        for (int k = 0; k < kBorderSegs; ++k) s += (double)part[((size_t)line * kBorderSegs + k) * Co + ch];
        sums[(size_t)line * Co + ch] = (float)s;
    }
    double k00 = 0.0, k01 = 0.0, k10 = 0.0, k11 = 0.0;
    for (int n = 0; n < N; ++n) {
        k00 += (double)dz[((size_t)(n * H + 0) * W + 0) * lddz + ch];         k01 += (double)dz[((size_t)(n * H + 0) * W + W - 1) * lddz + ch];
        k10 += (double)dz[((size_t)(n * H + H - 1) * W + 0) * lddz + ch];     k11 += (double)dz[((size_t)(n * H + H - 1) * W + W - 1) * lddz + ch];
    }
    sums[(size_t)4 * Co + ch] = (float)k00; sums[(size_t)5 * Co + ch] = (float)k01;
    sums[(size_t)6 * Co + ch] = (float)k10; sums[(size_t)7 * Co + ch] = (float)k11;
}

__global__ __launch_bounds__(256) void wgrad_fold_fix_kernel(float* __restrict__ dw, const float* __restrict__ scale, const float* __restrict__ shift,
        const float* __restrict__ sums, const float* __restrict__ total, int Ci, int Co) {
    const long n4 = (long)9 * Ci * Co / 4;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const long e0 = 4 * i;
    const int co = (int)(e0 % Co); const long rest = e0 / Co; const int ci = (int)(rest % Ci), tap = (int)(rest / Ci);
    const int a = tap / 3, b = tap % 3;
    const float sc = scale[ci], sh = shift[ci];
    f32x4 v = *reinterpret_cast<f32x4*>(dw + e0);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = co + e;
        float S = total[c];
        if (a != 1) S -= sums[(size_t)(a == 0 ? 0 : 1) * Co + c];
        if (b != 1) S -= sums[(size_t)(b == 0 ? 2 : 3) * Co + c];
        if (a != 1 && b != 1) S += sums[(size_t)(4 + (a == 0 ? 0 : 2) + (b == 0 ? 0 : 1)) * Co + c];
        v[e] = fmaf(sc, v[e], sh * S);
    }
    *reinterpret_cast<f32x4*>(dw + e0) = v;
}


int run_wino_fused(const float* x, int ldx, const float* Uc, const float* bias, float* out, int ldo, int N, int H, int W,
                   int K, int Nout, int relu, float* stat_part, hipStream_t st, const WinoBnBwd* bb = nullptr, const float* pad = nullptr,
                   int max_workgroups = 0) {
    WinoFusedArgs a{};
    a.pad = pad;
    a.x = x; a.Uc = Uc; a.bias = bias; a.out = out; a.ldx = ldx; a.ldo = ldo; a.N = N; a.H = H; a.W = W; a.K = K; a.Nout = Nout; a.relu = relu;
    a.tby = (H / 2 + 7) / 8; a.tbx = (W / 2 + 7) / 8; a.nt = Nout / 64; a.stat_part = stat_part;
    const long blocks = (long)N * a.tby * a.tbx * a.nt;
    if (blocks <= 0 || blocks > 0x7fffffffL) return UNET_EINVAL;
    if (K % 16 == 0 && K >= 32) {
        const int cus = unet_grid_slots(wino_stream_cus(), max_workgroups);
        const dim3 grid((unsigned)(blocks < cus ? blocks : cus));
        if (bb) {
            a.bn_r = bb->r; a.bn_ldr = bb->ldr; a.bn_c0 = bb->c0; a.bn_c1 = bb->c1;
            wino_fused_stream_bnbwd_kernel<<<grid, 256, 0, st>>>(a, (int)blocks);
        }
        else if (stat_part) wino_fused_stream_stats_kernel<<<grid, 256, 0, st>>>(a, (int)blocks);
        else                wino_fused_stream_kernel<<<grid, 256, 0, st>>>(a, (int)blocks);
    } else {
        if (stat_part) return UNET_EINVAL;
        wino_fused_kernel<<<dim3((unsigned)blocks), 256, 0, st>>>(a);
    }
    return UNET_LAUNCH_STATUS();
}

}  // namespace

// Fully fused Winograd forward.  Uc = unet_winograd_weight_transform(w, mode 2) / _batch / unet_winograd_weight_fold.
// pad (nullable): per-input-channel value the kernel reads for positions OUTSIDE the image (Cin + 8 floats); null = zero padding.
// With unet_winograd_weight_fold it carries BatchNorm-apply on load: x is the producer's conv output r, the weights are scaled by
// the BatchNorm scale per input channel, the bias takes the shift's contribution and pad = -shift / scale makes every padded
// position contribute exactly what the zero padding of the BatchNorm OUTPUT would.
// stat_part (nullable): BatchNorm statistics of the output, stat_part[Cout/64][rows][64][2] floats (sum, sum of squares per channel
// over the pixels each row's workgroup-half produced), rows = unet_conv3x3_fwd_winograd_fused_stats_rows(...) > 0; consumed by
// unet_bn_train_finalize_partials.
extern "C" int unet_conv3x3_fwd_winograd_fused_stats_rows_wg(int N, int H, int W, int Cin, int Cout, int max_workgroups) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin % 8 != 0 || Cout % 64 != 0 || Cin > kWinoFusedMaxK) return 0;
    return wino_stats_rows(N, H, W, Cin, Cout, max_workgroups);
}
extern "C" int unet_conv3x3_fwd_winograd_fused_stats_rows(int N, int H, int W, int Cin, int Cout) {
    return unet_conv3x3_fwd_winograd_fused_stats_rows_wg(N, H, W, Cin, Cout, 0);
}

// max_workgroups: cap on the persistent grid (common.h unet_grid_slots); the statistics rows follow it (.._stats_rows_wg)
extern "C" int unet_conv3x3_fwd_winograd_fused_wg(const float* x, int ldx, const float* pad, const float* Uc, const float* bias, float* out, int ldo,
        int N, int H, int W, int Cin, int Cout, int relu, float* stat_part, size_t stat_bytes, int max_workgroups, void* stream) {
    UNET_CHECK_ARG(x && Uc && out && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && Cin % 8 == 0 && Cout % 64 == 0);
    UNET_CHECK_ARG(Cin <= kWinoFusedMaxK);
    UNET_CHECK_ARG(ldx >= Cin && ldo >= Cout && ldx % 4 == 0 && ldo % 4 == 0 && unet_aligned16(x) && unet_aligned16(Uc) && unet_aligned16(out));
    UNET_CHECK_ARG((!bias || unet_aligned16(bias)) && (!pad || unet_aligned16(pad)));
    if (stat_part) {
        const int rows = wino_stats_rows(N, H, W, Cin, Cout, max_workgroups);
        UNET_CHECK_ARG(rows > 0);
        if (stat_bytes < (size_t)(Cout / 64) * rows * 128 * sizeof(float)) return UNET_ENOSPC;
    }
    return run_wino_fused(x, ldx, Uc, bias, out, ldo, N, H, W, Cin, Cout, relu, stat_part, (hipStream_t)stream, nullptr, pad, max_workgroups);
}
extern "C" int unet_conv3x3_fwd_winograd_fused(const float* x, int ldx, const float* pad, const float* Uc, const float* bias, float* out, int ldo,
        int N, int H, int W, int Cin, int Cout, int relu, float* stat_part, size_t stat_bytes, void* stream) {
    return unet_conv3x3_fwd_winograd_fused_wg(x, ldx, pad, Uc, bias, out, ldo, N, H, W, Cin, Cout, relu, stat_part, stat_bytes, 0, stream);
}

#if UNET_ABLATE == 8
extern "C" int unet_debug_wf_timeline(long long* out4) {
    return (int)hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_wf_timeline), 64);
}
#endif

// Data gradient (Ucd = weight transform mode 3) and, with r_prev / stat_part (both or neither), the BatchNorm-backward sums of the
// layer that PRODUCED this layer's input: dx channels [c0, c1) (multiples of 64) are that layer's dy, r_prev its saved activation
// (c1 - c0 channels, pixel stride ldr).  stat_part = (Cin/64) * rows * 128 floats, rows = unet_conv3x3_fwd_winograd_fused_stats_rows(
// N, H, W, Cout, Cin); blocks c0/64 .. c1/64 - 1 hold sum(dy) and sum(dy * r) per channel, consumed by unet_bn_bwd_from_partials.
extern "C" int unet_conv3x3_dgrad_winograd_fused_wg(const float* dz, int lddz, const float* Ucd, float* dx, int lddx,
        int N, int H, int W, int Cin, int Cout, const float* r_prev, int ldr, int c0, int c1,
        float* stat_part, size_t stat_bytes, int max_workgroups, void* stream) {
    UNET_CHECK_ARG(dz && Ucd && dx && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && Cout % 8 == 0 && Cin % 64 == 0);
    UNET_CHECK_ARG(Cout <= kWinoFusedMaxK && (r_prev == nullptr) == (stat_part == nullptr));
    UNET_CHECK_ARG(lddz >= Cout && lddx >= Cin && lddz % 4 == 0 && lddx % 4 == 0 && unet_aligned16(dz) && unet_aligned16(Ucd) && unet_aligned16(dx));
    if (!r_prev) return run_wino_fused(dz, lddz, Ucd, nullptr, dx, lddx, N, H, W, Cout, Cin, 0, nullptr, (hipStream_t)stream, nullptr, nullptr, max_workgroups);
    UNET_CHECK_ARG(c0 >= 0 && c1 > c0 && c1 <= Cin && c0 % 64 == 0 && c1 % 64 == 0 && ldr >= c1 - c0 && ldr % 4 == 0 && unet_aligned16(r_prev));
    const int rows = wino_stats_rows(N, H, W, Cout, Cin, max_workgroups);
    UNET_CHECK_ARG(rows > 0);
    if (stat_bytes < (size_t)(Cin / 64) * rows * 128 * sizeof(float)) return UNET_ENOSPC;
    const WinoBnBwd bb{r_prev, ldr, c0, c1};
    return run_wino_fused(dz, lddz, Ucd, nullptr, dx, lddx, N, H, W, Cout, Cin, 0, stat_part, (hipStream_t)stream, &bb, nullptr, max_workgroups);
}
extern "C" int unet_conv3x3_dgrad_winograd_fused(const float* dz, int lddz, const float* Ucd, float* dx, int lddx,
        int N, int H, int W, int Cin, int Cout, const float* r_prev, int ldr, int c0, int c1,
        float* stat_part, size_t stat_bytes, void* stream) {
    return unet_conv3x3_dgrad_winograd_fused_wg(dz, lddz, Ucd, dx, lddx, N, H, W, Cin, Cout, r_prev, ldr, c0, c1, stat_part, stat_bytes, 0, stream);
}

extern "C" int unet_winograd_wgrad_fused_supported(int N, int H, int W, int Cin, int Cout) {
    return (N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && Cin % 64 == 0 && Cout % 64 == 0) ? 1 : 0;
}

extern "C" size_t unet_conv3x3_wgrad_winograd_fused_workspace(int N, int H, int W, int Cin, int Cout, int max_workgroups) {
    return (size_t)wgrad_fused_splits(N, H, W, Cin, Cout, max_workgroups) * 9 * Cin * Cout * sizeof(float);
}

// dw[a][b][ci][co] = sum_{n,y,x} xin[n, y+a-1, x+b-1, ci] * dz[n,y,x,co] via the fused Winograd-domain kernel.
// max_workgroups: cap on the persistent grid (0 or out of [32, 256]: one workgroup per CU); honoured when it still leaves >= 192
extern "C" int unet_conv3x3_wgrad_winograd_fused(const float* xin, int ldx, const float* dz, int lddz, float* dw,
        int N, int H, int W, int Cin, int Cout, int max_workgroups, void* ws, size_t ws_bytes, void* stream) {
    UNET_CHECK_ARG(xin && dz && dw && ws && unet_winograd_wgrad_fused_supported(N, H, W, Cin, Cout));
    UNET_CHECK_ARG(ldx >= Cin && lddz >= Cout && ldx % 4 == 0 && lddz % 4 == 0);
    UNET_CHECK_ARG(unet_aligned16(xin) && unet_aligned16(dz) && unet_aligned16(dw) && unet_aligned16(ws));
    // the kernel's addressing: a 32-bit pixel index, and per-lane byte offsets inside one chunk's halo (4 image rows) that must stay
    // below the 1 GB its buffer descriptors admit
    UNET_CHECK_ARG((long long)N * H * W < (1ll << 31) && (3ll * W + 18) * (ldx > lddz ? ldx : lddz) * 4 < (1ll << 30));
    if (ws_bytes < unet_conv3x3_wgrad_winograd_fused_workspace(N, H, W, Cin, Cout, max_workgroups)) return UNET_ENOSPC;
    hipStream_t st = (hipStream_t)stream;
    WinoWgradArgs a{};
    a.x = xin; a.dz = dz; a.ws = (float*)ws; a.ldx = ldx; a.lddz = lddz; a.N = N; a.H = H; a.W = W; a.Ci = Cin; a.Co = Cout;
    a.mt = Cin / 64; a.nt = Cout / 64; a.tbx = (W / 2 + 7) / 8; a.nchunks = N * (H / 2) * a.tbx;
    a.splits = wgrad_fused_splits(N, H, W, Cin, Cout, max_workgroups);
    wino_wgrad_fused_kernel<<<dim3((unsigned)(a.mt * a.nt * a.splits)), 256, 0, st>>>(a);
    int rc = UNET_LAUNCH_STATUS(); if (rc) return rc;
    const long n4 = 9L * Cin * Cout / 4;
    int sl = 1;
    while (sl < 16 && 2 * sl <= a.splits && n4 * sl < 256 * 1024) sl *= 2;
    wino_partial_reduce_kernel<<<(unsigned)((n4 * sl + 255) / 256), 256, 0, st>>>((const float*)ws, dw, n4, a.splits, sl);
    return UNET_LAUNCH_STATUS();
}


// U must hold 16*Cin*Cout floats.  mode 0: forward kernel transform; mode 1: data-gradient kernel transform.
extern "C" int unet_winograd_weight_transform(const float* w, float* U, int Cin, int Cout, int mode, void* stream) {
    UNET_CHECK_ARG(w && U && Cin > 0 && Cout > 0 && mode >= 0 && mode <= 3 && (mode < 2 || (Cin % 8 == 0 && Cout % 8 == 0)));
    wino_weight_kernel<<<grid_for((long)Cin * Cout, 4096), 256, 0, (hipStream_t)stream>>>(w, U, Cin, Cout, mode);
    return UNET_LAUNCH_STATUS();
}

// BatchNorm-apply on load (see wino_weight_fold_kernel): from the layer's HWIO kernel w, its bias and the PRODUCER layer's BatchNorm
// scale / shift (Cin floats each) make Uc (forward operand of unet_conv3x3_fwd_winograd_fused), bias_out (Cout) and pad (Cin + 8).
extern "C" size_t unet_winograd_weight_fold_workspace(int Cin, int Cout) { return (size_t)(Cin / 8) * Cout * sizeof(float); }
extern "C" int unet_winograd_weight_fold(const float* w, const float* bias, const float* scale, const float* shift, float* Uc, float* bias_out,
                                         float* pad, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream) {
    UNET_CHECK_ARG(w && scale && shift && Uc && bias_out && pad && ws && Cin > 0 && Cin % 8 == 0 && Cout > 0 && Cout % 4 == 0 && unet_aligned16(Uc));
    if (ws_bytes < unet_winograd_weight_fold_workspace(Cin, Cout)) return UNET_ENOSPC;
    (void)ws;                                                          // (round 3: the bias is finished inside the fold kernel; the workspace is unused)
    wino_weight_fold_kernel<<<(unsigned)((Cout + kFoldCo - 1) / kFoldCo), 256, 0, (hipStream_t)stream>>>(w, scale, shift, bias, Uc, bias_out, pad, Cin, Cout);
    return UNET_LAUNCH_STATUS();
}

// Weight gradient of a layer whose input x was read as scale . r + shift (BatchNorm-apply on load): dw holds the kernel's result on the
// RAW r (any wgrad kernel, HWIO layout); in place dw = scale[ci] * dw + shift[ci] * S[tap][co], S from the border sums of dz and
// total = the column sums of dz (the bias gradient).
extern "C" size_t unet_conv3x3_wgrad_fold_fix_workspace(int Cout) { return (size_t)(8 + 4 * kBorderSegs) * Cout * sizeof(float); }
extern "C" int unet_conv3x3_wgrad_fold_fix(float* dw, const float* scale, const float* shift, const float* dz, int lddz, const float* total,
                                           int N, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream) {
    UNET_CHECK_ARG(dw && scale && shift && dz && total && ws && N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cout % 4 == 0 && lddz >= Cout && lddz % 4 == 0);
    UNET_CHECK_ARG(unet_aligned16(dw) && unet_aligned16(dz));
    if (ws_bytes < unet_conv3x3_wgrad_fold_fix_workspace(Cout)) return UNET_ENOSPC;
    hipStream_t st = (hipStream_t)stream;
    float* sums = (float*)ws; float* part = sums + (size_t)8 * Cout;
    dz_border_sums_kernel<<<(unsigned)(4 * ((Cout + 63) / 64) * kBorderSegs), 256, 0, st>>>(dz, lddz, N, H, W, Cout, part);
    int rc = UNET_LAUNCH_STATUS(); if (rc) return rc;
    dz_border_finalize_kernel<<<(unsigned)((Cout + 255) / 256), 256, 0, st>>>(part, dz, lddz, N, H, W, Cout, sums);
    rc = UNET_LAUNCH_STATUS(); if (rc) return rc;
    const long n4 = (long)9 * Cin * Cout / 4;
    wgrad_fold_fix_kernel<<<(unsigned)((n4 + 255) / 256), 256, 0, st>>>(dw, scale, shift, sums, total, Cin, Cout);
    return UNET_LAUNCH_STATUS();
}

extern "C" int unet_winograd_weight_transform_batch(const void* jobs, int njobs, int total_blocks, void* stream) {
    UNET_CHECK_ARG(jobs && njobs > 0 && total_blocks > 0);
    wino_weight_batch_kernel<<<dim3((unsigned)total_blocks), 256, 0, (hipStream_t)stream>>>((const long long*)jobs, njobs);
    return UNET_LAUNCH_STATUS();
}

