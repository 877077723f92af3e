// 2x2 / stride-2 transposed convolution (UNet._deconv_layer, UNet/model.py:39-46) forward, data gradient and weight gradient as GEMMs on
// the BF16 matrix pipe at fp32 grade ("BF16x6", see winograd_x6.hip for the arithmetic: every fp32 operand value is exactly h + m + l with three bf16
// pieces; the six products hh, hm, mh, hl, lh, mm are exact in the fp32 accumulator of v_mfma_f32_32x32x16_bf16, the dropped ones sum
// to at most 2^-21, on average 2^-24.5 of the product).  The fp32 reference layer:
//   forward   z[n,2i+a,2j+b,co] = bias[co] + sum_ci x[n,i,j,ci] * W[a,b,co,ci]          GEMM  M = input pixels, K = Cin,      N = 4 taps x Cout
//   gradient  dx[n,i,j,ci]      = sum_{a,b,co} dz[n,2i+a,2j+b,co] * W[a,b,co,ci]        GEMM  M = input pixels, K = 4 x Cout, N = Cin
//   weights   dW[a,b,co,ci]     = sum_{n,i,j} dz[n,2i+a,2j+b,co] * x[n,i,j,ci]          GEMM  per tap: M = Cout, N = Cin, K = input pixels   (further down)
// Unlike the Winograd kernels there is no transform in front of the products, and one staged A value feeds 128 (256) output columns, so
// the three-piece split is a few vector instructions per 48 (24) MFMAs and the LDS serves 18 (12) fragment reads per 48 (24) MFMAs.
// Workgroup = 128 pixels x NT columns (NT = 256 or 128), 4 waves as 2 (pixel halves) x 2 (column halves), a wave owns 2 x NB blocks of
// 32 x 32 (NB = NT / 64) = 32 NB accumulator registers.  K runs in chunks of 16 through two LDS stages (36 / 24 KB each), so two
// workgroups share a CU and fill each other's stalls: no hand-written instruction stream here.
//   A stage: [piece 3][pixel 128][16 k bf16] -- fp32 rows loaded one chunk ahead, split in registers, ds_write_b128
//   B stage: [piece 3][column NT][16 k bf16] -- LDS-DMA, verbatim from the pre-split weights (unet_convT2x2_weight_transform_x6)
// Rows are 32 bytes; the 16-byte k half sits in slot (half ^ bit 3 of the row): conflict-free ds_read_b128 for the MFMA operand fetch.
#include "common.h"

namespace {

typedef int cx_i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 cx_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned cx_u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void cx_lds_void;

struct ConvtX6Args {
    const float* a;          // forward: x [P][lda]; gradient: dz [N][2H][2W][lda]
    const uint16_t* b6;      // pre-split weights [K/16][piece 3][Ncols][16 bf16], swizzled
    const float* bias;       // forward (nullable)
    float* out;              // forward: [N][2H][2W][ldo]; gradient: dx [P][ldo]
    float* stat_part;        // forward, nullable: BatchNorm sums of the output, [Cout/64][rows][64][2], rows = 4 * P / 128;
                             // gradient (the _bnbwd kernels): BatchNorm-BACKWARD sums of the producer layer (sum dx, sum dx * r), [Cin/64][rows][64][2], rows = P / 128
    const float* bn_r; int bn_ldr;     // gradient + sums: the producer layer's saved activation r [P][bn_ldr] (its dy is dx)
    int lda, ldo, N, H, W, Cin, Cout;
    int K, Ncols, nct;       // reduction length, GEMM columns, column tiles per pixel tile
    long P;
};

// (round 6: pieces by round-to-nearest-even, v_cvt_pk_bf16_f32 -- see winograd_x6.hip: still an exact split, dropped products zero-mean)
typedef __bf16 cx_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cx_rn2(float lo, float hi) {       // { bf16(lo), bf16(hi) } rounded to nearest even, lo in the low half
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, cx_bf16x2));
}
__device__ __forceinline__ float cx_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float cx_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
// a value pair -> its three piece pairs (exact: v = h + m + l)
__device__ __forceinline__ void cx_split2(float v0, float v1, unsigned& h, unsigned& m, unsigned& l) {
    h = cx_rn2(v0, v1);
    const float a0 = v0 - cx_lo(h), a1 = v1 - cx_hi(h);
    m = cx_rn2(a0, a1);
    l = cx_rn2(a0 - cx_lo(m), a1 - cx_hi(m));
}

// 8 fp32 values -> three 16-byte rows of bf16 pieces (exact: v = h + m + l)
__device__ __forceinline__ void cx_split8(const f32x4& v0, const f32x4& v1, cx_i32x4& h, cx_i32x4& m, cx_i32x4& l) {
    const float t[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        unsigned hp, mp, lp;
        cx_split2(t[2 * e], t[2 * e + 1], hp, mp, lp);
        h[e] = (int)hp; m[e] = (int)mp; l[e] = (int)lp;
    }
}

// MODE 0: forward, MODE 1: data gradient.  NB: 32-column blocks per wave (4: 256-column tile, 2: 128-column tile).
template <int MODE, int NB, bool STATS>
__device__ __forceinline__ void convt_x6_body(const ConvtX6Args& p) {
    constexpr int NT = 64 * NB;                                  // columns per workgroup
    constexpr int ASZ = 3 * 128 * 32, BSZ = 3 * NT * 32, STG = ASZ + BSZ;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STG];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv & 1, wn = wv >> 1;
    const int li = lane & 31, lh = lane >> 5;
    const int ct = blockIdx.x % p.nct;
    const long pt = blockIdx.x / p.nct;
    const long p0 = pt * 128;
    const int n0 = ct * NT;

    // ---- A staging: thread = (pixel row, k half); 8 consecutive k per chunk
    const int arow = tid >> 1, akh = tid & 1;
    const float* abase[MODE == 0 ? 1 : 4];
    {
        const long pp = p0 + arow;
        if constexpr (MODE == 0) abase[0] = p.a + pp * p.lda + akh * 8;
        else {
            // input pixel pp = q * W + j (q = image row over the whole batch) -> output pixel (2i+a, 2j+b) has index 4 W q + 2 j + a 2W + b
            const unsigned q = (unsigned)pp / (unsigned)p.W, j = (unsigned)pp - q * (unsigned)p.W;
#pragma unroll
            for (int t = 0; t < 4; ++t)
                abase[t] = p.a + ((long)4 * p.W * q + 2 * j + (t >> 1) * 2 * p.W + (t & 1)) * p.lda + akh * 8;
        }
    }
    auto a_ptr = [&](int c) -> const float* {
        if constexpr (MODE == 0) return abase[0] + c * 16;
        else { const int k0 = c * 16; const int t = k0 / p.Cout; return abase[t] + (k0 - t * p.Cout); }
    };
    const unsigned a_wr = (unsigned)(arow * 32 + 16 * (akh ^ ((arow >> 3) & 1)));
    // ---- B staging: LDS-DMA, NT * 96 bytes per chunk = 3 NT / 32 pieces of 1 KB, spread over the four waves
    constexpr int BPIECES = 3 * NT / 32, BPW = BPIECES / 4;      // 24 / 6 (NT 256), 12 / 3 (NT 128)
    const unsigned char* bsrc = reinterpret_cast<const unsigned char*>(p.b6);
    const size_t bchunk = (size_t)3 * p.Ncols * 32;              // bytes of one 16-k chunk of B
    auto b_dma = [&](int c, int stage) {
#pragma unroll
        for (int j = 0; j < BPW; ++j) {
            const int q = wv * BPW + j;                          // piece index in the stage: [bf16 piece][NT / 32 KB-pieces]
            const int pc = q / (NT / 32), sub = q % (NT / 32);
            const unsigned char* src = bsrc + (size_t)c * bchunk + ((size_t)pc * p.Ncols + n0) * 32 + sub * 1024 + lane * 16;
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(src), (cx_lds_void*)(smem + stage * STG + ASZ + q * 1024), 16, 0, 0);
        }
    };
    // ---- MFMA operand reads
    unsigned a_rd[2], b_rd[NB];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) { const int r = wm * 64 + mb * 32 + li; a_rd[mb] = (unsigned)(r * 32 + 16 * (lh ^ ((r >> 3) & 1))); }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) { const int r = wn * (NT / 2) + nb * 32 + li; b_rd[nb] = (unsigned)(ASZ + r * 32 + 16 * (lh ^ ((r >> 3) & 1))); }

    f32x16 acc[2][NB];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;

    const int nchunks = p.K / 16;
    f32x4 v0, v1;
    {
        const float* ap = a_ptr(0);
        v0 = *reinterpret_cast<const f32x4*>(ap); v1 = *reinterpret_cast<const f32x4*>(ap + 4);
        b_dma(0, 0);
        cx_i32x4 h, m, l;
        cx_split8(v0, v1, h, m, l);
        *reinterpret_cast<cx_i32x4*>(smem + a_wr) = h;
        *reinterpret_cast<cx_i32x4*>(smem + 128 * 32 + a_wr) = m;
        *reinterpret_cast<cx_i32x4*>(smem + 2 * 128 * 32 + a_wr) = l;
    }
    // the B stage is written by LDS-DMA (global_load ... lds): its completion is tracked by vmcnt, which a workgroup barrier does not wait for
    // by itself -- every wave drains its own DMA pieces before the barrier publishes the stage to the other waves
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int s = c & 1;
        const bool more = c + 1 < nchunks;
        if (more) {
            const float* ap = a_ptr(c + 1);
            v0 = *reinterpret_cast<const f32x4*>(ap); v1 = *reinterpret_cast<const f32x4*>(ap + 4);
            b_dma(c + 1, s ^ 1);
        }
        const unsigned char* st = smem + s * STG;
        cx_bf16x8 af[2][3];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int k = 0; k < 3; ++k) af[mb][k] = *reinterpret_cast<const cx_bf16x8*>(st + k * 128 * 32 + a_rd[mb]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            cx_bf16x8 bf[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) bf[k] = *reinterpret_cast<const cx_bf16x8*>(st + k * NT * 32 + b_rd[nb]);
            // piece pairs (data, weight), smallest products first: l h, h l, m m, m h, h m, h h; the two pixel blocks alternate
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mb][PA[q]], bf[PB[q]], acc[mb][nb], 0, 0, 0);
        }
        if (more) {
            cx_i32x4 h, m, l;
            cx_split8(v0, v1, h, m, l);
            unsigned char* nx = smem + (s ^ 1) * STG;
            *reinterpret_cast<cx_i32x4*>(nx + a_wr) = h;
            *reinterpret_cast<cx_i32x4*>(nx + 128 * 32 + a_wr) = m;
            *reinterpret_cast<cx_i32x4*>(nx + 2 * 128 * 32 + a_wr) = l;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's DMA pieces of the next B stage have landed
        __syncthreads();
    }

    // ---- epilogue.  acc[mb][nb][i]: pixel row = wm*64 + mb*32 + 8*(i/4) + 4*lh + i%4, column = n0 + wn*NT/2 + nb*32 + li
    float ssum[NB], ssq[NB];
    int tap[NB], cho[NB]; float bia[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        ssum[nb] = 0.f; ssq[nb] = 0.f;
        const int n = n0 + wn * (NT / 2) + nb * 32 + li;
        if constexpr (MODE == 0) { tap[nb] = n / p.Cout; cho[nb] = n - tap[nb] * p.Cout; bia[nb] = p.bias ? p.bias[cho[nb]] : 0.f; }
        else { tap[nb] = 0; cho[nb] = n; bia[nb] = 0.f; }
    }
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        // the block's 16 rows of this lane are pixels base + {0..3, 8..11, 16..19, 24..27}; forward: output pixel index of input pixel
        // pp = q W + j is 4 W q + 2 j (+ a 2W + b per tap): one division for the base, the rest by stepping
        const unsigned base = (unsigned)(p0 + wm * 64 + mb * 32 + 4 * lh);
        const unsigned q0 = base / (unsigned)p.W, j0 = base - q0 * (unsigned)p.W;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const unsigned d = 8 * (i >> 2) + (i & 3);
            if constexpr (MODE == 0) {
                unsigned q = q0, j = j0 + d;
                while (j >= (unsigned)p.W) { j -= (unsigned)p.W; ++q; }
                const long obase = (long)4 * p.W * q + 2 * j;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const float y = acc[mb][nb][i] + bia[nb];
                    p.out[(obase + (tap[nb] >> 1) * 2 * p.W + (tap[nb] & 1)) * p.ldo + cho[nb]] = y;
                    if constexpr (STATS) { ssum[nb] += y; ssq[nb] = fmaf(y, y, ssq[nb]); }
                }
            } else {
                // (+ sums, round 6: dx IS the dy of the layer that produced x, so its BatchNorm-backward sums are taken here from the fp32 accumulators
                // and that layer's reduction pass -- a read of dy and r -- drops out; the four loads of a pixel are in flight together)
                float rv[NB];
                if constexpr (STATS) {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) rv[nb] = p.bn_r[(long)(base + d) * p.bn_ldr + cho[nb]];
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const float v = acc[mb][nb][i];
                    p.out[(long)(base + d) * p.ldo + cho[nb]] = v;
                    if constexpr (STATS) { ssum[nb] += v; ssq[nb] = fmaf(v, rv[nb], ssq[nb]); }
                }
            }
        }
    }
    if constexpr (STATS) {
        // per column: the two lane halves, then the two pixel halves of the workgroup; one row of partials per (pixel tile, tap, 64-channel block)
        float* red = reinterpret_cast<float*>(smem);             // [wm 2][NT][2] floats (the stages are dead: the loop ended with a barrier)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const float s2 = ssum[nb] + __shfl_xor(ssum[nb], 32), q2 = ssq[nb] + __shfl_xor(ssq[nb], 32);
            if (lh == 0) { const int col = wn * (NT / 2) + nb * 32 + li; red[(wm * NT + col) * 2] = s2; red[(wm * NT + col) * 2 + 1] = q2; }
        }
        __syncthreads();
        if (tid < NT) {
            const int n = n0 + tid, t = MODE == 0 ? n / p.Cout : 0, co = MODE == 0 ? n - t * p.Cout : n;
            const long rows = (MODE == 0 ? 4 : 1) * (p.P / 128), row = MODE == 0 ? pt * 4 + t : pt;
            float2 o; o.x = red[tid * 2] + red[(NT + tid) * 2]; o.y = red[tid * 2 + 1] + red[(NT + tid) * 2 + 1];
            *reinterpret_cast<float2*>(p.stat_part + (((size_t)(co >> 6) * rows + row) * 64 + (co & 63)) * 2) = o;
        }
    }
}

__global__ __launch_bounds__(256, 2) void convt_x6_fwd_kernel(ConvtX6Args p) { convt_x6_body<0, 4, false>(p); }
__global__ __launch_bounds__(256, 2) void convt_x6_fwd_stats_kernel(ConvtX6Args p) { convt_x6_body<0, 4, true>(p); }
__global__ __launch_bounds__(256, 2) void convt_x6_dgrad_kernel_256(ConvtX6Args p) { convt_x6_body<1, 4, false>(p); }
__global__ __launch_bounds__(256, 2) void convt_x6_dgrad_kernel_128(ConvtX6Args p) { convt_x6_body<1, 2, false>(p); }
__global__ __launch_bounds__(256, 2) void convt_x6_dgrad_bnbwd_kernel_256(ConvtX6Args p) { convt_x6_body<1, 4, true>(p); }
__global__ __launch_bounds__(256, 2) void convt_x6_dgrad_bnbwd_kernel_128(ConvtX6Args p) { convt_x6_body<1, 2, true>(p); }

// Weights W [tap 4][Cout][Cin] (Keras Conv2DTranspose layout, UNet/model.py:41) = Wflat [R = 4 Cout][Cin]  ->  B [K/16][piece 3][Ncols][16 bf16]
//   mode 0 (forward):       B[k = ci][n = tap * Cout + co] = Wflat[n][k]
//   mode 1 (data gradient): B[k = tap * Cout + co][n = ci] = Wflat[k][n]
// item = (16-k chunk, column n, k half): lane order half, then n -> a wave's stores cover contiguous memory
__device__ __forceinline__ void convt_x6_weight_item(const float* __restrict__ w, uint16_t* __restrict__ b6, int Cin, int Cout, int mode, long it) {
    const int R = 4 * Cout;
    const int K = mode ? R : Cin, Nc = mode ? Cin : R;
    if (it >= (long)K * Nc / 8) return;
    const int half = (int)(it & 1), n = (int)((it >> 1) % Nc), c16 = (int)((it >> 1) / Nc);
    const int k0 = c16 * 16 + half * 8;
    f32x4 v0, v1;
    if (mode == 0) { const float* s = w + (size_t)n * Cin + k0; v0 = *reinterpret_cast<const f32x4*>(s); v1 = *reinterpret_cast<const f32x4*>(s + 4); }
    else {
#pragma unroll
        for (int e = 0; e < 4; ++e) { v0[e] = w[(size_t)(k0 + e) * Cin + n]; v1[e] = w[(size_t)(k0 + 4 + e) * Cin + n]; }
    }
    cx_i32x4 h, m, l;
    cx_split8(v0, v1, h, m, l);
    uint16_t* o = b6 + (((size_t)c16 * 3) * Nc + n) * 16 + 8 * (half ^ ((n >> 3) & 1));
    *reinterpret_cast<cx_i32x4*>(o) = h;
    *reinterpret_cast<cx_i32x4*>(o + (size_t)Nc * 16) = m;
    *reinterpret_cast<cx_i32x4*>(o + (size_t)2 * Nc * 16) = l;
}

// jobs[j] = { w, W6, Cin | Cout << 32, first block, mode, 0 }: both directions of every transposed conv in ONE launch
__global__ __launch_bounds__(256) void convt_x6_weight_batch_kernel(const long long* __restrict__ jobs, int njobs) {
    int j = 0;
    while (j + 1 < njobs && (int)jobs[(j + 1) * 6 + 3] <= (int)blockIdx.x) ++j;
    convt_x6_weight_item(reinterpret_cast<const float*>(jobs[j * 6 + 0]), reinterpret_cast<uint16_t*>(jobs[j * 6 + 1]),
                         (int)(jobs[j * 6 + 2] & 0xffffffffll), (int)(jobs[j * 6 + 2] >> 32), (int)jobs[j * 6 + 4],
                         ((long)blockIdx.x - (int)jobs[j * 6 + 3]) * 256 + threadIdx.x);
}

__global__ __launch_bounds__(256) void convt_x6_weight_kernel(const float* __restrict__ w, uint16_t* __restrict__ b6, int Cin, int Cout, int mode) {
    convt_x6_weight_item(w, b6, Cin, Cout, mode, (long)blockIdx.x * 256 + threadIdx.x);
}


// ---- weight gradient ------------------------------------------------------------------------------------------------------------------
//   dW[a,b,co,ci] = sum_{n,i,j} dz[n,2i+a,2j+b,co] * x[n,i,j,ci]:   per tap a GEMM  M = co, N = ci, K = input pixels, in BF16x6.
// Both operands are activations: both are split into three bf16 pieces in registers and staged in their NATURAL [pixel][channel] order;
// the MFMA operands (8 consecutive pixels of one channel per lane) come out of ds_read_b64_tr_b16, gfx950's transposing LDS read
// (a 16-lane group reads a 4-pixel x 16-channel block and each lane receives one channel's four pixels).  Row pitches carry 64 bytes
// of padding (x: 128 channels -> 320 B, dz: 64 channels -> 192 B) so that the four pixel rows of a block start 16 banks apart.
// Workgroup = 64 co x 128 ci x 4 taps, wave = one tap (2 x 4 blocks of 32 x 32 = 128 accumulator registers), persistent over a contiguous
// range of 16-pixel chunks (split-K: partials [split][tap][co][ci] -> fixed-order reduction).  One 51 KB LDS stage; the next chunk's
// rows are loaded into registers before the MFMAs of this one and written behind them; two workgroups per CU overlap each other's barriers.
typedef short cx_s16x4 __attribute__((ext_vector_type(4)));
typedef short cx_s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) cx_s16x4 cx_lds_s16x4;

struct ConvtWgX6Args {
    const float* x; const float* dz; float* out;   // out: dw (splits == 1) or the split-K workspace
    int ldx, lddz, N, H, W, Cin, Cout;
    int nco, nci, splits;
    long P, chunks;                                  // input pixels, 16-pixel chunks (P / 16)
};

constexpr int kWgXP = 320, kWgZP = 192;                          // row pitches of the x (128 channels) and dz (64 channels) images, bytes
constexpr int kWgXImg = 16 * kWgXP, kWgZImg = 16 * kWgZP;        // one piece of one chunk
constexpr int kWgStage = 3 * kWgXImg + 12 * kWgZImg;             // 15360 + 36864 = 52224 B

__device__ __forceinline__ void cx_split4(const f32x4& v, cx_u32x2& h, cx_u32x2& m, cx_u32x2& l) {
    unsigned h0, m0, l0, h1, m1, l1;
    cx_split2(v[0], v[1], h0, m0, l0); cx_split2(v[2], v[3], h1, m1, l1);
    h = cx_u32x2{h0, h1}; m = cx_u32x2{m0, m1}; l = cx_u32x2{l0, l1};
}

__global__ __launch_bounds__(256, 2) void convt_x6_wgrad_kernel(ConvtWgX6Args p) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[kWgStage];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);     // = tap (a, b) = (wv >> 1, wv & 1)
    const int tile = blockIdx.x % (p.nco * p.nci), split = blockIdx.x / (p.nco * p.nci);
    const int co0 = (tile % p.nco) * 64, ci0 = (tile / p.nco) * 128;
    const long c_lo = p.chunks * split / p.splits, c_hi = p.chunks * (split + 1) / p.splits;

    // ---- staging roles: x rows r and r + 8, channel quad xc (two float4 per chunk); dz row zr, channel quad zc, all four taps
    const int xr = tid >> 5, xc = tid & 31, zr = tid >> 4, zc = tid & 15;
    f32x4 vx[2], vz[4];
    auto load = [&](long c) {
        const unsigned pp = (unsigned)(c * 16);
#pragma unroll
        for (int u = 0; u < 2; ++u) vx[u] = *reinterpret_cast<const f32x4*>(p.x + (long)(pp + xr + 8 * u) * p.ldx + ci0 + 4 * xc);
        const unsigned pz = pp + zr;
        const unsigned q = pz / (unsigned)p.W, j = pz - q * (unsigned)p.W;          // input pixel = q W + j  ->  output pixel 4 W q + 2 j + a 2W + b
        const float* zb = p.dz + ((long)4 * p.W * q + 2 * j) * p.lddz + co0 + 4 * zc;
#pragma unroll
        for (int t = 0; t < 4; ++t) vz[t] = *reinterpret_cast<const f32x4*>(zb + (long)((t >> 1) * 2 * p.W + (t & 1)) * p.lddz);
    };
    auto store = [&]() {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            cx_u32x2 h, m, l; cx_split4(vx[u], h, m, l);
            unsigned char* d = smem + (xr + 8 * u) * kWgXP + xc * 8;
            *reinterpret_cast<cx_u32x2*>(d) = h; *reinterpret_cast<cx_u32x2*>(d + kWgXImg) = m; *reinterpret_cast<cx_u32x2*>(d + 2 * kWgXImg) = l;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            cx_u32x2 h, m, l; cx_split4(vz[t], h, m, l);
            unsigned char* d = smem + 3 * kWgXImg + (t * 3) * kWgZImg + zr * kWgZP + zc * 8;
            *reinterpret_cast<cx_u32x2*>(d) = h; *reinterpret_cast<cx_u32x2*>(d + kWgZImg) = m; *reinterpret_cast<cx_u32x2*>(d + 2 * kWgZImg) = l;
        }
    };
    // ---- transposing operand reads: lane = 16 g + 4 q + pq supplies the address of pixel row 8 (g >> 1) + q, channels 16 (g & 1) + 4 pq .. + 3
    const int g = lane >> 4, q4 = (lane >> 2) & 3, pq = lane & 3;
    const unsigned x_rd = (unsigned)((8 * (g >> 1) + q4) * kWgXP + (16 * (g & 1) + 4 * pq) * 2);
    const unsigned z_rd = (unsigned)(3 * kWgXImg + (wv * 3) * kWgZImg + (8 * (g >> 1) + q4) * kWgZP + (16 * (g & 1) + 4 * pq) * 2);

    f32x16 acc[2][4];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;

    if (c_lo < c_hi) load(c_lo);
    for (long c = c_lo; c < c_hi; ++c) {
        store();
        __syncthreads();
        if (c + 1 < c_hi) load(c + 1);
        cx_bf16x8 af[2][3];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const cx_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((cx_lds_s16x4*)(smem + z_rd + k * kWgZImg + mb * 64));
                const cx_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((cx_lds_s16x4*)(smem + z_rd + k * kWgZImg + mb * 64 + 4 * kWgZP));
                af[mb][k] = __builtin_bit_cast(cx_bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            cx_bf16x8 bf[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const cx_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((cx_lds_s16x4*)(smem + x_rd + k * kWgXImg + nb * 64));
                const cx_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((cx_lds_s16x4*)(smem + x_rd + k * kWgXImg + nb * 64 + 4 * kWgXP));
                bf[k] = __builtin_bit_cast(cx_bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int qq = 0; qq < 6; ++qq)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mb][PA[qq]], bf[PB[qq]], acc[mb][nb], 0, 0, 0);
        }
        __syncthreads();
    }
    // ---- partial result: out[split][tap][co][ci]; acc[mb][nb][i]: co = co0 + mb*32 + 8*(i/4) + 4*(lane/32) + i%4, ci = ci0 + nb*32 + lane%32
    float* o = p.out + ((size_t)split * 4 + wv) * p.Cout * p.Cin;
    const int li = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int co = co0 + mb * 32 + 8 * (i >> 2) + 4 * lh + (i & 3);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) o[(size_t)co * p.Cin + ci0 + nb * 32 + li] = acc[mb][nb][i];
        }
}

// dw[i] = sum over the splits in a fixed order: SL slices of the split range per output (float4), summed slice by slice -- few outputs x
// many splits (up_1: 8192 float4 x 512 splits) still fill the chip (the scheme of conv_wgrad.hip's wgrad_reduce_kernel)
__global__ __launch_bounds__(256) void convt_x6_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, long n4, int splits, int sl) {
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

int convt_x6_wgrad_splits(int N, int H, int W, int Cin, int Cout, int max_workgroups) {
    const long chunks = (long)N * H * W / 16;
    const int tiles = (Cout / 64) * (Cin / 128);
    long s = (2 * unet_grid_slots(256, max_workgroups) + tiles - 1) / tiles;     // two workgroups per CU (or per slot of the caller's cap)
    if (s > chunks / 8) s = chunks / 8;                          // at least 8 chunks (128 pixels) per workgroup
    if (s < 1) s = 1;
    if (s > 1024) s = 1024;
    return (int)s;
}

bool convt_x6_shape_ok(int N, int H, int W, int Cin, int Cout) {
    const long P = (long)N * H * W;
    return N > 0 && H > 0 && W > 0 && Cin % 128 == 0 && Cout % 64 == 0 && P % 128 == 0 && Cin <= 4096 && Cout <= 4096 &&
           4 * P < ((long)1 << 31);
}

}  // namespace

// 1 when the BF16x6 transposed-conv kernels take the layer (N, H, W: INPUT dims): N*H*W a multiple of the 128-pixel tile, Cin % 128 == 0
// (the data gradient's column tile), Cout % 64 == 0
extern "C" int unet_convT2x2_x6_supported(int N, int H, int W, int Cin, int Cout) { return convt_x6_shape_ok(N, H, W, Cin, Cout) ? 1 : 0; }

// bytes of one direction's weight operand (three bf16 pieces per weight)
extern "C" size_t unet_convT2x2_x6_weight_bytes(int Cin, int Cout) { return (size_t)4 * Cin * Cout * 3 * sizeof(uint16_t); }

// mode 0: forward operand, mode 1: data-gradient operand; w = the layer's fp32 kernel [2][2][Cout][Cin]
extern "C" int unet_convT2x2_weight_transform_x6(const float* w, void* W6, int Cin, int Cout, int mode, void* stream) {
    UNET_CHECK_ARG(w && W6 && Cin > 0 && Cout > 0 && Cin % 16 == 0 && Cout % 16 == 0 && (mode == 0 || mode == 1) && unet_aligned16(w) && unet_aligned16(W6));
    const long items = (long)4 * Cin * Cout / 8;
    convt_x6_weight_kernel<<<(unsigned)((items + 255) / 256), 256, 0, (hipStream_t)stream>>>(w, (uint16_t*)W6, Cin, Cout, mode);
    return UNET_LAUNCH_STATUS();
}

// jobs: device array of njobs x 6 int64 = { w, W6, Cin | Cout << 32, first_block, mode, 0 }, first_block = running sum of ceil(4*Cin*Cout/8 / 256)
extern "C" int unet_convT2x2_weight_transform_x6_batch(const void* jobs, int njobs, int total_blocks, void* stream) {
    UNET_CHECK_ARG(jobs && njobs > 0 && total_blocks > 0);
    convt_x6_weight_batch_kernel<<<dim3((unsigned)total_blocks), 256, 0, (hipStream_t)stream>>>((const long long*)jobs, njobs);
    return UNET_LAUNCH_STATUS();
}

// rows of statistics partials per 64-channel block the forward kernel writes: one per (128-pixel tile, tap)
extern "C" int unet_convT2x2_x6_stats_rows(int N, int H, int W, int Cin, int Cout) {
    return convt_x6_shape_ok(N, H, W, Cin, Cout) ? (int)(4 * ((long)N * H * W / 128)) : 0;
}

// Forward: the arguments of unet_convT2x2_fwd_stream_stats with W6 (mode 0) in place of w; stat_part nullable ((Cout/64) * rows * 128 floats,
// rows = unet_convT2x2_x6_stats_rows; finish with unet_bn_train_finalize_partials)
extern "C" int unet_convT2x2_fwd_x6(const float* x, int ldx, const void* W6, const float* bias, float* out, int ldo,
                                    int N, int H, int W, int Cin, int Cout, float* stat_part, size_t stat_bytes, void* stream) {
    UNET_CHECK_ARG(x && W6 && out && convt_x6_shape_ok(N, H, W, Cin, Cout));
    UNET_CHECK_ARG(ldx >= Cin && ldo >= Cout && ldx % 4 == 0 && unet_aligned16(x) && unet_aligned16(W6) && (!stat_part || unet_aligned16(stat_part)));
    ConvtX6Args a{};
    a.a = x; a.b6 = (const uint16_t*)W6; a.bias = bias; a.out = out; a.stat_part = stat_part; a.lda = ldx; a.ldo = ldo;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.P = (long)N * H * W;
    a.K = Cin; a.Ncols = 4 * Cout; a.nct = a.Ncols / 256;
    const long blocks = (a.P / 128) * a.nct;
    if (blocks <= 0 || blocks > 0x7fffffffL) return UNET_EINVAL;
    if (stat_part) {
        if (stat_bytes < (size_t)(Cout / 64) * unet_convT2x2_x6_stats_rows(N, H, W, Cin, Cout) * 128 * sizeof(float)) return UNET_ENOSPC;
        convt_x6_fwd_stats_kernel<<<dim3((unsigned)blocks), 256, 0, (hipStream_t)stream>>>(a);
    } else convt_x6_fwd_kernel<<<dim3((unsigned)blocks), 256, 0, (hipStream_t)stream>>>(a);
    return UNET_LAUNCH_STATUS();
}

// rows of BatchNorm-backward partials per 64-channel block the data gradient with sums writes: one per 128-pixel tile
extern "C" int unet_convT2x2_x6_bnbwd_rows(int N, int H, int W, int Cin, int Cout) {
#ifdef UNET_CTX6_NO_SUMS        /* diagnostic build: the plan then keeps the producer's reduction pass (A/B of the fusion) */
    return 0;
#endif
    return convt_x6_shape_ok(N, H, W, Cin, Cout) ? (int)((long)N * H * W / 128) : 0;
}

// Data gradient: the arguments of unet_convT2x2_dgrad with W6d (mode 1) in place of w.  r_prev / stat_part (both or neither): dx is the
// gradient of the BatchNorm output of the layer that produced x, r_prev [N*H*W][ldr] that layer's saved activation -- the kernel also leaves its
// BatchNorm-backward sums (sum dx, sum dx * r_prev) as (Cin/64) * rows * 128 floats, rows = unet_convT2x2_x6_bnbwd_rows
// (the form unet_bn_bwd_any takes as `sums_part`)
extern "C" int unet_convT2x2_dgrad_x6_sums(const float* dz, int lddz, const void* W6d, float* dx, int lddx,
                                           int N, int H, int W, int Cin, int Cout, const float* r_prev, int ldr,
                                           float* stat_part, size_t stat_bytes, void* stream) {
    UNET_CHECK_ARG(dz && W6d && dx && convt_x6_shape_ok(N, H, W, Cin, Cout));
    UNET_CHECK_ARG(lddz >= Cout && lddx >= Cin && lddz % 4 == 0 && unet_aligned16(dz) && unet_aligned16(W6d));
    UNET_CHECK_ARG((r_prev != nullptr) == (stat_part != nullptr) && (!r_prev || (ldr >= Cin && unet_aligned16(stat_part))));
    ConvtX6Args a{};
    a.a = dz; a.b6 = (const uint16_t*)W6d; a.out = dx; a.lda = lddz; a.ldo = lddx;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.P = (long)N * H * W;
    a.K = 4 * Cout; a.Ncols = Cin;
    a.bn_r = r_prev; a.bn_ldr = ldr; a.stat_part = stat_part;
    const bool wide = Cin % 256 == 0;
    a.nct = Cin / (wide ? 256 : 128);
    const long blocks = (a.P / 128) * a.nct;
    if (blocks <= 0 || blocks > 0x7fffffffL) return UNET_EINVAL;
    if (stat_part) {
        if (stat_bytes < (size_t)(Cin / 64) * unet_convT2x2_x6_bnbwd_rows(N, H, W, Cin, Cout) * 128 * sizeof(float)) return UNET_ENOSPC;
        if (wide) convt_x6_dgrad_bnbwd_kernel_256<<<dim3((unsigned)blocks), 256, 0, (hipStream_t)stream>>>(a);
        else      convt_x6_dgrad_bnbwd_kernel_128<<<dim3((unsigned)blocks), 256, 0, (hipStream_t)stream>>>(a);
    } else {
        if (wide) convt_x6_dgrad_kernel_256<<<dim3((unsigned)blocks), 256, 0, (hipStream_t)stream>>>(a);
        else      convt_x6_dgrad_kernel_128<<<dim3((unsigned)blocks), 256, 0, (hipStream_t)stream>>>(a);
    }
    return UNET_LAUNCH_STATUS();
}
extern "C" int unet_convT2x2_dgrad_x6(const float* dz, int lddz, const void* W6d, float* dx, int lddx,
                                      int N, int H, int W, int Cin, int Cout, void* stream) {
    return unet_convT2x2_dgrad_x6_sums(dz, lddz, W6d, dx, lddx, N, H, W, Cin, Cout, nullptr, 0, nullptr, 0, stream);
}

// Weight gradient dw [2][2][Cout][Cin] (H, W: INPUT dims); supported as the forward (N*H*W % 128 == 0, Cin % 128 == 0, Cout % 64 == 0).
// ws: unet_convT2x2_wgrad_x6_workspace bytes (split-K partials; untouched when the layer needs no split)
// max_workgroups: cap on the one-wave grid (common.h unet_grid_slots; the kernel runs two workgroups per slot), same value for both calls
extern "C" size_t unet_convT2x2_wgrad_x6_workspace_wg(int N, int H, int W, int Cin, int Cout, int max_workgroups) {
    if (!convt_x6_shape_ok(N, H, W, Cin, Cout)) return 0;
    const int s = convt_x6_wgrad_splits(N, H, W, Cin, Cout, max_workgroups);
    return s > 1 ? (size_t)s * 4 * Cin * Cout * sizeof(float) : 16;
}
extern "C" size_t unet_convT2x2_wgrad_x6_workspace(int N, int H, int W, int Cin, int Cout) {
    return unet_convT2x2_wgrad_x6_workspace_wg(N, H, W, Cin, Cout, 0);
}

extern "C" int unet_convT2x2_wgrad_x6_wg(const float* xin, int ldx, const float* dz, int lddz, float* dw,
                                         int N, int H, int W, int Cin, int Cout, int max_workgroups, void* ws, size_t ws_bytes, void* stream) {
    UNET_CHECK_ARG(xin && dz && dw && ws && convt_x6_shape_ok(N, H, W, Cin, Cout));
    UNET_CHECK_ARG(ldx >= Cin && lddz >= Cout && ldx % 4 == 0 && lddz % 4 == 0 && unet_aligned16(xin) && unet_aligned16(dz) && unet_aligned16(dw) && unet_aligned16(ws));
    if (ws_bytes < unet_convT2x2_wgrad_x6_workspace_wg(N, H, W, Cin, Cout, max_workgroups)) return UNET_ENOSPC;
    ConvtWgX6Args a{};
    a.x = xin; a.dz = dz; a.ldx = ldx; a.lddz = lddz; a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.nco = Cout / 64; a.nci = Cin / 128; a.P = (long)N * H * W; a.chunks = a.P / 16;
    a.splits = convt_x6_wgrad_splits(N, H, W, Cin, Cout, max_workgroups);
    a.out = a.splits > 1 ? (float*)ws : dw;
    hipStream_t st = (hipStream_t)stream;
    convt_x6_wgrad_kernel<<<dim3((unsigned)(a.nco * a.nci * a.splits)), 256, 0, st>>>(a);
    int rc = UNET_LAUNCH_STATUS(); if (rc) return rc;
    if (a.splits > 1) {
        const long n4 = (long)4 * Cin * Cout / 4;
        int sl = 1;
        while (sl < 16 && 2 * sl <= a.splits && n4 * sl < 256 * 1024) sl *= 2;
        convt_x6_wgrad_reduce_kernel<<<(unsigned)((n4 * sl + 255) / 256), 256, 0, st>>>((const float*)ws, dw, n4, a.splits, sl);
        rc = UNET_LAUNCH_STATUS();
    }
    return rc;
}
extern "C" int unet_convT2x2_wgrad_x6(const float* xin, int ldx, const float* dz, int lddz, float* dw,
                                      int N, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream) {
    return unet_convT2x2_wgrad_x6_wg(xin, ldx, dz, lddz, dw, N, H, W, Cin, Cout, 0, ws, ws_bytes, stream);
}
