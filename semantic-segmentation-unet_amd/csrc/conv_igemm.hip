// Implicit-GEMM convolution on the gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
//   out[pixel][n] = sum_{tap} sum_{k} in[pixel (+) tap][k] * Wt[tap][k][n]      (+ bias, ReLU)
//
// One kernel template serves the four dense "activation-producing" contractions of the U-Net
// (reference layer recipes: UNet/model.py:28-48):
//   MODE 0  3x3 'same' conv forward (9 taps over an LDS halo tile) and, with flipped taps and the
//           [tap][n][k] weight view, its data gradient (dgrad);
//   MODE 1  one tap, output scattered with stride 2: the 2x2/stride-2 transposed conv forward
//           (blockIdx.z = tap (a,b), out pixel (2i+a, 2j+b));
//   MODE 2  four taps gathered with stride 2 as a K-extension: the transposed conv's data gradient.
//
// Tiling (wave64, MFMA 32x32x2 f32): workgroup = 4 waves; a wave owns 64 pixels x 64 channels
// (2x2 MFMA tiles, 64 accumulator VGPRs); the MFMA M dimension is 32 consecutive pixels of one image row.
// K is consumed in chunks of CK=32 channels staged through LDS:
//   A tile  [pixels][CK+4]  (pixel stride 36 floats: a lane's 4 consecutive k are one ds_read_b128,
//            16-lane groups land on 16 distinct 16-B slots -> conflict-free),
//   B tile  [CK][BN] (weights stored [k][n]: four ds_read_b32 per fragment) or [BN][CK+4] (stored [n][k]).
// The k order inside a chunk is permuted (lane half h takes k = 8*kk + 4*h + s); A and B use the same
// permutation so the sum is unchanged.  Weights for iteration i+1 are prefetched into registers while
// iteration i's MFMAs run (global -> VGPR -> LDS, write after the barrier).
#include "common.h"

namespace {

constexpr int CK = 32;
constexpr int SA = CK + 4;
constexpr int TW = 32;
#ifndef UNET_ABLATE
#define UNET_ABLATE 0
#endif

struct IgemmArgs {
    const float* x; const float* w; const float* bias; float* out;
    int ldx, ldo;
    int N, H, W;            // tile-grid image dims (conv output dims; convT fwd: input dims; convT dgrad: dx dims)
    int Hi, Wi;             // dims of the image `x` is read from
    int Ho, Wo;             // dims of the image `out` is written to
    int Kdim, Ndim;         // GEMM K per tap, GEMM N
    int in_scale;           // MODE 1/2: input pixel = grid pixel * in_scale + tap offset
    int out_scale;          // output pixel = grid pixel * out_scale + (tap offset when tap_by_z)
    int relu, flip, tap_by_z;
    long zs_x, zs_out;      // MODE 3: element strides between the blockIdx.z planes of x / out
    int tiles_y, tiles_x, tiles_n;
};

template <int MODE, int WM, int WN, bool B_NK>
__global__ __launch_bounds__(256, 2) void igemm_kernel(IgemmArgs p) {
    constexpr int TH = 2 * WM;
    constexpr int BN = 64 * WN;
    constexpr int NT = MODE == 0 ? 9 : (MODE == 2 ? 4 : 1);
    constexpr int A_ROWS = MODE == 0 ? TH + 2 : TH;
    constexpr int A_COLS = MODE == 0 ? TW + 2 : TW;
    constexpr int A_FLOATS = A_ROWS * A_COLS * SA;
    constexpr int B_FLOATS = B_NK ? BN * SA : CK * BN;
    constexpr int B_F4 = CK * BN / 4 / 256;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    __shared__ __attribute__((aligned(16))) float smem[A_FLOATS + B_FLOATS];
    float* sA = smem;
    float* sB = smem + A_FLOATS;

    int b = blockIdx.x;
    const int tn = b % p.tiles_n; b /= p.tiles_n;
    const int tx = b % p.tiles_x; b /= p.tiles_x;
    const int ty = b % p.tiles_y;
    const int img = b / p.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW, n0 = tn * BN;
    const int ztap = p.tap_by_z ? (int)blockIdx.z : 0;
    const float* xin = MODE == 3 ? p.x + (size_t)blockIdx.z * p.zs_x : p.x;
    float* xout = MODE == 3 ? p.out + (size_t)blockIdx.z * p.zs_out : p.out;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv / WN, wn = wv % WN;
    const int li = lane & 31, lh = lane >> 5;

    f32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

    const int n_chunks = p.Kdim / CK;
    const int n_it = n_chunks * NT;
    f32x4 rb[B_F4];

    auto gload_B = [&](int chunk, int tap) {
        int tapw = tap;
        if (MODE == 0 && p.flip) tapw = 8 - tap;
        if (p.tap_by_z) tapw = ztap;
        const int c0 = chunk * CK;
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const int idx = tid + i * 256;
            if (!B_NK) {
                const int k = idx / (BN / 4), q = idx % (BN / 4);
                rb[i] = *reinterpret_cast<const f32x4*>(p.w + ((size_t)(tapw * p.Kdim + c0 + k) * p.Ndim + n0 + 4 * q));
            } else {
                const int nn = idx >> 3, q = idx & 7;
                rb[i] = *reinterpret_cast<const f32x4*>(p.w + ((size_t)(tapw * p.Ndim + n0 + nn) * p.Kdim + c0 + 4 * q));
            }
        }
    };
    auto store_B = [&]() {
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const int idx = tid + i * 256;
            if (!B_NK) {
                const int k = idx / (BN / 4), q = idx % (BN / 4);
                *reinterpret_cast<f32x4*>(sB + k * BN + 4 * q) = rb[i];
            } else {
                const int nn = idx >> 3, q = idx & 7;
                *reinterpret_cast<f32x4*>(sB + nn * SA + 4 * q) = rb[i];
            }
        }
    };
    // A tile: global -> registers (gload_A) -> LDS (store_A), so the next tile's loads fly under the current MFMAs.
    constexpr int A_TOTAL = A_ROWS * A_COLS * (CK / 4);
    constexpr int A_F4 = (A_TOTAL + 255) / 256;
    f32x4 ra[A_F4];
    auto gload_A = [&](int chunk, int tap) {
        const int c0 = chunk * CK;
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int idx = tid + i * 256;
            const int pix = idx >> 3, q = idx & 7;
            const int iy = pix / A_COLS, ix = pix - iy * A_COLS;
            int gy, gx;
            bool ok = idx < A_TOTAL;
            if (MODE == 0) {
                gy = oy0 + iy - 1; gx = ox0 + ix - 1;
                ok = ok && (gy >= 0) && (gy < p.Hi) && (gx >= 0) && (gx < p.Wi);
            } else {
                const int oy = oy0 + iy, ox = ox0 + ix;
                ok = ok && (oy < p.H) && (ox < p.W);
                gy = oy * p.in_scale + (MODE == 2 ? (tap >> 1) : 0);
                gx = ox * p.in_scale + (MODE == 2 ? (tap & 1) : 0);
            }
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *reinterpret_cast<const f32x4*>(xin + ((size_t)(img * p.Hi + gy) * p.Wi + gx) * p.ldx + c0 + 4 * q);
            ra[i] = v;
        }
    };
    auto store_A = [&]() {
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int idx = tid + i * 256;
            if (idx < A_TOTAL) *reinterpret_cast<f32x4*>(sA + (idx >> 3) * SA + 4 * (idx & 7)) = ra[i];
        }
    };

    // per-lane fragment bases (floats)
    int a_base[2], b_base[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) a_base[t] = ((2 * wm + t) * A_COLS + li) * SA + 4 * lh;
#pragma unroll
    for (int u = 0; u < 2; ++u)
        b_base[u] = B_NK ? (wn * 64 + u * 32 + li) * SA + 4 * lh : (4 * lh) * BN + wn * 64 + u * 32 + li;

    // bias is fetched (and waited for) before the main loop, see the epilogue note
    float bias_v[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        bias_v[u] = p.bias ? p.bias[n0 + wn * 64 + u * 32 + li] : 0.f;
        asm volatile("" :: "v"(bias_v[u]));
    }

    constexpr bool PREFETCH_A = (MODE != 0) || (WM == 4);
    gload_A(0, 0);
    gload_B(0, 0);
    int chunk = 0, tap = 0;
    for (int it = 0; it < n_it; ++it) {
#if UNET_ABLATE >= 1        /* diagnostic builds only (scripts/ablate_igemm.sh): results are wrong by construction */
        if (it == 0) { store_A(); store_B(); __syncthreads(); }
#if UNET_ABLATE < 2
        __syncthreads(); __syncthreads();
#endif
#else
        __syncthreads();
        if (MODE != 0 || tap == 0) {
            if (!PREFETCH_A && it > 0) gload_A(chunk, tap);
            store_A();
        }
        store_B();
        __syncthreads();
#endif
        int nchunk = chunk, ntap = tap + 1;
        if (ntap == NT) { ntap = 0; nchunk = chunk + 1; }
#if UNET_ABLATE < 1
        if (it + 1 < n_it) {
            gload_B(nchunk, ntap);
            if (PREFETCH_A && (MODE != 0 || ntap == 0)) gload_A(nchunk, ntap);     // MODE 0: issued during the chunk's last tap
        }
#endif

        const int tap_off = MODE == 0 ? ((tap / 3) * A_COLS + (tap % 3)) * SA : 0;
#pragma unroll
        for (int kk = 0; kk < CK / 8; ++kk) {
            f32x4 af[2], bf[2];
#if UNET_ABLATE >= 3
            af[0] = f32x4{1.f, 2.f, 3.f, 4.f} * (float)(it + kk); af[1] = af[0] + 1.f; bf[0] = af[0] - 1.f; bf[1] = af[1] * 0.5f;
            asm volatile("" : "+v"(af[0]), "+v"(af[1]), "+v"(bf[0]), "+v"(bf[1]));
#else
#pragma unroll
            for (int t = 0; t < 2; ++t) af[t] = *reinterpret_cast<const f32x4*>(sA + a_base[t] + tap_off + kk * 8);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (B_NK) {
                    bf[u] = *reinterpret_cast<const f32x4*>(sB + b_base[u] + kk * 8);
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) bf[u][s] = sB[b_base[u] + (kk * 8 + s) * BN];
                }
            }
#endif
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[t][s], bf[u][s], acc[t][u], 0, 0, 0);
        }
        chunk = nchunk; tap = ntap;
    }

    // epilogue: C/D layout col = lane&31 (channel), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (pixel along x).
    // One 64-bit base per (row, channel tile); per-element offsets are 32-bit.  No load may be pending here:
    // stores count in vmcnt, so a late `s_waitcnt vmcnt(0)` for the bias would serialise all 64 stores.
    const int ooy = (p.tap_by_z && MODE != 3) ? (ztap >> 1) : 0, oox = (p.tap_by_z && MODE != 3) ? (ztap & 1) : 0;
    const bool full_x = (ox0 + TW <= p.W);
    const int xstep = p.out_scale * p.ldo;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int oy = oy0 + 2 * wm + t;
        if (oy >= p.H) continue;
        const int py = oy * p.out_scale + ooy;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float* base = xout + ((size_t)(img * p.Ho + py) * p.Wo + (ox0 * p.out_scale + oox)) * p.ldo + n0 + wn * 64 + u * 32 + li;
            const float bv = bias_v[u];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dx = (r & 3) + 8 * (r >> 2) + 4 * lh;
                float v = acc[t][u][r] + bv;
                if (p.relu) v = fmaxf(v, 0.f);
                if (full_x || ox0 + dx < p.W) base[dx * xstep] = v;
            }
        }
    }
}

template <int MODE, bool B_NK>
int launch_igemm(IgemmArgs a, int zdim, hipStream_t st) {
    // 128-wide channel tile when N allows it, else 64 wide x 8 rows.
    const bool wide = (a.Ndim % 128) == 0;
    const int TH = wide ? 4 : 8, BN = wide ? 128 : 64;
    a.tiles_y = unet_cdiv(a.H, TH);
    a.tiles_x = unet_cdiv(a.W, TW);
    a.tiles_n = a.Ndim / BN;
    const long blocks = (long)a.N * a.tiles_y * a.tiles_x * a.tiles_n;
    if (blocks <= 0 || blocks > 0x7fffffffL) return UNET_EINVAL;
    dim3 grid((unsigned)blocks, 1, (unsigned)zdim);
    if (wide) igemm_kernel<MODE, 2, 2, B_NK><<<grid, 256, 0, st>>>(a);
    else      igemm_kernel<MODE, 4, 1, B_NK><<<grid, 256, 0, st>>>(a);
    return UNET_LAUNCH_STATUS();
}

}  // namespace

// Batched pointwise GEMM over `planes` independent [T x K] x [K x N] products (the 16 Winograd points):
// out[z][t][n] = sum_k x[z][t][k] * w[z][k][n]; T must be a multiple of 32.  Used by winograd.hip.
int unet_igemm_batched_planes(const float* x, const float* w, float* out, long T, int K, int N, int planes, hipStream_t st) {
    if (!x || !w || !out || T <= 0 || T % 32 || K % CK || N % 64 || planes <= 0) return UNET_EINVAL;
    IgemmArgs a{};
    a.x = x; a.w = w; a.bias = nullptr; a.out = out; a.ldx = K; a.ldo = N;
    a.N = 1; a.H = (int)(T / 32); a.W = 32; a.Hi = a.H; a.Wi = 32; a.Ho = a.H; a.Wo = 32;
    a.Kdim = K; a.Ndim = N; a.in_scale = 1; a.out_scale = 1; a.relu = 0; a.flip = 0; a.tap_by_z = 1;
    a.zs_x = T * K; a.zs_out = T * N;
    return launch_igemm<3, false>(a, planes, st);
}

namespace {

bool igemm_shape_ok(int Kdim, int Ndim) { return Kdim > 0 && Ndim > 0 && (Kdim % CK) == 0 && (Ndim % 64) == 0; }

}  // namespace

// ---- C ABI (declared in include/unet_hip.h) -----------------------------------------------------------------
extern "C" int unet_conv3x3_mfma_supported(int Cin, int Cout) { return igemm_shape_ok(Cin, Cout) ? 1 : 0; }

extern "C" int unet_conv3x3_fwd_mfma(const float* x, int ldx, const float* w, const float* bias, float* out, int ldo,
                                     int N, int H, int W, int Cin, int Cout, int relu, void* stream) {
    UNET_CHECK_ARG(x && w && out && N > 0 && H > 0 && W > 0);
    UNET_CHECK_ARG(igemm_shape_ok(Cin, Cout) && ldx >= Cin && ldo >= Cout && (ldx % 4) == 0);
    UNET_CHECK_ARG(unet_aligned16(x) && unet_aligned16(w));
    IgemmArgs a{};
    a.x = x; a.w = w; a.bias = bias; a.out = out; a.ldx = ldx; a.ldo = ldo;
    a.N = N; a.H = H; a.W = W; a.Hi = H; a.Wi = W; a.Ho = H; a.Wo = W;
    a.Kdim = Cin; a.Ndim = Cout; a.in_scale = 1; a.out_scale = 1; a.relu = relu; a.flip = 0; a.tap_by_z = 0;
    return launch_igemm<0, false>(a, 1, (hipStream_t)stream);
}

// dx[N,H,W,Cin] = sum_{a,b,co} dz[n, y-(a-1), x-(b-1), co] * w[a,b,ci,co]   (w is the forward HWIO kernel)
extern "C" int unet_conv3x3_dgrad_mfma(const float* dz, int lddz, const float* w, float* dx, int lddx,
                                       int N, int H, int W, int Cin, int Cout, void* stream) {
    UNET_CHECK_ARG(dz && w && dx && N > 0 && H > 0 && W > 0);
    UNET_CHECK_ARG(igemm_shape_ok(Cout, Cin) && lddz >= Cout && lddx >= Cin && (lddz % 4) == 0);
    UNET_CHECK_ARG(unet_aligned16(dz) && unet_aligned16(w));
    IgemmArgs a{};
    a.x = dz; a.w = w; a.bias = nullptr; a.out = dx; a.ldx = lddz; a.ldo = lddx;
    a.N = N; a.H = H; a.W = W; a.Hi = H; a.Wi = W; a.Ho = H; a.Wo = W;
    a.Kdim = Cout; a.Ndim = Cin; a.in_scale = 1; a.out_scale = 1; a.relu = 0; a.flip = 1; a.tap_by_z = 0;
    return launch_igemm<0, true>(a, 1, (hipStream_t)stream);
}

// out[n, 2i+a, 2j+b, co] = bias[co] + sum_ci x[n,i,j,ci] * w[a,b,co,ci]      (Keras Conv2DTranspose kernel layout)
extern "C" int unet_convT2x2_fwd(const float* x, int ldx, const float* w, const float* bias, float* out, int ldo,
                                 int N, int H, int W, int Cin, int Cout, void* stream) {
    UNET_CHECK_ARG(x && w && out && N > 0 && H > 0 && W > 0);
    UNET_CHECK_ARG(igemm_shape_ok(Cin, Cout) && ldx >= Cin && ldo >= Cout && (ldx % 4) == 0);
    UNET_CHECK_ARG(unet_aligned16(x) && unet_aligned16(w));
    IgemmArgs a{};
    a.x = x; a.w = w; a.bias = bias; a.out = out; a.ldx = ldx; a.ldo = ldo;
    a.N = N; a.H = H; a.W = W; a.Hi = H; a.Wi = W; a.Ho = 2 * H; a.Wo = 2 * W;
    a.Kdim = Cin; a.Ndim = Cout; a.in_scale = 1; a.out_scale = 2; a.relu = 0; a.flip = 0; a.tap_by_z = 1;
    return launch_igemm<1, true>(a, 4, (hipStream_t)stream);
}

// dx[n,i,j,ci] = sum_{a,b,co} dz[n,2i+a,2j+b,co] * w[a,b,co,ci]
extern "C" int unet_convT2x2_dgrad(const float* dz, int lddz, const float* w, float* dx, int lddx,
                                   int N, int H, int W, int Cin, int Cout, void* stream) {
    UNET_CHECK_ARG(dz && w && dx && N > 0 && H > 0 && W > 0);
    UNET_CHECK_ARG(igemm_shape_ok(Cout, Cin) && lddz >= Cout && lddx >= Cin && (lddz % 4) == 0);
    UNET_CHECK_ARG(unet_aligned16(dz) && unet_aligned16(w));
    IgemmArgs a{};
    a.x = dz; a.w = w; a.bias = nullptr; a.out = dx; a.ldx = lddz; a.ldo = lddx;
    a.N = N; a.H = H; a.W = W; a.Hi = 2 * H; a.Wi = 2 * W; a.Ho = H; a.Wo = W;
    a.Kdim = Cout; a.Ndim = Cin; a.in_scale = 2; a.out_scale = 1; a.relu = 0; a.flip = 0; a.tap_by_z = 0;
    return launch_igemm<2, false>(a, 1, (hipStream_t)stream);
}
