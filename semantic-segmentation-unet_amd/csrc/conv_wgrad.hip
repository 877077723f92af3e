// Weight gradients on the fp32 matrix cores:
//
//   dW[tap][m][n] = sum_{pixels p} A[pixA(p, tap)][m] * B[p][n]
//
//   MODE 0  3x3 'same' conv (derived backward of UNet/model.py:28-35):  A = layer input (m = ci) read at
//           p + (a-1, b-1) with zero padding, B = dz (n = co)  ->  dW in the forward HWIO layout [a][b][ci][co];
//   MODE 1  2x2/stride-2 transposed conv (UNet/model.py:41-46):  A = dz (m = co) read at (2i+a, 2j+b),
//           B = layer input (n = ci) at (i, j)  ->  dW in the Keras layout [a][b][co][ci].
//
// The contraction runs over pixels (GEMM K = N*H*W, up to 2M), so the grid is
// (m tiles of 64) x (n tiles of 64) x S pixel-range splits.  A workgroup (4 waves) keeps ALL taps of its
// 64x64 block in registers (wave = one 32x32 quadrant x NT taps: 144 accumulator VGPRs for 3x3), stages a
// spatial tile of A (with halo) and B in LDS once and feeds every tap from it: 9x the MFMA work per staged byte.
// Both operands are channel-contiguous in LDS, so a fragment read is 32 consecutive floats per half-wave
// (conflict-free ds_read_b32); the two halves of a wave take adjacent pixels as MFMA k = 0/1.
// Each split writes its partial block to the workspace; a second pass sums the S partials in a fixed order
// (deterministic, no float atomics).
#include "common.h"

namespace {

constexpr int TW = 32;
constexpr int CT = 64;   // channel tile (both m and n)

struct WgradArgs {
    const float* a; const float* b; float* ws;
    int lda, ldb;
    int N, H, W;       // dims of the B image (= tile grid)
    int Ha, Wa;        // dims of the A image
    int Cm, Cn;
    int tiles_y, tiles_x, n_tiles, splits;
    int mt, nt;
};

template <int MODE>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(WgradArgs p) {
    constexpr int NT = MODE == 0 ? 9 : 4;
    constexpr int TH = MODE == 0 ? 2 : 1;
    constexpr int A_ROWS = MODE == 0 ? TH + 2 : 2 * TH;
    constexpr int A_COLS = MODE == 0 ? TW + 2 : 2 * TW;
    constexpr int A_PIX = A_ROWS * A_COLS, B_PIX = TH * TW;
    __shared__ __attribute__((aligned(16))) float smem[(A_PIX + B_PIX) * CT];
    float* sA = smem;
    float* sB = smem + A_PIX * CT;

    int bid = blockIdx.x;
    const int tmn = bid % (p.mt * p.nt);
    const int split = bid / (p.mt * p.nt);
    const int tm = tmn / p.nt, tn = tmn % p.nt;
    const int m0 = tm * CT, n0 = tn * CT;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int mi = wv & 1, ni = wv >> 1;
    const int li = lane & 31, lh = lane >> 5;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int a_lane = 32 * mi + li, b_lane = 32 * ni + li;

    for (int tile = split; tile < p.n_tiles; tile += p.splits) {
        int t = tile;
        const int tx = t % p.tiles_x; t /= p.tiles_x;
        const int ty = t % p.tiles_y;
        const int img = t / p.tiles_y;
        const int oy0 = ty * TH, ox0 = tx * TW;

        __syncthreads();
        for (int idx = tid; idx < A_PIX * (CT / 4); idx += 256) {
            const int pix = idx >> 4, q = idx & 15;
            const int iy = pix / A_COLS, ix = pix - iy * A_COLS;
            int gy, gx;
            if (MODE == 0) { gy = oy0 + iy - 1; gx = ox0 + ix - 1; }
            else           { gy = 2 * oy0 + iy; gx = 2 * ox0 + ix; }
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (gy >= 0 && gy < p.Ha && gx >= 0 && gx < p.Wa)
                v = *reinterpret_cast<const f32x4*>(p.a + ((size_t)(img * p.Ha + gy) * p.Wa + gx) * p.lda + m0 + 4 * q);
            *reinterpret_cast<f32x4*>(sA + pix * CT + 4 * q) = v;
        }
        for (int idx = tid; idx < B_PIX * (CT / 4); idx += 256) {
            const int pix = idx >> 4, q = idx & 15;
            const int iy = pix / TW, ix = pix - iy * TW;
            const int gy = oy0 + iy, gx = ox0 + ix;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (gy < p.H && gx < p.W)
                v = *reinterpret_cast<const f32x4*>(p.b + ((size_t)(img * p.H + gy) * p.W + gx) * p.ldb + n0 + 4 * q);
            *reinterpret_cast<f32x4*>(sB + pix * CT + 4 * q) = v;
        }
        __syncthreads();

#pragma unroll
        for (int py = 0; py < TH; ++py) {
#pragma unroll
            for (int sx = 0; sx < TW / 2; ++sx) {
                const int px = 2 * sx + lh;
                const float bv = sB[(py * TW + px) * CT + b_lane];
#pragma unroll
                for (int tap = 0; tap < NT; ++tap) {
                    int apix;
                    if (MODE == 0) apix = (py + tap / 3) * A_COLS + px + (tap % 3);
                    else           apix = (2 * py + (tap >> 1)) * A_COLS + 2 * px + (tap & 1);
                    const float av = sA[apix * CT + a_lane];
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[tap], 0, 0, 0);
                }
            }
        }
    }

    // partial block -> workspace [split][tap][Cm][Cn]
#pragma unroll
    for (int tap = 0; tap < NT; ++tap) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            p.ws[(((size_t)split * NT + tap) * p.Cm + m0 + 32 * mi + row) * p.Cn + n0 + 32 * ni + li] = acc[tap][r];
        }
    }
}

__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, long n4, int splits) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4 s = reinterpret_cast<const f32x4*>(ws)[i];
        for (int k = 1; k < splits; ++k) s += reinterpret_cast<const f32x4*>(ws)[(size_t)k * n4 + i];
        reinterpret_cast<f32x4*>(dw)[i] = s;
    }
}

int choose_splits(int mt_nt, int n_tiles) {
    int s = 1024 / mt_nt;
    if (s < 1) s = 1;
    if (s > n_tiles) s = n_tiles;
    return s;
}

template <int MODE>
int run_wgrad(const float* a, int lda, int Ha, int Wa, const float* b, int ldb, int N, int H, int W, int Cm, int Cn,
              float* dw, float* ws, size_t ws_bytes, hipStream_t st) {
    constexpr int NT = MODE == 0 ? 9 : 4;
    constexpr int TH = MODE == 0 ? 2 : 1;
    WgradArgs p{};
    p.a = a; p.b = b; p.ws = ws; p.lda = lda; p.ldb = ldb;
    p.N = N; p.H = H; p.W = W; p.Ha = Ha; p.Wa = Wa; p.Cm = Cm; p.Cn = Cn;
    p.tiles_y = unet_cdiv(H, TH); p.tiles_x = unet_cdiv(W, TW);
    p.n_tiles = N * p.tiles_y * p.tiles_x;
    p.mt = Cm / CT; p.nt = Cn / CT;
    p.splits = choose_splits(p.mt * p.nt, p.n_tiles);
    const size_t E = (size_t)NT * Cm * Cn;
    if (ws_bytes < (size_t)p.splits * E * sizeof(float)) return UNET_ENOSPC;
    wgrad_kernel<MODE><<<dim3((unsigned)(p.mt * p.nt * p.splits)), 256, 0, st>>>(p);
    int rc = UNET_LAUNCH_STATUS();
    if (rc) return rc;
    const long n4 = (long)(E / 4);
    int blocks = unet_cdiv(n4, 256);
    if (blocks > 2048) blocks = 2048;
    wgrad_reduce_kernel<<<blocks, 256, 0, st>>>(ws, dw, n4, p.splits);
    return UNET_LAUNCH_STATUS();
}

size_t wgrad_ws_bytes(int mode, int N, int H, int W, int Cm, int Cn) {
    const int NT = mode == 0 ? 9 : 4, TH = mode == 0 ? 2 : 1;
    const int n_tiles = N * unet_cdiv(H, TH) * unet_cdiv(W, TW);
    const int s = choose_splits((Cm / CT) * (Cn / CT), n_tiles);
    return (size_t)s * NT * Cm * Cn * sizeof(float);
}

}  // namespace

extern "C" size_t unet_conv3x3_wgrad_mfma_workspace(int N, int H, int W, int Cin, int Cout) {
    return wgrad_ws_bytes(0, N, H, W, Cin, Cout);
}

// dw[a][b][ci][co] = sum_{n,y,x} xin[n, y+a-1, x+b-1, ci] * dz[n,y,x,co]
extern "C" int unet_conv3x3_wgrad_mfma(const float* xin, int ldx, const float* dz, int lddz, float* dw,
                                       int N, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream) {
    UNET_CHECK_ARG(xin && dz && dw && ws && N > 0 && H > 0 && W > 0);
    UNET_CHECK_ARG(Cin % CT == 0 && Cout % CT == 0 && ldx >= Cin && lddz >= Cout && ldx % 4 == 0 && lddz % 4 == 0);
    UNET_CHECK_ARG(unet_aligned16(xin) && unet_aligned16(dz) && unet_aligned16(dw) && unet_aligned16(ws));
    return run_wgrad<0>(xin, ldx, H, W, dz, lddz, N, H, W, Cin, Cout, dw, (float*)ws, ws_bytes, (hipStream_t)stream);
}

extern "C" size_t unet_convT2x2_wgrad_workspace(int N, int H, int W, int Cin, int Cout) {
    return wgrad_ws_bytes(1, N, H, W, Cout, Cin);
}

// dw[a][b][co][ci] = sum_{n,i,j} dz[n,2i+a,2j+b,co] * xin[n,i,j,ci]      (H, W are the INPUT dims of the layer)
extern "C" int unet_convT2x2_wgrad(const float* xin, int ldx, const float* dz, int lddz, float* dw,
                                   int N, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream) {
    UNET_CHECK_ARG(xin && dz && dw && ws && N > 0 && H > 0 && W > 0);
    UNET_CHECK_ARG(Cin % CT == 0 && Cout % CT == 0 && ldx >= Cin && lddz >= Cout && ldx % 4 == 0 && lddz % 4 == 0);
    UNET_CHECK_ARG(unet_aligned16(xin) && unet_aligned16(dz) && unet_aligned16(dw) && unet_aligned16(ws));
    return run_wgrad<1>(dz, lddz, 2 * H, 2 * W, xin, ldx, N, H, W, Cout, Cin, dw, (float*)ws, ws_bytes, (hipStream_t)stream);
}
