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
// (m tiles of 64) x (n tiles of 64) x S pixel-range splits with S chosen so that the launch is ONE persistent
// workgroup per CU (4 waves, one per SIMD, up to 512 VGPRs each).  A workgroup keeps ALL taps of its 64x64 block in
// registers (wave = one 32x32 quadrant x NT taps: 144 accumulator VGPRs for 3x3) and walks its pixel tiles:
//   * a tile = A rows with halo + B rows, channel-contiguous [pixel][64] in LDS (51 KB for 3x3), double-buffered;
//   * tiles are filled by LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B = 4 pixels per instruction, no VGPR
//     staging); the DMA for tile i+1 is issued before tile i's MFMAs and retired (vmcnt(0) + one barrier) after
//     them, so HBM/L2 latency hides under ~18k cycles of matrix work; out-of-image pixels read a zero page;
//   * every tap is fed from the same LDS tile: 9x the MFMA work per staged byte; a fragment read is 32 consecutive
//     floats per half-wave (conflict-free ds_read_b32); the two halves of a wave take adjacent pixels as k = 0/1.
// Each split writes its partial block to the workspace; a second pass sums the S partials in a fixed order
// (deterministic, no float atomics).
#include "common.h"

namespace {

constexpr int TW = 32;
constexpr int CT = 64;   // channel tile (both m and n)

__device__ __attribute__((aligned(256))) float g_zero_page[CT];   // zero-initialised; source for padded pixels

struct WgradArgs {
    const float* a; const float* b; float* ws;
    int lda, ldb;
    int N, H, W;       // dims of the B image (= tile grid)
    int Ha, Wa;        // dims of the A image
    int Cm, Cn;
    int tiles_y, tiles_x, n_tiles, splits;
    int mt, nt;
};

typedef __attribute__((address_space(3))) void lds_void;

template <int MODE>
__global__ __launch_bounds__(256, 1) void wgrad_kernel(WgradArgs p) {
    constexpr int NT = MODE == 0 ? 9 : 4;
    constexpr int TH = MODE == 0 ? 2 : 1;
    constexpr int A_ROWS = MODE == 0 ? TH + 2 : 2 * TH;
    constexpr int A_COLS = MODE == 0 ? TW + 2 : 2 * TW;
    constexpr int A_PIX = A_ROWS * A_COLS;
    // DMA piece = 4 pixels x 256 B (one wave instruction).  Piece k of wave w is 4k + w; A and B regions are padded to
    // multiples of 16 pixels so that piece index k is an A piece for k < KA and a B piece otherwise, for every wave.
    constexpr int A_PIX_PAD = (A_PIX + 15) & ~15;
    constexpr int B_PIX = TH * TW;
    constexpr int B_PIX_PAD = (B_PIX + 15) & ~15;
    constexpr int KA = A_PIX_PAD / 16, KB = B_PIX_PAD / 16, KP = KA + KB;   // pieces per wave
    constexpr int TILE_PIX = A_PIX_PAD + B_PIX_PAD;
    constexpr int TILE_FLOATS = TILE_PIX * CT;
    constexpr int STEPS = TH * (TW / 2);
    static_assert(KP <= STEPS, "one DMA issue per MFMA step at most");
    constexpr int NBUF = 3;                           // tile i+2 is in flight while tile i multiplies: one tile of MFMAs (~2-8 us) does
                                                      // not always cover the DMA latency under load
    static_assert(NBUF * TILE_FLOATS * 4 <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(1024))) float smem[NBUF * TILE_FLOATS];

    int bid = blockIdx.x;
    const int tmn = bid % (p.mt * p.nt);
    const int split = bid / (p.mt * p.nt);
    const int tm = tmn / p.nt, tn = tmn % p.nt;
    const int m0 = tm * CT, n0 = tn * CT;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mi = wv & 1, ni = wv >> 1;
    const int li = lane & 31, lh = lane >> 5;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int a_lane = 32 * mi + li, b_lane = 32 * ni + li;
    const int dq = lane & 15, dp = lane >> 4;         // DMA lane -> (channel quad, pixel within the piece)
    const float* zsrc = g_zero_page + 4 * dq;
    const float* abase = p.a + m0 + 4 * dq;
    const float* bbase = p.b + n0 + 4 * dq;

    // tile-invariant part of each piece's lane address: (row, col) inside the A / B region
    int prow[KP], pcol[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        if (k < KA) { const int pix = (4 * k + wv) * 4 + dp; prow[k] = pix / A_COLS; pcol[k] = pix - prow[k] * A_COLS; if (pix >= A_PIX) prow[k] = 1 << 20; }
        else        { const int pix = (4 * (k - KA) + wv) * 4 + dp; prow[k] = pix / TW; pcol[k] = pix - prow[k] * TW; if (pix >= B_PIX) prow[k] = 1 << 20; }
    }

    int t_img = 0, t_oy0 = 0, t_ox0 = 0;
    auto set_tile = [&](int tile) {
        int t = tile;
        const int tx = t % p.tiles_x; t /= p.tiles_x;
        const int ty = t % p.tiles_y;
        t_img = t / p.tiles_y; t_oy0 = ty * TH; t_ox0 = tx * TW;
    };
    // branch-free: out-of-image (or padding) lanes read the zero page
    auto issue_piece = [&](int k, float* dst) {
        const float* src;
        if (k < KA) {
            int gy, gx;
            if (MODE == 0) { gy = t_oy0 + prow[k] - 1; gx = t_ox0 + pcol[k] - 1; }
            else           { gy = 2 * t_oy0 + prow[k]; gx = 2 * t_ox0 + pcol[k]; }
            const bool ok = (unsigned)gy < (unsigned)p.Ha && (unsigned)gx < (unsigned)p.Wa;
            const float* g = abase + ((size_t)(t_img * p.Ha + gy) * p.Wa + gx) * p.lda;
            src = ok ? g : zsrc;
            __builtin_amdgcn_global_load_lds(src, (lds_void*)(dst + (4 * k + wv) * 4 * CT), 16, 0, 0);
        } else {
            const int gy = t_oy0 + prow[k], gx = t_ox0 + pcol[k];
            const bool ok = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
            const float* g = bbase + ((size_t)(t_img * p.H + gy) * p.W + gx) * p.ldb;
            src = ok ? g : zsrc;
            __builtin_amdgcn_global_load_lds(src, (lds_void*)(dst + A_PIX_PAD * CT + (4 * (k - KA) + wv) * 4 * CT), 16, 0, 0);
        }
    };

    // prologue: tiles 0 and 1 of this split in flight (a missing tile still issues its KP pieces, from the zero page, so the
    // vmcnt arithmetic below is uniform)
    int tile = split;
    auto issue_tile = [&](int t, float* dst) {
        const bool valid = t < p.n_tiles;
        set_tile(valid ? t : 0);
        if (!valid) t_oy0 = 1 << 20;                  // every pixel out of the image -> zero page
#pragma unroll
        for (int k = 0; k < KP; ++k) issue_piece(k, dst);
    };
    issue_tile(tile, smem);
    issue_tile(tile + p.splits, smem + TILE_FLOATS);
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(KP) : "memory");      // tile 0 landed; tile 1 may still be in flight
    int cur = 0;
    for (; tile < p.n_tiles; tile += p.splits) {
        const float* sA = smem + cur * TILE_FLOATS;
        const float* sB = sA + A_PIX_PAD * CT;
        float* nxt = smem + ((cur + 2) % NBUF) * TILE_FLOATS;
        const int t2 = tile + 2 * p.splits;
        const bool valid2 = t2 < p.n_tiles;            // wave-uniform
        set_tile(valid2 ? t2 : 0);
        if (!valid2) t_oy0 = 1 << 20;

        // fragments of step st+1 are read while step st's MFMAs issue; the DMA pieces of the tile after next are issued
        // one per step so their address arithmetic hides under matrix-pipe time
        float ac[NT], an[NT], bc, bn;
        auto load_frag = [&](int st, float (&av)[NT], float& bv) {
            const int py = st / (TW / 2), px = 2 * (st % (TW / 2)) + lh;
            bv = sB[(py * TW + px) * CT + b_lane];
#pragma unroll
            for (int tap = 0; tap < NT; ++tap) {
                int apix;
                if (MODE == 0) apix = (py + tap / 3) * A_COLS + px + (tap % 3);
                else           apix = (2 * py + (tap >> 1)) * A_COLS + 2 * px + (tap & 1);
                av[tap] = sA[apix * CT + a_lane];
            }
        };
        load_frag(0, ac, bc);
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            if (st + 1 < STEPS) load_frag(st + 1, an, bn);
            if (st < KP) issue_piece(st, nxt);
#pragma unroll
            for (int tap = 0; tap < NT; ++tap) acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[tap], bc, acc[tap], 0, 0, 0);
#pragma unroll
            for (int tap = 0; tap < NT; ++tap) ac[tap] = an[tap];
            bc = bn;
        }
        // the next tile (issued one iteration ago) has landed; the newest batch may stay in flight.  A bare s_barrier: the
        // fence in __syncthreads() would make the compiler drain vmcnt to 0.  lgkmcnt(0): this wave is done reading `cur`.
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(KP) : "memory");
        cur = (cur + 1) % NBUF;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // drain the tail DMAs before the LDS is released

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

// fixed-order sum of the split partials; SL slices of the split range per output so that few outputs x many splits still fill
// the chip (see wino_partial_reduce_kernel)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, long n4, int splits, int sl) {
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

int choose_splits(int mt_nt, int n_tiles, int max_workgroups) {
    int s = unet_grid_slots(256, max_workgroups) / mt_nt;           // one persistent workgroup per CU (256 CUs), or the caller's cap
    if (s < 1) s = 1;
    if (s > n_tiles) s = n_tiles;
    return s;
}

template <int MODE>
int run_wgrad(const float* a, int lda, int Ha, int Wa, const float* b, int ldb, int N, int H, int W, int Cm, int Cn,
              float* dw, float* ws, size_t ws_bytes, hipStream_t st, int max_workgroups) {
    constexpr int NT = MODE == 0 ? 9 : 4;
    constexpr int TH = MODE == 0 ? 2 : 1;
    WgradArgs p{};
    p.a = a; p.b = b; p.ws = ws; p.lda = lda; p.ldb = ldb;
    p.N = N; p.H = H; p.W = W; p.Ha = Ha; p.Wa = Wa; p.Cm = Cm; p.Cn = Cn;
    p.tiles_y = unet_cdiv(H, TH); p.tiles_x = unet_cdiv(W, TW);
    p.n_tiles = N * p.tiles_y * p.tiles_x;
    p.mt = Cm / CT; p.nt = Cn / CT;
    p.splits = choose_splits(p.mt * p.nt, p.n_tiles, max_workgroups);
    const size_t E = (size_t)NT * Cm * Cn;
    if (ws_bytes < (size_t)p.splits * E * sizeof(float)) return UNET_ENOSPC;
    wgrad_kernel<MODE><<<dim3((unsigned)(p.mt * p.nt * p.splits)), 256, 0, st>>>(p);
    int rc = UNET_LAUNCH_STATUS();
    if (rc) return rc;
    const long n4 = (long)(E / 4);
    int sl = 1;
    while (sl < 16 && 2 * sl <= p.splits && n4 * sl < 256 * 1024) sl *= 2;
    wgrad_reduce_kernel<<<(unsigned)((n4 * sl + 255) / 256), 256, 0, st>>>(ws, dw, n4, p.splits, sl);
    return UNET_LAUNCH_STATUS();
}

size_t wgrad_ws_bytes(int mode, int N, int H, int W, int Cm, int Cn, int max_workgroups) {
    const int NT = mode == 0 ? 9 : 4, TH = mode == 0 ? 2 : 1;
    const int n_tiles = N * unet_cdiv(H, TH) * unet_cdiv(W, TW);
    const int s = choose_splits((Cm / CT) * (Cn / CT), n_tiles, max_workgroups);
    return (size_t)s * NT * Cm * Cn * sizeof(float);
}

}  // namespace

extern "C" size_t unet_conv3x3_wgrad_mfma_workspace_wg(int N, int H, int W, int Cin, int Cout, int max_workgroups) {
    return wgrad_ws_bytes(0, N, H, W, Cin, Cout, max_workgroups);
}
extern "C" size_t unet_conv3x3_wgrad_mfma_workspace(int N, int H, int W, int Cin, int Cout) { return wgrad_ws_bytes(0, N, H, W, Cin, Cout, 0); }

// dw[a][b][ci][co] = sum_{n,y,x} xin[n, y+a-1, x+b-1, ci] * dz[n,y,x,co]
// max_workgroups: cap on the split-K grid (common.h unet_grid_slots); the workspace follows it (.._workspace_wg)
extern "C" int unet_conv3x3_wgrad_mfma_wg(const float* xin, int ldx, const float* dz, int lddz, float* dw,
                                          int N, int H, int W, int Cin, int Cout, int max_workgroups, void* ws, size_t ws_bytes, void* stream) {
    UNET_CHECK_ARG(xin && dz && dw && ws && N > 0 && H > 0 && W > 0);
    UNET_CHECK_ARG(Cin % CT == 0 && Cout % CT == 0 && ldx >= Cin && lddz >= Cout && ldx % 4 == 0 && lddz % 4 == 0);
    UNET_CHECK_ARG(unet_aligned16(xin) && unet_aligned16(dz) && unet_aligned16(dw) && unet_aligned16(ws));
    return run_wgrad<0>(xin, ldx, H, W, dz, lddz, N, H, W, Cin, Cout, dw, (float*)ws, ws_bytes, (hipStream_t)stream, max_workgroups);
}
extern "C" int unet_conv3x3_wgrad_mfma(const float* xin, int ldx, const float* dz, int lddz, float* dw,
                                       int N, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream) {
    return unet_conv3x3_wgrad_mfma_wg(xin, ldx, dz, lddz, dw, N, H, W, Cin, Cout, 0, ws, ws_bytes, stream);
}

extern "C" size_t unet_convT2x2_wgrad_workspace_wg(int N, int H, int W, int Cin, int Cout, int max_workgroups) {
    return wgrad_ws_bytes(1, N, H, W, Cout, Cin, max_workgroups);
}
extern "C" size_t unet_convT2x2_wgrad_workspace(int N, int H, int W, int Cin, int Cout) { return wgrad_ws_bytes(1, N, H, W, Cout, Cin, 0); }

// dw[a][b][co][ci] = sum_{n,i,j} dz[n,2i+a,2j+b,co] * xin[n,i,j,ci]      (H, W are the INPUT dims of the layer)
extern "C" int unet_convT2x2_wgrad_wg(const float* xin, int ldx, const float* dz, int lddz, float* dw,
                                      int N, int H, int W, int Cin, int Cout, int max_workgroups, void* ws, size_t ws_bytes, void* stream) {
    UNET_CHECK_ARG(xin && dz && dw && ws && N > 0 && H > 0 && W > 0);
    UNET_CHECK_ARG(Cin % CT == 0 && Cout % CT == 0 && ldx >= Cin && lddz >= Cout && ldx % 4 == 0 && lddz % 4 == 0);
    UNET_CHECK_ARG(unet_aligned16(xin) && unet_aligned16(dz) && unet_aligned16(dw) && unet_aligned16(ws));
    return run_wgrad<1>(dz, lddz, 2 * H, 2 * W, xin, ldx, N, H, W, Cout, Cin, dw, (float*)ws, ws_bytes, (hipStream_t)stream, max_workgroups);
}
extern "C" int unet_convT2x2_wgrad(const float* xin, int ldx, const float* dz, int lddz, float* dw,
                                   int N, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream) {
    return unet_convT2x2_wgrad_wg(xin, ldx, dz, lddz, dw, N, H, W, Cin, Cout, 0, ws, ws_bytes, stream);
}
