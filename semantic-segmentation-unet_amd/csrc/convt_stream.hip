// 2x2 / stride-2 transposed convolution (UNet/model.py:41-46), forward, as a persistent fp32-MFMA stream kernel.
//
//   out[n, 2i+a, 2j+b, co] = bias[co] + sum_ci x[n,i,j,ci] * w[a,b,co,ci]          (Keras kernel layout [a][b][co][ci])
//
// k = s = 2 does not overlap: four independent 1x1 GEMMs [pixels x Cin] . [Cin x Cout] that share the data operand.  A wave
// keeps 4 taps x (64 pixels x 64 channels) = 16 accumulators of 32x32 (all 256 AGPRs, one wave per SIMD), a workgroup
// PW x CW such waves (2x2: 128 px x 128 co; 4x1: 256 px x 64 co for Cout = 64).  Cin streams in chunks of 8 through a 4-slot
// LDS ring filled by LDS-DMA straight from the tensors' own layouts (lane -> (row, 16-byte half); the half is XOR-ed with
// bit 3 of the row on the source side so operand reads are conflict-free ds_read_b128): no weight re-layout, no VGPR staging.
//
// The schedule follows the issue-cost model of DESIGN.md 3.1 (one wave per SIMD: VALU never hides, LDS reads and DMA do):
// per chunk ONE barrier, then 64 MFMAs as an explicit instruction stream with the next chunk's 10 operand reads and the
// wave's 4-5 DMAs for the chunk three ahead placed behind them; operand registers are double-buffered; there is no other
// VALU work in the loop than one 64-bit add per DMA.  The chunk stream runs across tile boundaries (a tile's first chunk
// starts its accumulators from C = 0), so only a workgroup's first tile pays the DMA latency.  Weights are the MFMA's A
// operand: a lane ends with 16 channels of one pixel in runs of 4 and the epilogue (+bias, scatter to (2i+a, 2j+b)) stores
// dwordx4.
#include "common.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;

struct ConvtFwdArgs {
    const float* x; const float* w; const float* bias; float* out;
    int ldx, ldo, N, H, W, Cin, Cout;
    long P;              // N*H*W input pixels
    int npt, nct, ntiles;
    float* stat_part;    // BatchNorm sums of the output (UNet/model.py:47), see the end of the kernel; or null
};

#define CT_RD128(dst, base, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst) : "v"(base), "n"(off))
#define CT_MFMA(accv, av, bv) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(accv) : "v"(av), "v"(bv) : "memory")
#define CT_MFMA0(accv, av, bv) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=a"(accv) : "v"(av), "v"(bv) : "memory")

struct CtFrags { f32x4 xf[2]; f32x4 wf[4][2]; };
#define CT_ALL(F) "+v"(F.xf[0]), "+v"(F.xf[1]), "+v"(F.wf[0][0]), "+v"(F.wf[0][1]), "+v"(F.wf[1][0]), "+v"(F.wf[1][1]), \
                  "+v"(F.wf[2][0]), "+v"(F.wf[2][1]), "+v"(F.wf[3][0]), "+v"(F.wf[3][1])

// One chunk: 64 MFMAs from `cur`; behind them the operand reads of the next chunk (ring slot NSLOT) into `nxt` and this wave's
// DMAs.  SLOT / NSLOT are the ring slots of this and the next chunk (compile-time: all LDS offsets are immediates).
template <int PW, int SLOT, bool FIRST, class DMA>
__device__ __forceinline__ void ct_chunk(f32x16 (&acc)[16], const CtFrags& cur, CtFrags& nxt, unsigned xa, unsigned wa_lo,
                                         unsigned wa_hi, DMA&& dma) {
    constexpr int CW = 4 / PW, XP = 2 * PW, NP = XP + 8 * CW, PPW = NP / 4;
    constexpr int SLOTB = NP * 1024;
    constexpr int NS = (SLOT + 1) & 3;
    constexpr int WROWB = CW * 64 * 32;                      // bytes of one tap's weight rows
#pragma unroll
    for (int n = 0; n < 16; ++n) {
        const int tap = n >> 2, pt = (n >> 1) & 1, ct = n & 1;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (FIRST && s == 0) CT_MFMA0(acc[n], cur.wf[tap][ct][s], cur.xf[pt][s]);
            else CT_MFMA(acc[n], cur.wf[tap][ct][s], cur.xf[pt][s]);
            const int m = 4 * n + s;                         // MFMA number 0..63
            if (m < 2) CT_RD128(nxt.xf[m], xa, NS * SLOTB + m * 1024);
            else if (m < 10) {
                const int t2 = (m - 2) >> 1, c2 = (m - 2) & 1;
                if (NS < 2) CT_RD128(nxt.wf[t2][c2], wa_lo, NS * SLOTB + t2 * WROWB + c2 * 1024);
                else CT_RD128(nxt.wf[t2][c2], wa_hi, (NS - 2) * SLOTB + t2 * WROWB + c2 * 1024);
            }
            if (m >= 12 && ((m - 12) & 7) == 0 && ((m - 12) >> 3) < PPW) dma((m - 12) >> 3);
        }
    }
}

template <int PW, bool STATS>
__device__ __forceinline__ void convt_fwd_stream_body(const ConvtFwdArgs& p) {
    constexpr int CW = 4 / PW, XP = 2 * PW, NP = XP + 8 * CW, PPW = NP / 4;
    constexpr int SLOTB = NP * 1024, SLOTF = SLOTB / 4;
    constexpr int TPX = PW * 64, TCO = CW * 64;
    __shared__ __attribute__((aligned(1024))) float smem[4 * SLOTF];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pi = wv % PW, cw = wv / PW;
    const int li = lane & 31, lh = lane >> 5;
    const int drow = lane >> 1, dh = lane & 1;
    const int nchunks = p.Cin >> 3;

    // DMA duty: pieces wv + 4k; per-lane byte offsets from the (tile, chunk) scalar bases
    const int hs = dh ^ ((drow >> 3) & 1);
    unsigned doff[PPW];
#pragma unroll
    for (int k = 0; k < PPW; ++k) {
        const int id = wv + 4 * k;
        if (id < XP) doff[k] = (unsigned)(((32 * id + drow) * p.ldx + 4 * hs) * 4);
        else {
            const int wp = id - XP, tap = wp / (2 * CW), rb = wp % (2 * CW);
            doff[k] = (unsigned)((((size_t)tap * p.Cout + 32 * rb + drow) * p.Cin + 4 * hs) * 4);
        }
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_t*)smem;
    const unsigned xa = lds0 + (unsigned)((64 * pi + li) * 32 + 16 * (lh ^ ((li >> 3) & 1)));
    const unsigned wa_lo = lds0 + XP * 1024 + (unsigned)((64 * cw + li) * 32 + 16 * (lh ^ ((li >> 3) & 1)));
    const unsigned wa_hi = wa_lo + 2 * SLOTB;

    // issue position of the DMA stream: (tile, chunk), three chunks ahead of the MFMAs
    int it_tile = blockIdx.x, it_chunk = 0;
    auto tile_bases = [&](int t, const char*& xb, const char*& wb) {
        const int tc = t < p.ntiles ? t : (int)blockIdx.x;                       // past the end: re-read, never used
        const int pt = tc / p.nct, ct = tc % p.nct;
        xb = reinterpret_cast<const char*>(p.x + (size_t)pt * TPX * p.ldx);
        wb = reinterpret_cast<const char*>(p.w + (size_t)ct * TCO * p.Cin);
    };
    const char* ixb; const char* iwb;
    tile_bases(it_tile, ixb, iwb);
    const float* src[PPW];
    auto next_sources = [&]() {                                                   // sources of the next batch, then advance
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            const bool isx = (wv + 4 * k) < XP;                                  // (same for every wave: XP is a multiple of 4)
            src[k] = reinterpret_cast<const float*>((isx ? ixb : iwb) + (size_t)it_chunk * 32 + doff[k]);
        }
        if (PPW == 5) asm volatile("" : "+v"(src[0]), "+v"(src[1]), "+v"(src[2]), "+v"(src[3]), "+v"(src[PPW - 1]));
        else asm volatile("" : "+v"(src[0]), "+v"(src[1]), "+v"(src[2]), "+v"(src[3]));
        if (++it_chunk == nchunks) { it_chunk = 0; it_tile += gridDim.x; tile_bases(it_tile, ixb, iwb); }
    };
    auto issue = [&](int k, int slot) {
        __builtin_amdgcn_global_load_lds(src[k], (lds_void_t*)(smem + slot * SLOTF + (wv + 4 * k) * 256), 16, 0, 0);
    };

    // prologue: chunks 0, 1, 2 in flight; chunk 0 landed; its operands in registers
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        next_sources();
#pragma unroll
        for (int k = 0; k < PPW; ++k) issue(k, s);
    }
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(PPW) : "memory");          // chunks 0 and 1 landed
    CtFrags fa, fb;
    {
        constexpr int WROWB = CW * 64 * 32;
        CT_RD128(fa.xf[0], xa, 0); CT_RD128(fa.xf[1], xa, 1024);
#pragma unroll
        for (int t2 = 0; t2 < 4; ++t2) { CT_RD128(fa.wf[t2][0], wa_lo, t2 * WROWB); CT_RD128(fa.wf[t2][1], wa_lo, t2 * WROWB + 1024); }
    }

    f32x16 acc[16];
    const int Wo = 2 * p.W;
    // STATS: per-lane running sum / sum of squares of the stored values for the lane's 32 channels.  A persistent workgroup
    // only ever sees one channel tile (tile ids advance by gridDim.x, a multiple of nct), so they run over all its tiles.
    f32x4 st1[2][4], st2[2][4];
#pragma unroll
    for (int i = 0; i < 8; ++i) { st1[i >> 2][i & 3] = f32x4{0.f, 0.f, 0.f, 0.f}; st2[i >> 2][i & 3] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    for (int t = blockIdx.x; t < p.ntiles; t += gridDim.x) {
        // the wave's 16 bias values (latency hidden under the chunk loop)
        const int ptile = t / p.nct, ctile = t % p.nct;
        const int cb = ctile * TCO + 64 * cw + 4 * lh;
        f32x4 bias4[2][4];
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bias4[c2][g] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (p.bias) bias4[c2][g] = *reinterpret_cast<const f32x4*>(p.bias + cb + 32 * c2 + 8 * g);
            }
        // top of a chunk: its operands are back (lgkmcnt), the NEXT chunk has landed (all but the newest DMA batch), and
        // every wave is past the previous chunk (bare s_barrier: __syncthreads() would drain vmcnt to 0).
        // vmcnt(PPW) is exact while the newest outstanding operations are this wave's DMA loads (loads retire in order).
        // A tile's first chunk comes right after the previous tile's epilogue STORES, which retire independently of loads:
        // there the "next chunk landed" wait is done before the epilogue instead (below) and the top waits for the operands
        // only.  (The first group of four is peeled: a FIRST / non-FIRST branch inside the loop makes the allocator spill.)
#define CT_TOP(F) asm volatile("s_waitcnt vmcnt(%10) lgkmcnt(0)\n\ts_barrier" : CT_ALL(F) : "n"(PPW) : "memory")
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : CT_ALL(fa) : : "memory");
        next_sources();
        ct_chunk<PW, 0, true>(acc, fa, fb, xa, wa_lo, wa_hi, [&](int k) { issue(k, 3); });
        CT_TOP(fb); next_sources();
        ct_chunk<PW, 1, false>(acc, fb, fa, xa, wa_lo, wa_hi, [&](int k) { issue(k, 0); });
        CT_TOP(fa); next_sources();
        ct_chunk<PW, 2, false>(acc, fa, fb, xa, wa_lo, wa_hi, [&](int k) { issue(k, 1); });
        CT_TOP(fb); next_sources();
        ct_chunk<PW, 3, false>(acc, fb, fa, xa, wa_lo, wa_hi, [&](int k) { issue(k, 2); });
        for (int c = 4; c < nchunks; c += 4) {
            CT_TOP(fa); next_sources();
            ct_chunk<PW, 0, false>(acc, fa, fb, xa, wa_lo, wa_hi, [&](int k) { issue(k, 3); });
            CT_TOP(fb); next_sources();
            ct_chunk<PW, 1, false>(acc, fb, fa, xa, wa_lo, wa_hi, [&](int k) { issue(k, 0); });
            CT_TOP(fa); next_sources();
            ct_chunk<PW, 2, false>(acc, fa, fb, xa, wa_lo, wa_hi, [&](int k) { issue(k, 1); });
            CT_TOP(fb); next_sources();
            ct_chunk<PW, 3, false>(acc, fb, fa, xa, wa_lo, wa_hi, [&](int k) { issue(k, 2); });
        }
#undef CT_TOP
        // the chunk after the tile's last one (= the next tile's second) has landed: only DMA loads are outstanding here;
        // + MFMA -> accumulator-read distance (inline-asm MFMAs are invisible to the hazard recogniser)
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_nop 15\n\ts_nop 15" :: "n"(PPW) : "memory");

        // epilogue: acc[(tap*2 + pt)*2 + ct][4g + e] = channel cb + 32*ct + 8*g + e of pixel ptile*TPX + 64*pi + 32*pt + li
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            const long px = (long)ptile * TPX + 64 * pi + 32 * pt + li;
            long q = px; const int xx = (int)(q % p.W); q /= p.W; const int yy = (int)(q % p.H); const int n = (int)(q / p.H);
            float* o00 = p.out + ((size_t)(n * 2 * p.H + 2 * yy) * Wo + 2 * xx) * p.ldo + cb;
#pragma unroll
            for (int tap = 0; tap < 4; ++tap) {
                float* o = o00 + ((size_t)(tap >> 1) * Wo + (tap & 1)) * p.ldo;
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float e[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            asm("v_accvgpr_read_b32 %0, %1" : "=v"(e[k]) : "a"(acc[(tap * 2 + pt) * 2 + c2][4 * g + k]));
                        const f32x4 v = f32x4{e[0], e[1], e[2], e[3]} + bias4[c2][g];
                        *reinterpret_cast<f32x4*>(o + 32 * c2 + 8 * g) = v;
                        if (STATS) { st1[c2][g] += v; st2[c2][g] += v * v; }
                    }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : CT_ALL(fa) : : "memory");      // retire the tail prefetches
    if (STATS) {
        // reduce over the 32 pixel lanes of each half-wave; row layout as unet_bn_train_finalize_partials expects:
        // stat_part[64-channel block][row][64][2], block = ctile*CW + cw, row = (first tile / nct) * PW + pi
        float v[64];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[8 * i + 2 * k] = st1[i >> 2][i & 3][k]; v[8 * i + 2 * k + 1] = st2[i >> 2][i & 3][k]; }
#pragma unroll
        for (int i = 0; i < 64; ++i)
#pragma unroll
            for (int m = 1; m < 32; m <<= 1) v[i] += __shfl_xor(v[i], m, 32);
        if (li == 0) {
            const int t0 = blockIdx.x, ctile0 = t0 % p.nct;
            const int rows = ((int)gridDim.x / p.nct) * PW;
            float* o = p.stat_part + ((size_t)(ctile0 * CW + cw) * rows + (t0 / p.nct) * PW + pi) * 128;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int ch = 32 * (i >> 2) + 8 * (i & 3) + 4 * lh + k;
                    o[2 * ch] = v[8 * i + 2 * k]; o[2 * ch + 1] = v[8 * i + 2 * k + 1];
                }
        }
    }
}

// (plain kernels around the templated body: the host-side stub of a kernel TEMPLATE containing this inline asm is not emitted)
__global__ __launch_bounds__(256, 1) void convt_fwd_stream_kernel_2x2(ConvtFwdArgs p) { convt_fwd_stream_body<2, false>(p); }
__global__ __launch_bounds__(256, 1) void convt_fwd_stream_kernel_4x1(ConvtFwdArgs p) { convt_fwd_stream_body<4, false>(p); }
__global__ __launch_bounds__(256, 1) void convt_fwd_stream_stats_kernel_2x2(ConvtFwdArgs p) { convt_fwd_stream_body<2, true>(p); }
__global__ __launch_bounds__(256, 1) void convt_fwd_stream_stats_kernel_4x1(ConvtFwdArgs p) { convt_fwd_stream_body<4, true>(p); }

int convt_cus() {
    static const int cus = [] { int d = 0, n = 0; if (hipGetDevice(&d) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || n <= 0) n = 256; return n; }();
    return cus;
}

}  // namespace

extern "C" int unet_convT2x2_fwd_stream_supported(int N, int H, int W, int Cin, int Cout) {
    const long P = (long)N * H * W;
    if (N <= 0 || H <= 0 || W <= 0 || Cin % 32 != 0 || Cout % 64 != 0) return 0;
    const int tpx = Cout % 128 == 0 ? 128 : 256;
    return (P % tpx == 0 && (long)4 * Cout * Cin * 4 < (1L << 31) && (long)tpx * 4096 * 4 < (1L << 31)) ? 1 : 0;
}

extern "C" int unet_convT2x2_fwd_stream_stats_rows_wg(int N, int H, int W, int Cin, int Cout, int max_workgroups);

static int convt_fwd_stream_launch(const float* x, int ldx, const float* w, const float* bias, float* out, int ldo,
                                   int N, int H, int W, int Cin, int Cout, float* stat_part, size_t stat_bytes, int max_workgroups, void* stream) {
    UNET_CHECK_ARG(x && w && out && unet_convT2x2_fwd_stream_supported(N, H, W, Cin, Cout));
    UNET_CHECK_ARG(ldx >= Cin && ldo >= Cout && ldx % 4 == 0 && ldo % 4 == 0 && ldx <= 4096);
    UNET_CHECK_ARG(unet_aligned16(x) && unet_aligned16(w) && unet_aligned16(out) && (!bias || unet_aligned16(bias)));
    ConvtFwdArgs a{};
    a.x = x; a.w = w; a.bias = bias; a.out = out; a.ldx = ldx; a.ldo = ldo; a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.P = (long)N * H * W; a.stat_part = stat_part;
    const bool wide = Cout % 128 == 0;
    a.npt = (int)(a.P / (wide ? 128 : 256)); a.nct = Cout / (wide ? 128 : 64);
    const long tiles = (long)a.npt * a.nct;
    if (tiles > 0x7fffffffL) return UNET_EINVAL;
    a.ntiles = (int)tiles;
    const long slots = unet_grid_slots(convt_cus(), max_workgroups);
    const unsigned grid = (unsigned)(tiles < slots ? tiles : slots);
    hipStream_t st = (hipStream_t)stream;
    if (stat_part) {
        const int rows = unet_convT2x2_fwd_stream_stats_rows_wg(N, H, W, Cin, Cout, max_workgroups);
        UNET_CHECK_ARG(rows > 0);
        if (stat_bytes < (size_t)(Cout / 64) * rows * 128 * sizeof(float)) return UNET_ENOSPC;
        if (wide) convt_fwd_stream_stats_kernel_2x2<<<dim3(grid), 256, 0, st>>>(a);
        else      convt_fwd_stream_stats_kernel_4x1<<<dim3(grid), 256, 0, st>>>(a);
    } else {
        if (wide) convt_fwd_stream_kernel_2x2<<<dim3(grid), 256, 0, st>>>(a);
        else      convt_fwd_stream_kernel_4x1<<<dim3(grid), 256, 0, st>>>(a);
    }
    return UNET_LAUNCH_STATUS();
}

// rows of statistics partials per 64-channel block (0: shape not supported / grid not a multiple of the channel-tile count)
extern "C" int unet_convT2x2_fwd_stream_stats_rows_wg(int N, int H, int W, int Cin, int Cout, int max_workgroups) {
    if (!unet_convT2x2_fwd_stream_supported(N, H, W, Cin, Cout)) return 0;
    const bool wide = Cout % 128 == 0;
    const long tiles = ((long)N * H * W / (wide ? 128 : 256)) * (Cout / (wide ? 128 : 64));
    const long slots = unet_grid_slots(convt_cus(), max_workgroups);
    const long grid = tiles < slots ? tiles : slots;
    const int nct = Cout / (wide ? 128 : 64);
    return grid % nct == 0 ? (int)((grid / nct) * (wide ? 2 : 4)) : 0;
}
extern "C" int unet_convT2x2_fwd_stream_stats_rows(int N, int H, int W, int Cin, int Cout) {
    return unet_convT2x2_fwd_stream_stats_rows_wg(N, H, W, Cin, Cout, 0);
}

extern "C" int unet_convT2x2_fwd_stream(const float* x, int ldx, const float* w, const float* bias, float* out, int ldo,
                                        int N, int H, int W, int Cin, int Cout, void* stream) {
    return convt_fwd_stream_launch(x, ldx, w, bias, out, ldo, N, H, W, Cin, Cout, nullptr, 0, 0, stream);
}
// max_workgroups: cap on the persistent grid (common.h unet_grid_slots); stat_part nullable here
extern "C" int unet_convT2x2_fwd_stream_wg(const float* x, int ldx, const float* w, const float* bias, float* out, int ldo,
                                           int N, int H, int W, int Cin, int Cout, float* stat_part, size_t stat_bytes, int max_workgroups, void* stream) {
    return convt_fwd_stream_launch(x, ldx, w, bias, out, ldo, N, H, W, Cin, Cout, stat_part, stat_bytes, max_workgroups, stream);
}

// + BatchNorm sums of the output (layout and finalize as for unet_conv3x3_fwd_winograd_fused_stats)
extern "C" int unet_convT2x2_fwd_stream_stats(const float* x, int ldx, const float* w, const float* bias, float* out, int ldo,
                                              int N, int H, int W, int Cin, int Cout, float* stat_part, size_t stat_bytes, void* stream) {
    UNET_CHECK_ARG(stat_part);
    return convt_fwd_stream_launch(x, ldx, w, bias, out, ldo, N, H, W, Cin, Cout, stat_part, stat_bytes, 0, stream);
}

// ---- transposed-conv weight gradient, wide tiles ---------------------------------------------------------------------------
//   dw[a][b][co][ci] = sum_{n,i,j} dz[n,2i+a,2j+b,co] * x[n,i,j,ci]
// The contraction runs over pixels, so operands are read from LDS images in their natural [pixel][channel] layout
// (conflict-free ds_read_b32, the lane half picks the pixel of the k pair) like conv_wgrad.hip's kernel -- but a wave owns
// 4 taps x (64 or 32 co) x 64 ci = 16 (8) accumulators instead of 4, which doubles the MFMA work per LDS-filled byte (the
// narrow kernel sat at busy x clock = 1.35-1.48 GHz against the 1.85-1.9 every other MFMA kernel here reaches).  Workgroup =
// CTM (128, or 64 when Cout % 128 != 0) output channels x 128 input channels, persistent over a split of the pixel tiles
// (16 input pixels of one row = 2 x 32 dz pixels); 3-slot LDS-DMA ring; per tile 8 k-steps of 16 (8) MFMAs as an explicit
// stream with the next step's operand reads and the DMAs of the tile after next behind them.
namespace {

struct ConvtWgArgs {
    const float* x; const float* dz; float* ws;
    int ldx, lddz, N, H, W, Cin, Cout;
    int mt, nt, splits, n_tiles;
};

#define CW_RD32(dst, base, off) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=&v"(dst) : "v"(base), "n"(off))

template <int TM>
__device__ __forceinline__ void convt_wgrad_wide_body(const ConvtWgArgs& p) {
    constexpr int CTM = 64 * TM, CTN = 128;
    constexpr int QA = CTM / 4, PPA = 64 / QA;               // lanes per dz pixel, dz pixels per 1-KB piece
    constexpr int NPA = QA, NPB = 8, NP = NPA + NPB, PPW = NP / 4;
    constexpr int SLOTB = NP * 1024, SLOTF = SLOTB / 4;
    constexpr int NMF = 4 * TM * 2;                          // MFMAs per k-step
    __shared__ __attribute__((aligned(1024))) float smem[3 * SLOTF];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mi = wv & 1, ni = wv >> 1;
    const int li = lane & 31, lh = lane >> 5;
    int bid = blockIdx.x;
    const int tmn = bid % (p.mt * p.nt), split = bid / (p.mt * p.nt);
    const int m0 = (tmn / p.nt) * CTM, n0 = (tmn % p.nt) * CTN;
    const int tpr = p.W >> 4;                                // tiles per image row

    // DMA duty: pieces wv + 4k; k < KAW are dz pieces for every wave (NPA is a multiple of 4), the rest x pieces
    unsigned doff[PPW];
#pragma unroll
    for (int k = 0; k < PPW; ++k) {
        const int id = wv + 4 * k;
        if (id < NPA) { const int pix = id * PPA + lane / QA; doff[k] = (unsigned)((((pix >> 5) * 2 * p.W + (pix & 31)) * p.lddz + 4 * (lane % QA)) * 4); }
        else          { const int pix = (id - NPA) * 2 + (lane >> 5); doff[k] = (unsigned)((pix * p.ldx + 4 * (lane & 31)) * 4); }
    }
    // tile t = (image row `row` = t / tpr, 16-pixel segment t % tpr):  x pixel offset = 16 t,  dz pixel offset = 32 t + 2W row
    const float* src_next[PPW];
    auto sources = [&](int t, int row) {
        const bool ok = t < p.n_tiles;                       // past the end: re-read the first tile, never used
        const int tc = ok ? t : split, rc = ok ? row : split / tpr;
        const char* dzb = reinterpret_cast<const char*>(p.dz + ((size_t)32 * tc + (size_t)2 * p.W * rc) * p.lddz + m0);
        const char* xb = reinterpret_cast<const char*>(p.x + (size_t)16 * tc * p.ldx + n0);
#pragma unroll
        for (int k = 0; k < PPW; ++k) src_next[k] = reinterpret_cast<const float*>(((wv + 4 * k) < NPA ? dzb : xb) + doff[k]);
    };
    auto issue_tile = [&](int t, int slot) {
        sources(t, t / tpr);
#pragma unroll
        for (int k = 0; k < PPW; ++k)
            __builtin_amdgcn_global_load_lds(src_next[k], (lds_void_t*)(smem + slot * SLOTF + (wv + 4 * k) * 256), 16, 0, 0);
    };

    f32x16 acc[4][TM][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][a][b][r] = 0.f;

    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_t*)smem;
    const unsigned a_lane = lds0 + 4u * (2 * lh * CTM + 32 * TM * mi + li);
    const unsigned b_lane = lds0 + NPA * 1024 + 4u * (lh * CTN + 64 * ni + li);

    int tile = split;
    issue_tile(tile, 0);
    issue_tile(tile + p.splits, 1);
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(PPW) : "memory");
    int slot = 0;
    // look-ahead tile (two splits on), advanced incrementally: no division in the loop
    const int dtx = p.splits % tpr, drow = p.splits / tpr;
    int t2 = tile + 2 * p.splits, tx2 = t2 % tpr, row2 = t2 / tpr;
    for (; tile < p.n_tiles; tile += p.splits) {
        const unsigned ab = a_lane + (unsigned)slot * SLOTB, bb = b_lane + (unsigned)slot * SLOTB;
        float* const dst = smem + ((slot + 2) % 3) * SLOTF;
        float fa[2][4][TM], fb[2][2];
        // operand reads of k-step 0 (exposed once per tile), then 8 steps; reads of step s+1 ride behind step s's MFMAs
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int a = 0; a < TM; ++a) CW_RD32(fa[0][t][a], ab, (((t >> 1) * 32 + (t & 1)) * CTM + 32 * a) * 4);
        CW_RD32(fb[0][0], bb, 0); CW_RD32(fb[0][1], bb, 128);
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            const int cur = st & 1, nx = cur ^ 1;
            if (TM == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[cur][0][0]), "+v"(fa[cur][1][0]), "+v"(fa[cur][2][0]), "+v"(fa[cur][3][0]),
                                       "+v"(fa[cur][0][TM - 1]), "+v"(fa[cur][1][TM - 1]), "+v"(fa[cur][2][TM - 1]), "+v"(fa[cur][3][TM - 1]),
                                       "+v"(fb[cur][0]), "+v"(fb[cur][1]));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[cur][0][0]), "+v"(fa[cur][1][0]), "+v"(fa[cur][2][0]), "+v"(fa[cur][3][0]),
                              "+v"(fb[cur][0]), "+v"(fb[cur][1]));
#pragma unroll
            for (int m = 0; m < NMF; ++m) {
                const int t = m / (2 * TM), a = (m / 2) % TM, b = m & 1;
                asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[t][a][b]) : "v"(fa[cur][t][a]), "v"(fb[cur][b]) : "memory");
                if (st == 0 && m == 0) {
                    sources(t2, row2);
                    t2 += p.splits; tx2 += dtx; row2 += drow;
                    if (tx2 >= tpr) { tx2 -= tpr; ++row2; }
                }
                if (st + 1 < 8) {
                    if (m < 4 * TM) {
                        const int t2 = m / TM, a2 = m % TM;
                        CW_RD32(fa[nx][t2][a2], ab, (((t2 >> 1) * 32 + (t2 & 1) + 4 * (st + 1)) * CTM + 32 * a2) * 4);
                    } else if (m < 4 * TM + 2) {
                        CW_RD32(fb[nx][m - 4 * TM], bb, (2 * (st + 1) * CTN + 32 * (m - 4 * TM)) * 4);
                    }
                }
                // DMAs of the tile after next: PPW pieces over the 8 steps (at most 2 per step, behind MFMAs NMF-4 and NMF-2)
                if (m == NMF - 4 && 2 * st < PPW) __builtin_amdgcn_global_load_lds(src_next[2 * st], (lds_void_t*)(dst + (wv + 4 * (2 * st)) * 256), 16, 0, 0);
                if (m == NMF - 2 && 2 * st + 1 < PPW) __builtin_amdgcn_global_load_lds(src_next[2 * st + 1], (lds_void_t*)(dst + (wv + 4 * (2 * st + 1)) * 256), 16, 0, 0);
            }
        }
        // the next tile has landed (only the newest batch may be in flight; loads retire in order); everyone is done with `slot`
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(PPW) : "memory");
        slot = (slot + 1) % 3;
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");

    // partial block -> workspace [split][tap][Cout][Cin]
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                    float e;
                    asm("v_accvgpr_read_b32 %0, %1" : "=v"(e) : "a"(acc[t][a][b][r]));
                    p.ws[(((size_t)split * 4 + t) * p.Cout + m0 + 32 * TM * mi + 32 * a + row) * p.Cin + n0 + 64 * ni + 32 * b + li] = e;
                }
}

__global__ __launch_bounds__(256, 1) void convt_wgrad_wide_kernel_128(ConvtWgArgs p) { convt_wgrad_wide_body<2>(p); }
__global__ __launch_bounds__(256, 1) void convt_wgrad_wide_kernel_64(ConvtWgArgs p) { convt_wgrad_wide_body<1>(p); }

// dw = sum over the split partials, in a fixed order: a block covers 256/SL consecutive float4 outputs x SL slices of the split
// range (SL a power of two <= 16, chosen on the host so that small weight tensors with many splits still fill the chip); each
// thread sums its slice front to back, thread 0 of each output adds the slice sums front to back.
__global__ __launch_bounds__(256) void convt_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, long n4, int splits, int sl) {
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

int convt_wg_splits(int N, int H, int W, int Cin, int Cout) {
    const int ctm = Cout % 128 == 0 ? 128 : 64;
    const int mn = (Cout / ctm) * (Cin / 128);
    const long tiles = (long)N * H * (W / 16);
    long s = convt_cus() / mn; if (s < 1) s = 1; if (s > tiles) s = tiles;
    return (int)s;
}

}  // namespace

