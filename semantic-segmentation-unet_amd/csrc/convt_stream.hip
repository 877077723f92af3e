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

extern "C" int unet_convT2x2_fwd_stream_stats_rows(int N, int H, int W, int Cin, int Cout);

static int convt_fwd_stream_launch(const float* x, int ldx, const float* w, const float* bias, float* out, int ldo,
                                   int N, int H, int W, int Cin, int Cout, float* stat_part, size_t stat_bytes, void* stream) {
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
    const unsigned grid = (unsigned)(tiles < convt_cus() ? tiles : convt_cus());
    hipStream_t st = (hipStream_t)stream;
    if (stat_part) {
        const int rows = unet_convT2x2_fwd_stream_stats_rows(N, H, W, Cin, Cout);
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
extern "C" int unet_convT2x2_fwd_stream_stats_rows(int N, int H, int W, int Cin, int Cout) {
    if (!unet_convT2x2_fwd_stream_supported(N, H, W, Cin, Cout)) return 0;
    const bool wide = Cout % 128 == 0;
    const long tiles = ((long)N * H * W / (wide ? 128 : 256)) * (Cout / (wide ? 128 : 64));
    const long grid = tiles < convt_cus() ? tiles : convt_cus();
    const int nct = Cout / (wide ? 128 : 64);
    return grid % nct == 0 ? (int)((grid / nct) * (wide ? 2 : 4)) : 0;
}

extern "C" int unet_convT2x2_fwd_stream(const float* x, int ldx, const float* w, const float* bias, float* out, int ldo,
                                        int N, int H, int W, int Cin, int Cout, void* stream) {
    return convt_fwd_stream_launch(x, ldx, w, bias, out, ldo, N, H, W, Cin, Cout, nullptr, 0, stream);
}

// + BatchNorm sums of the output (layout and finalize as for unet_conv3x3_fwd_winograd_fused_stats)
extern "C" int unet_convT2x2_fwd_stream_stats(const float* x, int ldx, const float* w, const float* bias, float* out, int ldo,
                                              int N, int H, int W, int Cin, int Cout, float* stat_part, size_t stat_bytes, void* stream) {
    UNET_CHECK_ARG(stat_part);
    return convt_fwd_stream_launch(x, ldx, w, bias, out, ldo, N, H, W, Cin, Cout, stat_part, stat_bytes, stream);
}
