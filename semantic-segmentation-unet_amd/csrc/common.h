// Shared helpers for the gfx950 U-Net kernels.  All tensors are fp32, activations NHWC with an explicit
// channel stride `ld` (elements between consecutive pixels) so producers can write into / consumers can read
// from channel slices of a wider buffer (the zero-copy concat of UNet/model.py:55-58).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define UNET_OK 0
#define UNET_EINVAL (-1)      // bad argument (shape / alignment / null pointer)
#define UNET_ENOSPC (-2)      // workspace too small

#define UNET_CHECK_ARG(cond) do { if (!(cond)) return UNET_EINVAL; } while (0)
#define UNET_LAUNCH_STATUS() ((int)hipGetLastError())

// Streaming stores (nt policy).  UNET_NT is the compile-time set of kernel families that use them (bits below; scripts/build_variant_all.sh
// varies it).  Measured, round 3: write-only kernels launched back to back gain a third (class-map input gradient, 268 MB bf16: 0.099 -> 0.066 ms;
// fp32: 0.144 -> 0.117).  Inside the training step (same box, alternating runs of 60 steps, ms per bf16 step): FIRST only 13.03 / 13.09 / 13.07,
// FIRST + CONV16 12.98 / 13.00 / 12.99, + BN16 13.02 / 13.00 / 13.00; a first pass had BN (generic BatchNorm kernels) at +-0.  So the write-only
// kernels and the bf16 conv epilogues stream; the BatchNorm passes, whose output the next kernel reads back at once, do not.  The fp32 step
// loses with streaming outputs: fused Winograd forward / data gradient 44.05 -> 45.2 ms, transposed-conv forward 44.05 -> 44.3 (three
// alternating runs each) -- their consumers (BatchNorm-apply on load, the next layer) find the tensor in the Infinity Cache today.
// Streaming LOADS where a kernel reads a tensor for the last time: the packed bf16 BatchNorm backward apply kernels (dy, r), so that the dz they
// write -- read next by the data- and weight-gradient kernels -- keeps the cache: 12.84 -> 12.70 / 12.84 -> 12.77 / 12.83 -> 12.69 ms per bf16
// step.  The same in the generic kernels of the fp32 step LOSES (44.07 -> 44.27, 44.09 -> 44.18, 44.12 -> 44.26): there r is read again by the
// consumer layer's weight gradient (BatchNorm-apply on load), so it is not a last use; not kept.
// LDS-DMA SOURCES with the streaming policy (diagnostic build, same A/B): the conv patch 12.78 -> 14.05 ms, the weight gradient's input rows
// -> 13.08, its dz rows -> 13.05: every staged tensor is re-read inside its kernel by the other channel tiles and wants the L2; not kept.
#ifndef UNET_NT
#define UNET_NT 201
#endif
#define UNET_NT_FIRST 1       /* first layer forward, class-map input gradient */
#define UNET_NT_BN 2          /* BatchNorm apply / backward-apply (generic kernels) */
#define UNET_NT_BN16 4        /* packed bf16 BatchNorm backward kernels */
#define UNET_NT_CONV16 8      /* bf16 3x3 / transposed conv epilogues */
#define UNET_NT_LDBN16 64     /* last-use LOADS of dy and r in the packed bf16 BatchNorm backward apply kernels */
#define UNET_NT_LDAPPLY 128   /* the conv output r read by BatchNorm apply: its next reader is the backward pass (12.74/12.78/12.77 -> 12.71/12.74/12.71 ms
                               * per bf16 step, fp32 unchanged within 0.05) */
#define UNET_NT_AUX(bit) ((UNET_NT & (bit)) ? 2 : 0)             /* cache-policy operand of the raw buffer stores: bit 1 = nt */
typedef unsigned unet_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned unet_u32x2 __attribute__((ext_vector_type(2)));
template <int BIT, class T> __device__ __forceinline__ void unet_store(T* p, T v) {
    if constexpr ((UNET_NT & BIT) != 0) __builtin_nontemporal_store(v, p); else *p = v;
}

static inline bool unet_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int unet_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
// Workgroup slots a persistent kernel sizes its grid by: one per CU, or the caller's max_workgroups when that lies in [32, cus).  A
// data-parallel caller passes ~224 so that the collective's kernels (RCCL runs 16-32 workgroups of its own) find ~4 free CUs per XCD
// instead of waiting for a resident workgroup to drain its share of the tiles (parallel.py; scripts/overlap_probe.py measures it).
static inline int unet_grid_slots(int cus, int max_workgroups) { return (max_workgroups >= 32 && max_workgroups < cus) ? max_workgroups : cus; }

// counter-based RNG for dropout: one 32-bit hash per element, keep = top bit.  The same (seed, index)
// regenerates the mask in backward, so no mask tensor is stored.
__device__ __forceinline__ uint32_t unet_hash32(uint32_t seed, uint64_t idx) {
    uint64_t z = idx + 0x9E3779B97F4A7C15ull * (uint64_t)(seed + 1u);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (uint32_t)(z >> 32);
}

// conv_igemm.hip: `planes` independent fp32-MFMA GEMMs out[z][t][n] = sum_k x[z][t][k] * w[z][k][n] (T % 32 == 0)
int unet_igemm_batched_planes(const float* x, const float* w, float* out, long T, int K, int N, int planes, hipStream_t st);

