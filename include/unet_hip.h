/* libunet_hip.so -- C ABI of the MI355X (gfx950) U-Net train / inference hot path.
 *
 * Drop-in boundary for the ONE hot path of usnistgov/semantic-segmentation-unet: the body of class `UNet`
 * (reference UNet/model.py:19-256).  The reference has no FFI of its own for this path (its arithmetic is implicit in
 * TensorFlow/Keras layer calls); each entry point below cites the reference call site whose arithmetic it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless it says "host"; nothing is allocated, freed or retained;
 *   - activations are fp32 NHWC; `ld*` = channel stride in elements between consecutive pixels, so a tensor may be a
 *     channel slice of a wider buffer (the zero-copy concat of UNet/model.py:55-58); P = N*H*W pixels;
 *   - weights keep the Keras layouts: Conv2D kernel HWIO [kh][kw][Cin][Cout]; Conv2DTranspose kernel [kh][kw][Cout][Cin];
 *   - `stream` is a hipStream_t (NULL = default stream); all work is enqueued asynchronously;
 *   - return value: 0 ok, UNET_EINVAL (-1) bad argument, UNET_ENOSPC (-2) workspace too small, >0 a hipError_t;
 *   - no global mutable state and no environment variables: every option is an argument; the library caches only immutable device
 *     properties (CU count, occupancy of its own kernels), so calls are safe from different streams/threads;
 *   - `*_workspace()` return the scratch bytes the matching call needs for the same shape arguments.
 */
#ifndef UNET_HIP_H
#define UNET_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define UNET_OK 0
#define UNET_EINVAL (-1)
#define UNET_ENOSPC (-2)

/* Bumped whenever an exported signature changes; a loader must refuse a library whose unet_hip_abi_version() differs
 * (8: round 6 -- + unet_bn_finalize_apply_any (statistics finalize merged into the apply launch); the bf16 3x3 kernels start their accumulators at the bias;
 *  7: round 5 -- no signature changed, but the operand unet_winograd_weight_transform_x6 / _fold_x6 write is laid out for the round-5 kernel
 *  (MFMA A-operand order per 64-channel tile, chunk and point row): an operand and the conv entry point that reads it must come from one library;
 *  6: round 4 -- unet_convT2x2_*_x6; 5: round 4 -- the `_wg` (max_workgroups) entry points of every persistent kernel, unet_standin_collective; 4: round 4 -- the BF16x6 Winograd route: unet_*_x6; 3: + deferred bias gradient of unet_bn_bwd_any; 2: round 3 -- max_workgroups of the fused Winograd weight gradient; 1: rounds 1-2). */
#define UNET_HIP_ABI_VERSION 9
int unet_hip_abi_version(void);

/* ---- Conv2D(3x3, 'same', relu) of UNet._conv_layer, UNet/model.py:28-35 (18 instances, :88-134) ------------------ */
/* fp32 matrix-core implicit GEMM; needs Cin % 32 == 0 and Cout % 64 == 0 (unet_conv3x3_mfma_supported). */
int unet_conv3x3_mfma_supported(int Cin, int Cout);
int unet_conv3x3_fwd_mfma(const float* x, int ldx, const float* w, const float* bias, float* out, int ldo,
                          int N, int H, int W, int Cin, int Cout, int relu, void* stream);
/* derived backward of the same layer (tf.GradientTape, UNet/model.py:219): data gradient and weight gradient */
int unet_conv3x3_dgrad_mfma(const float* dz, int lddz, const float* w, float* dx, int lddx,
                            int N, int H, int W, int Cin, int Cout, void* stream);
size_t unet_conv3x3_wgrad_mfma_workspace(int N, int H, int W, int Cin, int Cout);
int unet_conv3x3_wgrad_mfma(const float* xin, int ldx, const float* dz, int lddz, float* dw,
                            int N, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);
/* ---- Winograd F(2x2,3x3) for the wide 3x3 layers: exact fp32, 2.25x fewer matrix multiplies -------------------------------------
 * Fully fused kernels (raw patch -> LDS, transforms in-kernel, 16-point MFMA, output transform in the epilogue).
 * Weight operands: unet_winograd_weight_transform(w, mode 2 forward / 3 data gradient) -> Uc (16*Cin*Cout floats), or every
 * layer's pair in one launch with _batch.  jobs: device array of njobs x 6 int64 = { w, Uc_fwd, Uc_dgrad, Cin | Cout << 32,
 * first_block, 0 }, first_block = running sum of ceil(Cin*Cout / 2048); total_blocks = that sum.  Cin, Cout multiples of 8. */
int unet_winograd_weight_transform(const float* w, float* U, int Cin, int Cout, int mode, void* stream);
int unet_winograd_weight_transform_batch(const void* jobs, int njobs, int total_blocks, void* stream);
/* BatchNorm-apply on load for the forward kernel (UNet/model.py:36 feeding :30 without materialising the BatchNorm output): from the
 * layer's kernel w, its bias and the PRODUCER's BatchNorm scale / shift (Cin floats each) make Uc = scale . transform(w), bias_out =
 * bias + sum_taps shift . w and pad[c] = -shift[c] / scale[c] (Cin + 8 floats): the forward kernel then reads the producer's conv
 * output r directly, with pad as the value of positions outside the image, and computes conv(zero-padded BatchNorm output). */
size_t unet_winograd_weight_fold_workspace(int Cin, int Cout);
int unet_winograd_weight_fold(const float* w, const float* bias, const float* scale, const float* shift, float* Uc, float* bias_out,
                              float* pad, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);
/* forward: H, W even, Cin % 8 == 0, Cout % 64 == 0.  pad nullable (zero padding).  stat_part nullable: BatchNorm statistics of the
 * output in the same kernel (rows > 0 when the persistent kernel takes the shape; (Cout/64) * rows * 128 floats; finish with
 * unet_bn_train_finalize_partials) */
int unet_conv3x3_fwd_winograd_fused_stats_rows(int N, int H, int W, int Cin, int Cout);
int unet_conv3x3_fwd_winograd_fused(const float* x, int ldx, const float* pad, const float* Uc, const float* bias, float* out, int ldo,
                                    int N, int H, int W, int Cin, int Cout, int relu, float* stat_part, size_t stat_bytes, void* stream);
/* data gradient; with r_prev / stat_part (both or neither) also the BatchNorm-backward sums (sum dy, sum dy*r) of the layer that
 * produced this layer's input: dx channels [c0, c1) are that layer's dy, r_prev its saved activation; rows =
 * unet_conv3x3_fwd_winograd_fused_stats_rows(N,H,W,Cout,Cin); the sums replace the reduction pass of unet_bn_bwd
 * (unet_bn_bwd_from_partials, part = stat_part + (c0/64)*rows*128) */
int unet_conv3x3_dgrad_winograd_fused(const float* dz, int lddz, const float* Ucd, float* dx, int lddx,
                                      int N, int H, int W, int Cin, int Cout, const float* r_prev, int ldr, int c0, int c1,
                                      float* stat_part, size_t stat_bytes, void* stream);
/* fused Winograd weight gradient: raw rows through LDS, per-lane transforms in registers, G^T dU G in the epilogue;
 * needs H, W even and Cin, Cout multiples of 64 */
int unet_winograd_wgrad_fused_supported(int N, int H, int W, int Cin, int Cout);
/* max_workgroups: cap on the persistent grid, 0 = one workgroup per CU.  A data-parallel caller passes ~224 so that the collective's
 * kernels (tf.distribute's all-reduce inside apply_gradients, UNet/model.py:223 under UNet/train.py:57-61) find free CUs while this
 * chip-filling kernel runs; the same value must be given to the _workspace() query. */
size_t unet_conv3x3_wgrad_winograd_fused_workspace(int N, int H, int W, int Cin, int Cout, int max_workgroups);
int unet_conv3x3_wgrad_winograd_fused(const float* xin, int ldx, const float* dz, int lddz, float* dw,
                                      int N, int H, int W, int Cin, int Cout, int max_workgroups, void* ws, size_t ws_bytes, void* stream);
/* weight gradient of a layer whose input was read through BatchNorm-apply on load (x = scale . r + shift inside the image): dw holds
 * any wgrad kernel's result on the RAW r; in place dw = scale[ci] * dw + shift[ci] * S[tap][co], S = sum of dz over the pixels whose
 * tap lies inside the image (border sums of dz; total = column sums of dz = the bias gradient). */
size_t unet_conv3x3_wgrad_fold_fix_workspace(int Cout);
int unet_conv3x3_wgrad_fold_fix(float* dw, const float* scale, const float* shift, const float* dz, int lddz, const float* total,
                                int N, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);
/* ---- the same Winograd F(2x2,3x3) layers with the 16 point products on the BF16 matrix pipe at fp32 grade ("BF16x6") ---------------
 * A second route to the fp32 result of UNet._conv_layer (UNet/model.py:28-35), beside unet_conv3x3_*_winograd_fused: every fp32 operand of a
 * Winograd-domain product is split EXACTLY into three bf16 pieces (h + m + l), the six piece products hh, hm, mh, hl, lh, mm are exact
 * and accumulate in fp32 (v_mfma_f32_32x32x16_bf16); the three dropped products sum to at most 2^-21, on average 2^-24.5 of the product, i.e. one fp32 multiply's
 * rounding, so the error against an fp64 evaluation equals that of the fp32 matrix instruction (tests/test_gpu_x6.py asserts <= 1.25 x).
 * The transforms stay fp32.  Shapes: H, W even, reduce channels % 32 == 0 and >= 64, output channels % 64 == 0 (unet_winograd_x6_supported).
 * Weight operands (16 * 3 * Cin * Cout bf16 = unet_winograd_x6_weight_bytes; opaque: made and read by the same library): _x6(w, mode 0 forward / 1 data gradient), every layer and
 * direction in one launch with _x6_batch (jobs: njobs x 6 int64 = { w, U6, Cin | Cout << 32, first_block, mode, 0 }, first_block = running
 * sum of ceil(Cin * Cout / 2048)), or _fold_x6 = unet_winograd_weight_fold for this route.  The two conv entry points take the arguments of
 * unet_conv3x3_fwd_winograd_fused / unet_conv3x3_dgrad_winograd_fused with U6 in place of Uc; stat_part rows are the same function. */
int unet_winograd_x6_supported(int N, int H, int W, int K, int Nout);
size_t unet_winograd_x6_weight_bytes(int Cin, int Cout);
int unet_winograd_weight_transform_x6(const float* w, void* U6, int Cin, int Cout, int mode, void* stream);
int unet_winograd_weight_transform_x6_batch(const void* jobs, int njobs, int total_blocks, void* stream);
int unet_winograd_weight_fold_x6(const float* w, const float* bias, const float* scale, const float* shift, void* U6, float* bias_out,
                                 float* pad, int Cin, int Cout, void* stream);
int unet_conv3x3_fwd_winograd_x6(const float* x, int ldx, const float* pad, const void* U6, const float* bias, float* out, int ldo,
                                 int N, int H, int W, int Cin, int Cout, int relu, float* stat_part, size_t stat_bytes, void* stream);
int unet_conv3x3_dgrad_winograd_x6(const float* dz, int lddz, const void* U6d, float* dx, int lddx,
                                   int N, int H, int W, int Cin, int Cout, const float* r_prev, int ldr, int c0, int c1,
                                   float* stat_part, size_t stat_bytes, void* stream);
/* ---- bf16 matrix-core 3x3 convolution (BASELINE config 4: bf16 forward/backward, fp32 master weights; the reference keeps
 * its mixed-precision policy commented out, UNet/train.py:52-54) ------------------------------------------------------
 * Operands rounded to bf16 (nearest-even), exact products, fp32 accumulation.
 * Cin % 64 == 0, Cout % 64 == 0, tensors < 2 GiB.  Weights are packed on the device once per step from the fp32 master copy
 * (HWIO, UNet/model.py:31): mode 0 for the forward, mode 1 for the data gradient. */
int unet_conv3x3_bf16_supported(int N, int H, int W, int Cin, int Cout);
size_t unet_conv3x3_bf16_packed_bytes(int Cin, int Cout);
int unet_conv3x3_bf16_pack_weights(const float* w, void* packed, int Cin, int Cout, int mode, void* stream);
/* One entry point per direction, every option a parameter:
 *  - x_bf16 / dz_bf16: that operand is STORED as bf16 (leading dimension in elements).  A producer (unet_bn_apply_any,
 *    unet_bn_bwd_any) rounds exactly as these kernels' staging would, so bf16 storage of an operand is bit-identical to fp32 storage;
 *  - in_scale / in_shift (forward, nullable): BatchNorm-apply on load -- x is the producer layer's conv output r and the operand is
 *    bf16(in_scale[c] * r + in_shift[c]) inside the image, exactly 0 at the padding: bit-identical to applying the BatchNorm in a pass
 *    of its own (UNet/model.py:36 then :30), without that pass;
 *  - out_bf16 / dx_bf16 / r_bf16: the OUTPUT (resp. the producer's saved activation) is a bf16 tensor -- Keras mixed_bfloat16
 *    semantics (activations and their gradients bf16, BatchNorm arithmetic fp32; the fused sums are taken before the rounding);
 *  - stat_part (nullable): one row of partial sums per 16 x 32 pixel tile, [C/64][rows][64][2]: forward (sum y, sum y^2) of the output
 *    for unet_bn_train_finalize_partials; data gradient, with r_prev = the saved activation of the layer whose dy is dx[..., c0:c1):
 *    (sum dx, sum dx * r_prev) for unet_bn_bwd_any / unet_bn_bwd_from_partials. */
int unet_conv3x3_bf16_stats_rows(int N, int H, int W, int Cin, int Cout);
int unet_conv3x3_fwd_bf16(const void* x, int ldx, int x_bf16, const float* in_scale, const float* in_shift, const void* wp,
                          const float* bias, void* out, int ldo, int out_bf16,
                          int N, int H, int W, int Cin, int Cout, int relu, float* stat_part, size_t stat_bytes, void* stream);
int unet_conv3x3_dgrad_bf16(const void* dz, int lddz, int dz_bf16, const void* wpd, void* dx, int lddx, int dx_bf16,
                            int N, int H, int W, int Cin, int Cout, const void* r_prev, int ldr, int r_bf16, int c0, int c1,
                            float* stat_part, size_t stat_bytes, void* stream);
/* weight gradient in the same arithmetic (both operands rounded to bf16, fp32 accumulation, split partial sums added in a
 * fixed order): dw[a,b,ci,co] = sum xin[n,y+a-1,x+b-1,ci] * dz[n,y,x,co] */
int unet_conv3x3_wgrad_bf16_supported(int N, int H, int W, int Cin, int Cout);
size_t unet_conv3x3_wgrad_bf16_workspace(int N, int H, int W, int Cin, int Cout);
int unet_conv3x3_wgrad_bf16(const void* xin, int ldx, int x_bf16, const void* dz, int lddz, int dz_bf16, float* dw,
                            int N, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);
/* first layer (Cin = number_channels, UNet/model.py:88): any Cin, Cout/4 a power of two <= 256; Cin <= 4 with Cout == 64 runs on the fp32
 * matrix cores (window gathered from global memory), the rest as a VALU stencil */
int unet_conv3x3_fwd_direct(const float* x, int ldx, const float* w, const float* bias, float* out, int ldo,
                            int N, int H, int W, int Cin, int Cout, int relu, void* stream);
/* + BatchNorm sums of the output in the same kernel (Cin <= 4 and Cout == 64, or Cin <= 4, W % 4 == 0, Cout % 64 == 0): rows of partial
 * sums it writes per 64-channel block, 0 when no such kernel applies */
int unet_conv3x3_fwd_direct_stats_rows(int N, int H, int W, int Cin, int Cout);
/* out_bf16: the output tensor is stored as bf16 (ldo in elements; the sums are those of the fp32 values) */
int unet_conv3x3_fwd_direct_stats(const float* x, int ldx, const float* w, const float* bias, void* out, int ldo, int out_bf16,
                                  int N, int H, int W, int Cin, int Cout, int relu, float* stat_part, size_t stat_bytes, void* stream);
size_t unet_conv3x3_wgrad_direct_workspace(int N, int H, int W, int Cin, int Cout);
int unet_conv3x3_wgrad_direct(const float* xin, int ldx, const void* dz, int lddz, int dz_bf16, float* dw,
                              int N, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);

/* ---- Conv2D(1x1, relu) class map, UNet/model.py:136 (w is [Cin][Cout]) ------------------------------------------- */
/* x_bf16 / dx_bf16: the Cin-channel side (the last decoder layer's BatchNorm output / its gradient) is stored as bf16, leading
 * dimension in elements; the arithmetic and the class-map side stay fp32 */
int unet_conv1x1_fwd(const void* x, int ldx, int x_bf16, const float* w, const float* bias, float* out, int ldo,
                     long P, int Cin, int Cout, int relu, void* stream);
int unet_conv1x1_dgrad(const float* dz, int lddz, const float* w, void* dx, int lddx, int dx_bf16, long P, int Cin, int Cout, void* stream);
size_t unet_conv1x1_wgrad_workspace(long P, int Cin, int Cout);
int unet_conv1x1_wgrad(const void* xin, int ldx, int x_bf16, const float* dz, int lddz, float* dw,
                       long P, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);

/* ---- Conv2DTranspose(2x2, stride 2) of UNet._deconv_layer, UNet/model.py:39-46 (:116,121,126,131) ---------------- */
/* N,H,W = INPUT dims of the layer; output is [N,2H,2W,Cout].  Needs Cin % 64 == 0 and Cout % 64 == 0. */
int unet_convT2x2_fwd(const float* x, int ldx, const float* w, const float* bias, float* out, int ldo,
                      int N, int H, int W, int Cin, int Cout, void* stream);
/* same result from the persistent stream kernel (convt_stream.hip): Cin % 32 == 0, Cout % 64 == 0, N*H*W a multiple of the
 * 128 (Cout % 128 == 0) or 256 pixel tile, ldo % 4 == 0 */
int unet_convT2x2_fwd_stream_supported(int N, int H, int W, int Cin, int Cout);
int unet_convT2x2_fwd_stream(const float* x, int ldx, const float* w, const float* bias, float* out, int ldo,
                             int N, int H, int W, int Cin, int Cout, void* stream);
/* + BatchNorm sums of the output (UNet/model.py:47); rows / layout / finalize as for unet_conv3x3_fwd_winograd_fused_stats */
int unet_convT2x2_fwd_stream_stats_rows(int N, int H, int W, int Cin, int Cout);
int unet_convT2x2_fwd_stream_stats(const float* x, int ldx, const float* w, const float* bias, float* out, int ldo,
                                   int N, int H, int W, int Cin, int Cout, float* stat_part, size_t stat_bytes, void* stream);
int unet_convT2x2_dgrad(const float* dz, int lddz, const float* w, float* dx, int lddx,
                        int N, int H, int W, int Cin, int Cout, void* stream);
/* forward / data gradient as GEMMs on the bf16 matrix pipe at fp32 grade (BF16x6, csrc/convt_x6.hip; the arithmetic of the unet_*_x6
 * 3x3 kernels above): need N*H*W % 128 == 0, Cin % 128 == 0, Cout % 64 == 0 (unet_convT2x2_x6_supported).  W6 = the layer's kernel as
 * three bf16 pieces per weight in the GEMM's operand layout, unet_convT2x2_weight_transform_x6 mode 0 (forward) / 1 (data gradient),
 * unet_convT2x2_x6_weight_bytes each, once per optimizer step.  Forward: stat_part nullable -- BatchNorm sums of the output,
 * (Cout/64) * rows * 128 floats with rows = unet_convT2x2_x6_stats_rows (one per 128-pixel tile and tap), for
 * unet_bn_train_finalize_partials.  Forward and data gradient launch one workgroup per tile (no max_workgroups form); the weight
 * gradient's grid is sized by the CU count and has one (unet_convT2x2_wgrad_x6_wg below). */
int unet_convT2x2_x6_supported(int N, int H, int W, int Cin, int Cout);
size_t unet_convT2x2_x6_weight_bytes(int Cin, int Cout);
int unet_convT2x2_weight_transform_x6(const float* w, void* W6, int Cin, int Cout, int mode, void* stream);
/* jobs: device array of njobs x 6 int64 = { w, W6, Cin | Cout << 32, first_block, mode, 0 }, first_block = running sum of ceil(4*Cin*Cout/8 / 256) */
int unet_convT2x2_weight_transform_x6_batch(const void* jobs, int njobs, int total_blocks, void* stream);
int unet_convT2x2_x6_stats_rows(int N, int H, int W, int Cin, int Cout);
int unet_convT2x2_fwd_x6(const float* x, int ldx, const void* W6, const float* bias, float* out, int ldo,
                         int N, int H, int W, int Cin, int Cout, float* stat_part, size_t stat_bytes, void* stream);
int unet_convT2x2_dgrad_x6(const float* dz, int lddz, const void* W6d, float* dx, int lddx,
                           int N, int H, int W, int Cin, int Cout, void* stream);
/* ... with the BatchNorm-backward sums of the layer that produced x (UNet/model.py:36 under the tape of :216-219): dx is that layer's dy, r_prev
 * [N*H*W][ldr] its saved activation; stat_part receives (sum dx, sum dx * r_prev) as (Cin/64) * rows * 128 floats, rows =
 * unet_convT2x2_x6_bnbwd_rows (one per 128-pixel tile) -- the `sums_part` of unet_bn_bwd_any, which then needs no reduction pass.
 * r_prev and stat_part both null = unet_convT2x2_dgrad_x6. */
int unet_convT2x2_x6_bnbwd_rows(int N, int H, int W, int Cin, int Cout);
int unet_convT2x2_dgrad_x6_sums(const float* dz, int lddz, const void* W6d, float* dx, int lddx,
                                int N, int H, int W, int Cin, int Cout, const float* r_prev, int ldr,
                                float* stat_part, size_t stat_bytes, void* stream);
/* weight gradient on the same arithmetic: both operands are activations, split into three bf16 pieces inside the kernel; split-K partials
 * in ws (unet_convT2x2_wgrad_x6_workspace bytes), reduced in split order */
size_t unet_convT2x2_wgrad_x6_workspace(int N, int H, int W, int Cin, int Cout);
int unet_convT2x2_wgrad_x6(const float* xin, int ldx, const float* dz, int lddz, float* dw,
                           int N, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);
size_t unet_convT2x2_wgrad_workspace(int N, int H, int W, int Cin, int Cout);
int unet_convT2x2_wgrad(const float* xin, int ldx, const float* dz, int lddz, float* dw,
                        int N, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);

/* bf16 matrix-core forms of the transposed-conv forward and data gradient (operands rounded to bf16, fp32 accumulation; Cin, Cout
 * multiples of 64).  Weights packed per step from the fp32 master kernel: mode 0 forward operand, mode 1 data-gradient operand.
 * x_bf16 / dz_bf16 / out_bf16 / dx_bf16 / r_bf16: that tensor is stored as bf16.  stat_part nullable: forward -- BatchNorm sums of the output, rows =
 * unet_convT2x2_bf16_stats_rows(..., 0) (one per input-pixel tile and tap); data gradient -- with r_prev, the BatchNorm-backward
 * sums (sum dx, sum dx * r_prev) of the layer that produced x, rows = unet_convT2x2_bf16_stats_rows(..., 1). */
/* every pack of a step in one launch: jobs[njobs][6] int64 = {fp32 kernel, forward operand, data-gradient operand,
 * Cin | Cout << 32, kind (0: 3x3 HWIO, 1: transposed conv [2][2][Cout][Cin]), first 256-thread block of the job};
 * a job needs ceil(taps * Cin * Cout / 8 / 256) blocks (taps = 9 / 4) */
int unet_bf16_pack_weights_batch(const void* jobs, int njobs, int total_blocks, void* stream);
int unet_convT2x2_bf16_supported(int N, int H, int W, int Cin, int Cout);
size_t unet_convT2x2_bf16_packed_bytes(int Cin, int Cout);
int unet_convT2x2_bf16_pack_weights(const float* w, void* packed, int Cin, int Cout, int mode, void* stream);
int unet_convT2x2_bf16_stats_rows(int N, int H, int W, int Cin, int Cout, int dgrad);
int unet_convT2x2_fwd_bf16(const void* x, int ldx, int x_bf16, const void* wp, const float* bias, void* out, int ldo, int out_bf16,
                           int N, int H, int W, int Cin, int Cout, float* stat_part, size_t stat_bytes, void* stream);
int unet_convT2x2_dgrad_bf16(const void* dz, int lddz, int dz_bf16, const void* wpd, void* dx, int lddx, int dx_bf16,
                             int N, int H, int W, int Cin, int Cout, const void* r_prev, int ldr, int r_bf16,
                             float* stat_part, size_t stat_bytes, void* stream);
/* weight gradient in the same arithmetic (additionally Cin % 128 == 0): dw[a,b,co,ci] = sum dz[n,2i+a,2j+b,co] * xin[n,i,j,ci] */
int unet_convT2x2_wgrad_bf16_supported(int N, int H, int W, int Cin, int Cout);
size_t unet_convT2x2_wgrad_bf16_workspace(int N, int H, int W, int Cin, int Cout);
int unet_convT2x2_wgrad_bf16(const void* xin, int ldx, int x_bf16, const void* dz, int lddz, int dz_bf16, float* dw,
                                int N, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);

/* ---- BatchNormalization(axis=1), UNet/model.py:36,47 ------------------------------------------------------------- */
size_t unet_bn_workspace(long P, int C);
/* training=True: batch mean / biased variance of r -> mean, invstd = 1/sqrt(var+eps), scale = gamma*invstd,
 * shift = beta - mean*scale; moving_mean/var (nullable pair) updated with `momentum`. */
int unet_bn_train_stats(const float* r, int ldr, long P, int C, const float* gamma, const float* beta,
                        float eps, float momentum, int unbiased_moving_var, float* moving_mean, float* moving_var,
                        float* mean, float* invstd, float* scale, float* shift, void* ws, size_t ws_bytes, void* stream);
/* the same outputs as unet_bn_train_stats from the partial sums written by unet_conv3x3_fwd_winograd_fused_stats */
int unet_bn_train_finalize_partials(const float* part, int rows, long P, int C, const float* gamma, const float* beta,
                                    float eps, float momentum, int unbiased_moving_var, float* moving_mean, float* moving_var,
                                    float* mean, float* invstd, float* scale, float* shift, void* stream);
/* training=False (UNet/model.py:239, UNet/inference.py:105,164): coefficients from the moving statistics */
int unet_bn_eval_coeffs(const float* gamma, const float* beta, const float* moving_mean, const float* moving_var,
                        float eps, int C, float* scale, float* shift, void* stream);
int unet_bn_apply(const float* r, int ldr, const float* scale, const float* shift, float* y, int ldy, long P, int C, void* stream);
/* the same plus the MaxPool2D(2) that follows (UNet/model.py:50-53) in one pass: pooled [N,H/2,W/2,C] and first-max indices */
int unet_bn_apply_maxpool(const float* r, int ldr, const float* scale, const float* shift, float* y, int ldy,
                          float* pooled, int ldp, uint8_t* idx, int N, int H, int W, int C, void* stream);
/* backward of [ReLU ->] BN: dz = relu'(r) * d r, plus dgamma, dbeta and dbias = column sums of dz */
int unet_bn_bwd(const float* dy, int lddy, const float* r, int ldr, const float* gamma, const float* mean,
                const float* invstd, long P, int C, int relu, float* dz, int lddz, float* dgamma, float* dbeta,
                float* dbias, void* ws, size_t ws_bytes, void* stream);

/* the layer's output also fed MaxPool2D(2): dy = dy_skip + unpool(pooled_dy, idx) formed on the fly (no pool-backward pass) */
int unet_bn_bwd_pooled(const float* dy_skip, int lddy, const float* pooled_dy, int ldp, const uint8_t* idx, int N, int H, int W,
                       const float* r, int ldr, const float* gamma, const float* mean, const float* invstd, int C, int relu,
                       float* dz, int lddz, float* dgamma, float* dbeta, float* dbias, void* ws, size_t ws_bytes, void* stream);
/* all three forms in one call (pooled_dy/idx nullable, part_sums nullable), dz optionally stored as bf16 (dz_bf16 != 0).
 * host_bias_rows (HOST pointer, nullable): when given, the bias gradient sum(dz) is not finalized here (dbias may be NULL): its
 * per-block partial sums stay in `ws`, *host_bias_rows receives their row count, and unet_bn_bwd_bias(ws, rows, C, dbias, stream)
 * finishes it -- on any stream ordered behind this call, typically beside the layer's weight gradient: nothing on the critical chain
 * BatchNorm backward -> data gradient needs the bias gradient.  `ws` must not be reused until then. */
int unet_bn_bwd_any(const void* dy, int lddy, const void* pooled_dy, int ldp, const uint8_t* idx, int N, int H, int W,
                    const void* r, int ldr, const float* gamma, const float* mean, const float* invstd, int C, int relu,
                    void* dz, int lddz, int dz_bf16, float* dgamma, float* dbeta, float* dbias, const float* part_sums, int rows,
                    void* ws, size_t ws_bytes, void* stream, int r_bf16, int dy_bf16, int pooled_dy_bf16, int* host_bias_rows);
int unet_bn_bwd_bias(const void* ws, int rows, int C, float* dbias, void* stream);
/* BatchNorm apply (+ the 2x2 max pool when pooled / idx are given) with the conv output it reads (r_bf16) and / or what it writes
 * (y_bf16: y and pooled) stored as bf16 */
int unet_bn_apply_any(const void* r, int ldr, int r_bf16, const float* scale, const float* shift, void* y, int ldy, int y_bf16,
                      void* pooled, int ldp, uint8_t* idx, int N, int H, int W, int C, void* stream);
/* unet_bn_train_finalize_partials + unet_bn_apply_any in ONE launch (round 6; UNet/model.py:36,47 -- the BatchNormalization of every layer
 * whose conv left fused partial sums): workgroup b of the apply grid first finalizes channels b, b + grid, ... with the finalize kernel's own
 * arithmetic (bit-identical mean / invstd / scale / shift / moving statistics), publishes them and every workgroup waits on `counter` until all
 * C channels are there.  `counter`: one device word owned by this call sequence (zero before the first call); `counter_target` = the value it
 * must reach = (sum of C over all earlier calls on this counter) + C modulo 2^32 -- the caller keeps the running sum, so the library holds no
 * state.  scale / shift are written, then read, by this launch. */
int unet_bn_finalize_apply_any(const float* part, int rows, const float* gamma, const float* beta, float eps, float momentum,
                               int unbiased_moving_var, float* moving_mean, float* moving_var, float* mean, float* invstd,
                               uint32_t* counter, uint32_t counter_target,
                               const void* r, int ldr, int r_bf16, float* scale, float* shift, void* y, int ldy, int y_bf16,
                               void* pooled, int ldp, uint8_t* idx, int N, int H, int W, int C, void* stream);
int unet_bn_bwd_from_partials(const float* dy, int lddy, const float* r, int ldr, const float* gamma, const float* mean,
                              const float* invstd, long P, int C, int relu, float* dz, int lddz, float* dgamma, float* dbeta,
                              float* dbias, const float* part_sums, int rows, void* ws, size_t ws_bytes, void* stream);

/* ---- MaxPool2D(2), UNet/model.py:50-53; Dropout(0.5), UNet/model.py:60-63 ---------------------------------------- */
/* t_bf16 (these three): the activation tensors are stored as bf16 (leading dimensions in elements, C % 4 == 0) */
int unet_maxpool2x2_fwd(const void* x, int ldx, void* y, int ldy, uint8_t* idx, int N, int H, int W, int C, int t_bf16, void* stream);
int unet_maxpool2x2_bwd(const void* dy, int lddy, const uint8_t* idx, void* dx, int lddx,
                        int N, int H, int W, int C, int accumulate, int t_bf16, void* stream);
/* Dropout(0.5), UNet/model.py:62: out = x * keep / (1 - rate); mask nullable -> counter hash of (seed, element) */
int unet_dropout(const void* x, int ldx, void* out, int ldo, long P, int C, const uint8_t* mask,
                 uint32_t seed, float rate, int t_bf16, void* stream);

/* ---- Softmax(axis=-1) + CategoricalCrossentropy + loss reduction + accuracy, UNet/model.py:142,77,211-215,226 ----- */
/* ce_clip_eps = 0: cross-entropy from the softmax's logits; > 0 (Keras: 1e-7): the probability path of
 * keras.backend.categorical_crossentropy (clip to [eps, 1-eps], no gradient outside the range) -- SURVEY.md 8(a) a8 (K) */
size_t unet_softmax_ce_workspace(long P);
int unet_softmax_ce(const float* logits, int ldz, const int* labels_onehot, float* prob, float* dlogits, int lddz,
                    long P, int K, float label_smoothing, float loss_scale, float grad_scale, float ce_clip_eps,
                    float* loss_out, float* correct_out, void* ws, size_t ws_bytes, void* stream);
/* np.argmax(softmax, axis=-1), UNet/inference.py:107,166 (first maximum wins) */
int unet_argmax(const float* p, int ldp, int* out, long P, int K, void* stream);

/* ---- eval-mode backward to the input image: the ERF probe of UNet.estimate_radius, UNet/model.py:165-202 ----------- */
/* dlogits_k = p_k (g_k - sum_j g_j p_j); BN with moving statistics is affine: dz = dy*scale (times the ReLU mask);
 * data gradient of the first 3x3 layer (Cin = number_channels), which the training path never needs. */
int unet_softmax_bwd(const float* prob, const float* dprob, float* dlogits, int lddz, long P, int K, void* stream);
int unet_bn_eval_bwd(const float* dy, int lddy, const float* r, int ldr, const float* scale, float* dz, int lddz,
                     long P, int C, int relu, void* stream);
int unet_conv3x3_dgrad_direct(const float* dz, int lddz, const float* w, float* dx, int lddx,
                              int N, int H, int W, int Cin, int Cout, void* stream);

/* ---- tf.keras.optimizers.Adam.apply_gradients, UNet/model.py:79,223; alpha = lr*sqrt(1-b2^t)/(1-b1^t) from host -- */
int unet_adam_keras(float* theta, const float* grad, float* m, float* v, long n, float alpha, float beta1,
                    float beta2, float eps, void* stream);

/* ---- reference input contract is NCHW (UNet/model.py:73; UNet/imagereader.py:298) -------------------------------- */
int unet_nchw_to_nhwc(const float* x, float* y, int N, int C, int H, int W, void* stream);
/* ---- augmentation stage, UNet/augment.py:19-174, batched over N images [N][H][W][C] fp32 ------------
 * warp: skimage rotate / warp(order 1, mode='reflect') -- output pixel (r, c) samples src at mats[n] . (c, r, 1), mats = N x 6
 *       floats (rows 0 and 1 of the output->input matrix); flips[n] bit 0 = left-right, bit 1 = up-down flip of the warped image
 *       (NULL = none); round_output = np.round of the result (the mask path).  src != dst.
 * gaussian_blur: scipy.ndimage.gaussian_filter(sigma, mode='reflect') along H, W and C (in place, tmp same size); sigmas[n] <= 0
 *       leaves image n unchanged.
 * minmax: per-image (min, max) -> minmax[N][2].
 * noise_intensity: img += field * (coef_noise[n] * range_n) + coef_add[n] * range_n, range_n = max - min from minmax. */
int unet_augment_warp(const float* src, float* dst, int N, int H, int W, int C, const float* mats, const int* flips,
                      int round_output, void* stream);
int unet_augment_gaussian_blur(float* img, float* tmp, int N, int H, int W, int C, const float* sigmas, void* stream);
size_t unet_augment_minmax_workspace(int N);
int unet_augment_minmax(const float* img, int N, long elems_per_image, float* minmax, void* ws, size_t ws_bytes, void* stream);
int unet_augment_noise_intensity(float* img, const float* field, int N, long elems_per_image, const float* minmax,
                                 const float* coef_noise, const float* coef_add, void* stream);
/* reader contract, UNet/imagereader.py:33-49,298-301: per-(image, channel) z-score ((x - mean) / std, only x - mean when
 * std <= 1) of img [N][H][W][C], written as the network's [N][C][H][W] input */
size_t unet_zscore_workspace(int N, int C);
int unet_zscore_nhwc_to_nchw(const float* img, float* out, int N, int H, int W, int C, void* ws, size_t ws_bytes, void* stream);
/* reader output contract, UNet/imagereader.py:302-312,353-355: class map [P] (uint8) -> one-hot int32 [P][K] on the device,
 * so the feed ships 1 byte per pixel instead of 4K; *out_of_range (may be NULL, else zeroed by the caller) counts labels >= K,
 * the condition the reference raises IndexError for */
int unet_labels_onehot(const uint8_t* classmap, int* onehot, long P, int K, unsigned* out_of_range, void* stream);

/* ---- workgroup-capped forms of every kernel whose grid is sized by the CU count ------------------------------------------------------
 * The reference's data-parallel step (tf.distribute.MirroredStrategy, UNet/train.py:57-61; the gradient all-reduce inside
 * apply_gradients, UNet/model.py:223) overlaps a collective with the backward pass.  RCCL's kernels need CUs of their own; a persistent
 * grid of one workgroup per CU leaves none until a workgroup has drained its share of the tiles.  Each function below is the function
 * of the same name without `_wg` plus max_workgroups (0, or out of [32, CUs): one workgroup per CU -- what the plain name passes); the
 * `_stats_rows_wg` / `_workspace_wg` query must be given the same value as the launch.  Kernels launched as one workgroup per TILE
 * (first / last layer, BatchNorm passes, max pool, the bf16 transposed-conv forward / data gradient, Adam) are not listed: their
 * workgroups retire within microseconds and the hardware scheduler hands the freed CUs to whichever queue is waiting.
 * (unet_conv3x3_wgrad_winograd_fused has carried the argument since ABI 2.) */
int unet_conv3x3_fwd_winograd_fused_stats_rows_wg(int N, int H, int W, int Cin, int Cout, int max_workgroups);
int unet_conv3x3_fwd_winograd_fused_wg(const float* x, int ldx, const float* pad, const float* Uc, const float* bias, float* out, int ldo,
                                       int N, int H, int W, int Cin, int Cout, int relu, float* stat_part, size_t stat_bytes,
                                       int max_workgroups, void* stream);
int unet_conv3x3_dgrad_winograd_fused_wg(const float* dz, int lddz, const float* Ucd, float* dx, int lddx,
                                         int N, int H, int W, int Cin, int Cout, const float* r_prev, int ldr, int c0, int c1,
                                         float* stat_part, size_t stat_bytes, int max_workgroups, void* stream);
int unet_conv3x3_fwd_winograd_x6_wg(const float* x, int ldx, const float* pad, const void* U6, const float* bias, float* out, int ldo,
                                    int N, int H, int W, int Cin, int Cout, int relu, float* stat_part, size_t stat_bytes,
                                    int max_workgroups, void* stream);
int unet_conv3x3_dgrad_winograd_x6_wg(const float* dz, int lddz, const void* U6d, float* dx, int lddx,
                                      int N, int H, int W, int Cin, int Cout, const float* r_prev, int ldr, int c0, int c1,
                                      float* stat_part, size_t stat_bytes, int max_workgroups, void* stream);
size_t unet_conv3x3_wgrad_mfma_workspace_wg(int N, int H, int W, int Cin, int Cout, int max_workgroups);
int unet_conv3x3_wgrad_mfma_wg(const float* xin, int ldx, const float* dz, int lddz, float* dw,
                               int N, int H, int W, int Cin, int Cout, int max_workgroups, void* ws, size_t ws_bytes, void* stream);
int unet_conv3x3_fwd_bf16_wg(const void* x, int ldx, int x_bf16, const float* in_scale, const float* in_shift, const void* wp,
                             const float* bias, void* out, int ldo, int out_bf16,
                             int N, int H, int W, int Cin, int Cout, int relu, float* stat_part, size_t stat_bytes,
                             int max_workgroups, void* stream);
int unet_conv3x3_dgrad_bf16_wg(const void* dz, int lddz, int dz_bf16, const void* wpd, void* dx, int lddx, int dx_bf16,
                               int N, int H, int W, int Cin, int Cout, const void* r_prev, int ldr, int r_bf16, int c0, int c1,
                               float* stat_part, size_t stat_bytes, int max_workgroups, void* stream);
size_t unet_conv3x3_wgrad_bf16_workspace_wg(int N, int H, int W, int Cin, int Cout, int max_workgroups);
int unet_conv3x3_wgrad_bf16_wg(const void* xin, int ldx, int x_bf16, const void* dz, int lddz, int dz_bf16, float* dw,
                               int N, int H, int W, int Cin, int Cout, int max_workgroups, void* ws, size_t ws_bytes, void* stream);
/* (stat_part nullable: the one function covers unet_convT2x2_fwd_stream and unet_convT2x2_fwd_stream_stats) */
int unet_convT2x2_fwd_stream_stats_rows_wg(int N, int H, int W, int Cin, int Cout, int max_workgroups);
int unet_convT2x2_fwd_stream_wg(const float* x, int ldx, const float* w, const float* bias, float* out, int ldo,
                                int N, int H, int W, int Cin, int Cout, float* stat_part, size_t stat_bytes, int max_workgroups, void* stream);
size_t unet_convT2x2_wgrad_workspace_wg(int N, int H, int W, int Cin, int Cout, int max_workgroups);
int unet_convT2x2_wgrad_wg(const float* xin, int ldx, const float* dz, int lddz, float* dw,
                           int N, int H, int W, int Cin, int Cout, int max_workgroups, void* ws, size_t ws_bytes, void* stream);
size_t unet_convT2x2_wgrad_x6_workspace_wg(int N, int H, int W, int Cin, int Cout, int max_workgroups);
int unet_convT2x2_wgrad_x6_wg(const float* xin, int ldx, const float* dz, int lddz, float* dw,
                              int N, int H, int W, int Cin, int Cout, int max_workgroups, void* ws, size_t ws_bytes, void* stream);
size_t unet_convT2x2_wgrad_bf16_workspace_wg(int N, int H, int W, int Cin, int Cout, int max_workgroups);
int unet_convT2x2_wgrad_bf16_wg(const void* xin, int ldx, int x_bf16, const void* dz, int lddz, int dz_bf16, float* dw,
                                int N, int H, int W, int Cin, int Cout, int max_workgroups, void* ws, size_t ws_bytes, void* stream);
/* Measurement aid for the above (tests/test_gpu_overlap.py, scripts/overlap_probe.py): a kernel that behaves like a collective's --
 * `workgroups` workgroups of 512 threads, each holding lds_bytes of LDS (<= 65536) and its CU slot for `microseconds` after it STARTS
 * (every wave leaves when the constant-rate wall clock passes its deadline; nothing else is read or written).  Not called by the engine. */
int unet_standin_collective(int workgroups, int lds_bytes, int microseconds, void* stream);

/* ---- host-side helper of the checkpoint format (tf.train.Checkpoint = TensorBundle, UNet/train.py:96,184; UNet/model.py:81-83):
 * CRC-32C as TensorFlow's crc32c::Extend(init, data, n) -- the block trailers of `ckpt.index` and the per-tensor checksums of
 * BundleEntryProto.  Pure host code, no device work, no stream. */
uint32_t unet_crc32c_extend(uint32_t init, const void* data, size_t n);

#ifdef __cplusplus
}
#endif
#endif
