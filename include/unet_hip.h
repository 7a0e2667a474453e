/*
 * libunet_hip.so -- C-ABI of the MI355X-native U-Net anomaly-segmentation hot path.
 *
 * The reference (ukeSJTU/tiaozhanbei-unet) has no FFI: its hot path is reached through
 * Python objects (src/model.py, src/train_utils.py) and the arithmetic lives in ATen.
 * This header is the boundary BELOW those Python objects: every entry point names the
 * reference call site (file:line under /root/reference) whose arithmetic it replaces.
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers + sizes; no torch types.  All device tensors are dense NHWC
 *     ("channels last": [N][H][W][C], C fastest) in the compute dtype (UNET_F32 or
 *     UNET_BF16) unless a parameter says otherwise; parameters/gradients of the model
 *     are fp32 in PyTorch's own layouts (OIHW etc.).
 *   - the caller owns every buffer (inputs, outputs, workspace); the library keeps no
 *     device state, never allocates or frees device memory and never synchronises:
 *     all work is enqueued on `stream` (a hipStream_t passed as void*).
 *   - return value: UNET_OK (0) or a negative unet_status; unet_last_error() gives the
 *     thread-local text of the last failure.  No C++ exception crosses the ABI.
 *   - `unet_view` places an NHWC tensor inside a larger logical frame: it is how the
 *     skip-concat (`torch.cat([x2, x1], 1)`, src/model.py:65) and the centre pad
 *     (`F.pad`, src/model.py:57-61) are expressed without ever materialising them.
 */
#ifndef UNET_HIP_H_
#define UNET_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UNET_ABI_VERSION 1

enum unet_dtype { UNET_F32 = 0, UNET_BF16 = 1 };

enum unet_status {
  UNET_OK = 0,
  UNET_ERR_BAD_ARG = -1,      /* null pointer / negative size / inconsistent description */
  UNET_ERR_UNSUPPORTED = -2,  /* shape outside what the kernels cover (see each function)  */
  UNET_ERR_WORKSPACE = -3,    /* workspace too small                                        */
  UNET_ERR_LAUNCH = -4        /* HIP reported a launch error (text in unet_last_error)      */
};

/* An NHWC tensor [n][h][w][c] whose pixel (0,0) sits at (off_y, off_x) of a logical frame.
 * Reads outside the tensor give 0; writes outside are dropped. */
typedef struct unet_view {
  void* ptr;
  int32_t c;
  int32_t h, w;
  int32_t off_y, off_x;
} unet_view;

int32_t unet_abi_version(void);
const char* unet_last_error(void);

/* The UNET_* tuning variables are read once, at the first launch that consults them; this re-reads
 * them (same-process A/B tools such as tools/bench_layer.py --ab). */
int32_t unet_tuning_reload(void);

/* Data parallelism (no reference counterpart): CUs the persistent one-block-per-CU kernels leave free for
 * the RCCL all-reduce kernels that overlap the backward pass.  The statically partitioned conv / weight-gradient /
 * transposed-conv launchers size their grids by unet_get_cu_budget() = (multiprocessor count - reserved) rounded down
 * to a multiple of 8 XCDs.  Default 0 (or UNET_RESERVED_CUS); ddp.py sets it when the world size is > 1. */
int32_t unet_set_reserved_cus(int32_t n);
int32_t unet_get_cu_budget(void);
/* Diagnostic (tools/cu_share_probe.py): `blocks` workgroups holding `lds_bytes` of LDS spin for `microseconds` on `stream`
 * -- a stand-in for resident collective kernels, to measure what CU sharing costs the persistent kernels. */
int32_t unet_debug_spin(int32_t blocks, int32_t lds_bytes, int32_t microseconds, void* stream);

/* ---- per-kernel-class timing (used by bench.py for the roofline object) -------------- */
enum unet_kclass {
  UNET_K_CONV_FWD = 0, UNET_K_CONV_DGRAD, UNET_K_CONV_WGRAD, UNET_K_CONVT_FWD, UNET_K_CONVT_DGRAD,
  UNET_K_CONVT_WGRAD, UNET_K_BN, UNET_K_POOL, UNET_K_HEAD, UNET_K_LOSS, UNET_K_PACK, UNET_K_OTHER,
  UNET_K_COUNT
};
/* When enabled, every launch of a timed class is bracketed by hipEvents on its own stream. */
int32_t unet_prof_enable(int32_t on);
/* Synchronises the recorded events; fills ms[UNET_K_COUNT], launches[UNET_K_COUNT],
 * flops[UNET_K_COUNT] (algorithmic FLOPs summed over the launches) and clears the log. */
int32_t unet_prof_collect(double* ms, int64_t* launches, double* flops);
/* Per-KERNEL breakdown of the brackets consumed by the last unet_prof_collect(): entry `index` (0, 1, ... until
 * UNET_ERR_BAD_ARG) -> kernel name (static string), summed event time, launches, algorithmic FLOPs. */
int32_t unet_prof_kernel_stats(int32_t index, const char** name, double* ms, int64_t* launches, double* flops);
/* Algorithmic bytes (inputs + weights + outputs, each once) summed over the launches of entry `index` of the same
 * breakdown; 0 for kernels whose launcher does not state them (the figure roofline.traffic is held against). */
int32_t unet_prof_kernel_bytes(int32_t index, double* bytes);

/* ---- layout ------------------------------------------------------------------------- */
/* NCHW fp32 -> NHWC compute dtype with the channel dim zero-padded to c_pad
 * (the `.to(device)` batch of src/train_utils.py:118 entering src/model.py:190). */
int32_t unet_nchw_to_nhwc(const float* src, void* dst, int32_t n, int32_t c, int32_t h, int32_t w,
                          int32_t c_pad, int32_t dtype, void* stream);
/* NHWC compute dtype (row stride c_pad) -> NCHW fp32, first c channels. */
int32_t unet_nhwc_to_nchw(const void* src, float* dst, int32_t n, int32_t c, int32_t h, int32_t w,
                          int32_t c_pad, int32_t dtype, void* stream);

/* ---- weight packing: fp32 parameters -> GEMM operand layouts in the compute dtype ------ */
enum unet_pack_mode {
  UNET_PACK_CONV_FWD = 0,   /* Conv2d w[co][ci][3][3]   -> [9][co_rows][ci_k]            */
  UNET_PACK_CONV_DGRAD = 1, /* same weight              -> [9 flipped][ci_rows][co_k]    */
  UNET_PACK_CONVT_FWD = 2,  /* ConvTranspose2d w[ci][co][2][2] -> [4][co_rows][ci_k]     */
  UNET_PACK_CONVT_DGRAD = 3 /* same weight              -> [ci_rows][4*co_k]             */
};
/* rows / k are the padded GEMM dims (zero filled); c_out, c_in are the parameter's dims. */
int32_t unet_pack_weight(const float* w, void* out, int32_t c_out, int32_t c_in, int32_t rows,
                         int32_t k, int32_t mode, int32_t dtype, void* stream);

/* All of a model's weights in ONE launch (after each optimiser step): `descs` is a DEVICE array of n
 * descriptors (the fields of unet_pack_weight; dtype is per call). */
typedef struct unet_pack_desc {
  const float* w;
  void* out;
  int32_t c_out, c_in, rows, k, mode, reserved;
} unet_pack_desc;
int32_t unet_pack_weights_batched(const unet_pack_desc* descs, int32_t n, int32_t dtype, void* stream);

/* ---- 3x3 convolution, pad 1, stride 1, no bias (nn.Conv2d at src/model.py:14,17) ------ */
/* y = conv(concat(src[0], src[1])) as an implicit GEMM on MFMA.  Output channels below
 * dst_split go to dst[0], the rest to dst[1] (dst[1].ptr may be NULL when dst_split == c_out).
 * The same entry computes the data gradient when given UNET_PACK_CONV_DGRAD weights:
 * dX = conv(dY, flipped W^T).  `accumulate` is a bit mask: bit 0 -> dst[0] += result, bit 1 -> dst[1] += result
 * (gradient fan-in of the skip connections: the three consumers of a skip tensor add into one buffer).
 * Requires: channel counts of each src multiples of 64 (or a single src of 8/16 for the
 * image layer), c_out multiple of 64. */
int32_t unet_conv3x3(int32_t dtype, int32_t n, int32_t h, int32_t w, const unet_view src[2],
                     const void* w_packed, int32_t c_out, const unet_view dst[2], int32_t dst_split,
                     int32_t accumulate, int32_t kclass, void* stream);

/* Inference form of conv + BatchNorm(eval) + ReLU (src/model.py:14-19 under model.eval(), src/test.py:68): BatchNorm's
 * running statistics are folded into the layer -- scale = gamma / sqrt(running_var + eps) into the packed weights
 * (unet_pack_conv_weight_folded; scale / shift from unet_bn_eval_coeffs), shift = beta - running_mean * scale as the
 * bias of the convolution's epilogue, followed by max(., 0) when relu != 0.  One kernel per layer, the activation is
 * written once (y: dense NHWC [n][h][w][c_out] in the compute dtype). */
int32_t unet_pack_conv_weight_folded(const float* w, const float* scale, void* out, int32_t c_out, int32_t c_in,
                                     int32_t rows, int32_t k, int32_t dtype, void* stream);
int32_t unet_conv3x3_bias_relu(int32_t dtype, int32_t n, int32_t h, int32_t w, const unet_view src[2],
                               const void* w_packed, int32_t c_out, void* y, const float* bias, int32_t relu,
                               void* stream);

/* The FIRST convolution of the network (inc.double_conv.0: nn.Conv2d(n_channels, 64, 3, padding=1), src/model.py:14
 * reached from UNet.forward :98 / AnomalyUNet.forward :190), bf16 mode, straight from the caller's fp32 NCHW image
 * and fp32 OIHW weight: with 9*c_in <= 32 the whole reduction is one 32-deep MFMA step, so the image is never padded
 * to 64 channels.  y: bf16 NHWC [n][h][w][64].  partial (may be NULL: no statistics) receives *n_parts ordered
 * BatchNorm partials [part][2][64] for unet_bn_finalize_partials (capacity: unet_conv3x3_stats_max_parts).
 * unet_conv3x3_first_supported() tells whether a layer qualifies (c_out == 64, 9*c_in <= 32, w % 16 == 0). */
int32_t unet_conv3x3_first_supported(int32_t c_in, int32_t c_out, int32_t h, int32_t w);
int32_t unet_conv3x3_first_stats(int32_t n, int32_t h, int32_t w, const float* x_nchw, int32_t c_in,
                                 const float* weight_oihw, void* y, float* partial, int32_t* n_parts, void* stream);
/* dW[64][c_in][3][3] (fp32) = sum over pixels of dY (bf16 NHWC) x the shifted image; ordered reductions. */
size_t unet_conv3x3_first_wgrad_workspace(int32_t n, int32_t h, int32_t w);
int32_t unet_conv3x3_first_wgrad(int32_t n, int32_t h, int32_t w, const float* x_nchw, int32_t c_in, const void* dy,
                                 float* dw, void* workspace, size_t workspace_bytes, void* stream);
/* The same weight gradient with the BatchNorm-backward apply of the layer (src/model.py:15, autograd) folded in: the
 * image layer's dy has no other consumer (no data gradient towards the image), so instead of a standalone pass
 * dy = A*dz + B*y + K the kernel streams the ReLU-masked gradient dz and the raw conv output y (both bf16 NHWC
 * [n][h][w][64]) and forms dy -- rounded to bf16 as the pass stored it: bit-identical dW -- on its operand.
 * coefs: DEVICE [3][64] = A, B, K, as unet_bn_bwd_premasked leaves them in its workspace when called with dy = NULL. */
int32_t unet_conv3x3_first_wgrad_bn(int32_t n, int32_t h, int32_t w, const float* x_nchw, int32_t c_in, const void* dz,
                                    const void* y, const float* coefs, float* dw, void* workspace,
                                    size_t workspace_bytes, void* stream);

/* The forward convolution of DoubleConv fused with the BatchNorm batch statistics (src/model.py:14-15,
 * 17-18): y = conv(concat(src)) AND per-channel partial sums (sum, sum of squares of the stored y) written
 * by the conv epilogue through wavefront reductions -- or, for kernel variants without that epilogue, by one
 * extra streaming pass.  partial: fp32 [parts][2][c_out], capacity unet_conv3x3_stats_max_parts() parts;
 * *n_parts receives the number written.  Feed them to unet_bn_finalize_partials(). */
size_t unet_conv3x3_stats_max_parts(int32_t n, int32_t h, int32_t w);
int32_t unet_conv3x3_stats(int32_t dtype, int32_t n, int32_t h, int32_t w, const unet_view src[2],
                           const void* w_packed, int32_t c_out, void* y, float* partial, int32_t* n_parts,
                           void* stream);

/* Data gradient of the SECOND convolution of DoubleConv (src/model.py:17) fused with the backward of the ReLU and
 * the reduction half of the BatchNorm2d backward of the FIRST one (src/model.py:15-16), whose activation is this
 * convolution's only input: dz = conv(dy, flipped W^T) * [fma(y_prev, bn_scale, bn_shift) > 0] (dense NHWC
 * [n][h][w][c_dx], compute dtype) and partial[*n_parts][2][c_dx] = per-channel (sum dz, sum dz * (y_prev - bn_mean))
 * of the stored dz, written by the kernel's epilogue -- feed both to unet_bn_bwd_premasked.  The (y, da) reduction pass
 * of unet_bn_relu_bwd does not exist on this route.  w_packed = UNET_PACK_CONV_DGRAD [9][c_dx][c_dy]; partial capacity:
 * unet_conv3x3_stats_max_parts(n, h, w) parts.  unet_conv3x3_dgrad_bnrelu_supported() says whether a shape is covered
 * (bf16, 16-aligned frames, c_dy >= 128); otherwise use unet_conv3x3 + unet_bn_relu_bwd. */
int32_t unet_conv3x3_dgrad_bnrelu_supported(int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c_dy, int32_t c_dx);
int32_t unet_conv3x3_dgrad_bnrelu(int32_t dtype, int32_t n, int32_t h, int32_t w, const void* dy, int32_t c_dy,
                                  const void* w_packed, int32_t c_dx, const void* y_prev, const float* bn_scale,
                                  const float* bn_shift, const float* bn_mean, void* dz, float* partial,
                                  int32_t* n_parts, void* stream);

/* dW[co][ci][3][3] (fp32, OIHW) = sum over pixels of dY (x) shifted X; split-K over pixel
 * tiles with fp32 partial slabs in `workspace`, reduced in a fixed order (deterministic).
 * (autograd of nn.Conv2d, reached from total_loss.backward() at src/train_utils.py:132) */
size_t unet_conv3x3_wgrad_workspace(int32_t n, int32_t h, int32_t w, int32_t c_in, int32_t c_out);
int32_t unet_conv3x3_wgrad(int32_t dtype, int32_t n, int32_t h, int32_t w, const unet_view src[2],
                           const void* dy, int32_t c_out, float* dw, int32_t c_in_param,
                           void* workspace, size_t workspace_bytes, void* stream);

/* ---- 2x2 stride-2 transposed convolution with bias (nn.ConvTranspose2d, src/model.py:51) */
/* y[n][2i+k][2j+l][co] = b[co] + sum_ci x[n][i][j][ci] * w[ci][co][k][l]  (one GEMM + pixel shuffle) */
int32_t unet_convt2x2_fwd(int32_t dtype, int32_t n, int32_t h, int32_t w, const void* x, int32_t c_in,
                          const void* w_packed, const float* bias, void* y, int32_t c_out, void* stream);
/* dx from dy (dy is the [n][2h][2w][c_out] gradient); w_packed = UNET_PACK_CONVT_DGRAD. */
int32_t unet_convt2x2_dgrad(int32_t dtype, int32_t n, int32_t h, int32_t w, const void* dy,
                            int32_t c_out, const void* w_packed, void* dx, int32_t c_in, void* stream);

/* The same data gradient when the transposed convolution's input is the activation of a conv-BatchNorm-ReLU layer with
 * no other consumer (the DoubleConv in front of an Up block: src/model.py:14-19 -> :51, autograd of Up.forward :55):
 * the epilogue applies that layer's ReLU mask (from its raw conv output y_prev and scale/shift) and reduces the
 * BatchNorm-backward sums, as unet_conv3x3_dgrad_bnrelu does for a 3x3 consumer.  dz = masked gradient
 * [n][h][w][c_in]; partial = [unet_convt2x2_dgrad_bnrelu_max_parts()][2][c_in] floats (sum dz, sum dz*(y-mean)),
 * *n_parts rows written; finish with unet_bn_bwd_premasked. */
int32_t unet_convt2x2_dgrad_bnrelu_supported(int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c_in, int32_t c_out);
size_t unet_convt2x2_dgrad_bnrelu_max_parts(void);
int32_t unet_convt2x2_dgrad_bnrelu(int32_t dtype, int32_t n, int32_t h, int32_t w, const void* dy, int32_t c_out,
                                   const void* w_packed, const void* y_prev, const float* bn_scale, const float* bn_shift,
                                   const float* bn_mean, void* dz, int32_t c_in, float* partial, int32_t* n_parts,
                                   void* stream);
size_t unet_convt2x2_wgrad_workspace(int32_t n, int32_t h, int32_t w, int32_t c_in, int32_t c_out);
/* dw[ci][co][2][2], db[co] in fp32. */
int32_t unet_convt2x2_wgrad(int32_t dtype, int32_t n, int32_t h, int32_t w, const void* x, int32_t c_in,
                            const void* dy, int32_t c_out, float* dw, float* db, void* workspace,
                            size_t workspace_bytes, void* stream);

/* ---- BatchNorm2d (+ ReLU) (nn.BatchNorm2d / nn.ReLU at src/model.py:15-16,18-19) -------- */
size_t unet_bn_workspace(int64_t pixels, int32_t c);
/* training statistics of y[pixels][c]: biased variance for normalisation, running stats
 * updated with `momentum` towards mean / UNBIASED variance (running_* may be NULL).
 * Outputs (fp32[c]): save_mean, save_istd, scale = gamma*istd, shift = beta - mean*scale. */
int32_t unet_bn_train_stats(int32_t dtype, const void* y, int64_t pixels, int32_t c, const float* gamma,
                            const float* beta, float* running_mean, float* running_var, float momentum,
                            float eps, float* save_mean, float* save_istd, float* scale, float* shift,
                            void* workspace, size_t workspace_bytes, void* stream);
/* same outputs as unet_bn_train_stats, from partial sums [n_parts][2][c] (ordered fp64 reduction). */
int32_t unet_bn_finalize_partials(const float* partial, int32_t n_parts, int64_t pixels, int32_t c,
                                  const float* gamma, const float* beta, float* running_mean,
                                  float* running_var, float momentum, float eps, float* save_mean,
                                  float* save_istd, float* scale, float* shift, void* stream);
/* eval: scale/shift from the running statistics. */
int32_t unet_bn_eval_coeffs(int32_t c, const float* gamma, const float* beta, const float* running_mean,
                            const float* running_var, float eps, float* scale, float* shift, void* stream);
/* eval coefficients plus the constants the FROZEN backward needs: mean = running_mean, istd = 1/sqrt(running_var+eps). */
int32_t unet_bn_eval_coeffs4(int32_t c, const float* gamma, const float* beta, const float* running_mean,
                             const float* running_var, float eps, float* mean, float* istd, float* scale,
                             float* shift, void* stream);
/* a = max(fma(y, scale, shift), 0) */
int32_t unet_bn_relu_apply(int32_t dtype, const void* y, int64_t pixels, int32_t c, const float* scale,
                           const float* shift, void* a, void* stream);
/* backward of a = relu(bn(y)) in training mode: dgamma, dbeta (fp32[c]) and
 * dy = gamma*istd*(dz - dbeta/M - yhat*dgamma/M), dz = da*[z>0]. */
int32_t unet_bn_relu_bwd(int32_t dtype, const void* da, const void* y, int64_t pixels, int32_t c,
                         const float* gamma, const float* save_mean, const float* save_istd,
                         const float* scale, const float* shift, float* dgamma, float* dbeta, void* dy,
                         void* workspace, size_t workspace_bytes, void* stream);

/* backward of a = relu(bn(y)) with FROZEN statistics (a BatchNorm2d put in eval() inside a training model -- the
 * fine-tuning pattern the reference's nn.Sequential honours, src/model.py:13-20): mean / istd are constants, so
 * dy = gamma*istd*dz; dgamma = sum dz*yhat, dbeta = sum dz as usual.  Same workspace as unet_bn_relu_bwd. */
int32_t unet_bn_relu_bwd_frozen(int32_t dtype, const void* da, const void* y, int64_t pixels, int32_t c,
                                const float* gamma, const float* mean, const float* istd, const float* scale,
                                const float* shift, float* dgamma, float* dbeta, void* dy, void* workspace,
                                size_t workspace_bytes, void* stream);
/* The same backward when the producer of the gradient has already applied the ReLU mask in its own epilogue and left
 * the two per-channel sums behind (unet_head_bnrelu_bwd, unet_conv3x3_dgrad_bnrelu): dz = da*[z>0] (compute dtype,
 * NHWC), partial = fp32 [n_parts][2][c] holding sum dz and sum dz*(y - mean).  Ordered fp64 finalize -> dgamma, dbeta,
 * then ONE pass dy = A*dz + B*y + K (dy may alias dz).  workspace: 3*c floats = A, B, K on return; dy = NULL skips the
 * pass (dz / y may then be NULL too): the caller's consumer applies the coefficients (unet_conv3x3_first_wgrad_bn). */
int32_t unet_bn_bwd_premasked(int32_t dtype, const void* dz, const void* y, int64_t pixels, int32_t c,
                              const float* gamma, const float* save_mean, const float* save_istd,
                              const float* partial, int32_t n_parts, float* dgamma, float* dbeta, void* dy,
                              void* workspace, size_t workspace_bytes, void* stream);

/* ---- MaxPool2d(2) (src/model.py:32): stride 2, floor; first maximum wins ties ---------- */
int32_t unet_maxpool2_fwd(int32_t dtype, const void* x, int32_t n, int32_t h, int32_t w, int32_t c,
                          void* y, void* stream);
/* accumulate != 0: dx += routed gradient (dx already holds the other consumers' gradient of x). */
int32_t unet_maxpool2_bwd(int32_t dtype, const void* x, const void* dy, int32_t n, int32_t h, int32_t w,
                          int32_t c, void* dx, int32_t accumulate, void* stream);

/* BatchNorm-apply + ReLU of an encoder level fused with the next level's MaxPool2d(2) (src/model.py:18-19 -> :32): one
 * pass over the raw conv output y writes the activation a (the skip tensor) and pooled = maxpool2(a).  The backward takes
 * the gradient other consumers already accumulated for a (da_old, NULL = none) and the pooled gradient, routes the latter
 * to the first maximum of each window, applies this layer's ReLU mask and reduces the BatchNorm-backward sums:
 * dz (may alias da_old) + partial[*n_parts][2][c] (capacity unet_bn_relu_pool_max_parts()) -> unet_bn_bwd_premasked.
 * Needs 256 % (c / 8) == 0 (bf16) or 256 % (c / 4) == 0 (fp32): unet_bn_relu_pool_supported(). */
int32_t unet_bn_relu_pool_supported(int32_t dtype, int32_t c);
int32_t unet_bn_relu_pool_fwd(int32_t dtype, const void* y, int32_t n, int32_t h, int32_t w, int32_t c,
                              const float* scale, const float* shift, void* a, void* pooled, void* stream);
size_t unet_bn_relu_pool_max_parts(void);
int32_t unet_bn_relu_pool_bwd(int32_t dtype, const void* y, const void* dpooled, const void* da_old, int32_t n, int32_t h,
                              int32_t w, int32_t c, const float* scale, const float* shift, const float* mean, void* dz,
                              float* partial, int32_t* n_parts, void* stream);

/* ---- bilinear x2, align_corners=True (nn.Upsample, src/model.py:48) -------------------- */
int32_t unet_upsample_bilinear2x_fwd(int32_t dtype, const void* x, int32_t n, int32_t h, int32_t w,
                                     int32_t c, void* y, void* stream);
int32_t unet_upsample_bilinear2x_bwd(int32_t dtype, const void* dy, int32_t n, int32_t h, int32_t w,
                                     int32_t c, void* dx, void* stream);

/* ---- 1x1 head (OutConv, src/model.py:72) with optional sigmoid (src/model.py:201,208) --- */
/* x: NHWC compute dtype [pixels][c_in]; w fp32 [c_out][c_in]; out: NCHW fp32. c_out <= 8. */
int32_t unet_head_fwd(int32_t dtype, const void* x, int32_t n, int32_t h, int32_t w, int32_t c_in,
                      const float* weight, const float* bias, int32_t c_out, int32_t sigmoid, float* out,
                      void* stream);
size_t unet_head_bwd_workspace(int32_t n, int32_t h, int32_t w, int32_t c_in, int32_t c_out);
/* dout: NCHW fp32 gradient w.r.t. `out`; out: the forward result (needed when sigmoid). */
int32_t unet_head_bwd(int32_t dtype, const void* x, const float* out, const float* dout, int32_t n,
                      int32_t h, int32_t w, int32_t c_in, const float* weight, int32_t c_out,
                      int32_t sigmoid, void* dx, float* dweight, float* dbias, void* workspace,
                      size_t workspace_bytes, void* stream);

/* OutConv fed by the RAW convolution output y of the last conv-BatchNorm-ReLU layer (src/model.py:17-19 followed by
 * :72 / :107 / :200,207): a = max(fma(y, bn_scale, bn_shift), 0) is formed on load -- the activation tensor is never
 * written, its BatchNorm-apply pass does not exist.  bn_scale / bn_shift / bn_mean: outputs of
 * unet_bn_finalize_partials.  The backward recomputes a for dweight, writes dz = da*[z>0] (NHWC compute dtype) and the
 * BatchNorm-backward partial sums bn_partial[*n_parts][2][c_in] (capacity unet_head_bnrelu_max_parts() parts) for
 * unet_bn_bwd_premasked: the (y, da) reduction pass does not exist either. */
int32_t unet_head_bnrelu_fwd(int32_t dtype, const void* y, int32_t n, int32_t h, int32_t w, int32_t c_in,
                             const float* bn_scale, const float* bn_shift, const float* weight, const float* bias,
                             int32_t c_out, int32_t sigmoid, float* out, void* stream);
size_t unet_head_bnrelu_max_parts(void);
int32_t unet_head_bnrelu_bwd(int32_t dtype, const void* y, const float* bn_scale, const float* bn_shift,
                             const float* bn_mean, const float* out, const float* dout, int32_t n, int32_t h, int32_t w,
                             int32_t c_in, const float* weight, int32_t c_out, int32_t sigmoid, void* dz,
                             float* dweight, float* dbias, float* bn_partial, int32_t* n_parts, void* workspace,
                             size_t workspace_bytes, void* stream);

/* ---- loss heads (src/train_utils.py) ----------------------------------------------------- */
size_t unet_loss_workspace(int64_t elems);
/* CombinedLoss.forward (train_utils.py:30-44): losses[0] = mean((recon-image)^2),
 * losses[1] = focal(amap, mask) (train_utils.py:23-28; BCE log clamp -100, backward
 * denominator clamp 1e-12 as ATen).  Also writes d losses[0]/d recon and d losses[1]/d amap. */
int32_t unet_loss_mse_focal(const float* recon, const float* image, int64_t n_recon, const float* amap,
                            const float* mask, int64_t n_amap, float alpha, float gamma, float* losses,
                            float* d_recon, float* d_amap, void* workspace, size_t workspace_bytes,
                            void* stream);
/* SSIMLoss (train_utils.py:67-104): 11-tap separable Gaussian (sigma 1.5), zero pad 5, NCHW fp32
 * planes.  loss[0] = 1 - mean(ssim_map); d_img1/d_img2 (may be NULL) get d loss / d img. */
size_t unet_ssim_workspace(int32_t planes, int32_t h, int32_t w);
int32_t unet_ssim_loss(const float* img1, const float* img2, int32_t planes, int32_t h, int32_t w,
                       int32_t window, float* loss, float* d_img1, float* d_img2, void* workspace,
                       size_t workspace_bytes, void* stream);

/* SSIMLoss(size_average=False) (train_utils.py:84-87): loss[i] = 1 - mean_{c,h,w} ssim_map of image i (n images of c
 * planes); d_img1 / d_img2 (may be NULL) receive d loss[i] / d img for image i.  workspace: unet_ssim_workspace(c, h, w). */
int32_t unet_ssim_loss_per_image(const float* img1, const float* img2, int32_t n, int32_t c, int32_t h, int32_t w,
                                 int32_t window, float* loss, float* d_img1, float* d_img2, void* workspace,
                                 size_t workspace_bytes, void* stream);

/* ---- optimiser (torch.optim.Adam of get_optimizer, src/train_utils.py:266) -------------- */
/* ---- Multi-class segmentation head of the Gear/Kolektor trainers (src/metrics.py) ------------------------ */
/* CombinedSegmentationLoss.forward (src/metrics.py:300-335): loss = ce_weight * CE(class weights, ignore_index)
 * + dice_weight * dice_loss(softmax(logits)) (:233-261) + focal_weight * focal_loss (:264-282), value AND gradient.
 * logits: fp32 NCHW [n][c][hw], c <= 8; target: int64 [n][hw]; class_weights: [c] or NULL; ignore_index < 0: none.
 * input_is_prob != 0: `logits` already holds probabilities (stand-alone dice_loss(pred, target)); Dice term only.
 * loss[4] = {total, ce, dice, focal}; dlogits (same shape as logits) may be NULL.  Ordered reductions. */
size_t unet_seg_loss_workspace(int32_t n, int32_t c, int64_t hw);
int32_t unet_seg_loss(const float* logits, const int64_t* target, int32_t n, int32_t c, int64_t hw,
                      const float* class_weights, int64_t ignore_index, int32_t input_is_prob, float ce_weight,
                      float dice_weight, float focal_weight, float focal_alpha, float focal_gamma, float* loss,
                      float* dlogits, void* workspace, size_t workspace_bytes, void* stream);
/* SegmentationMetrics.update (src/metrics.py:22-45): labels[n][hw] = argmax over classes (the FIRST maximum wins
 * ties, as torch.argmax), confusion[t][p] += 1 over the pixels whose target is a class and not ignore_index.
 * labels or confusion (with target) may be NULL.  Integer atomics: exact. */
int32_t unet_seg_confusion(const float* logits, const int64_t* target, int32_t n, int32_t c, int64_t hw,
                           int64_t ignore_index, int64_t* labels, int64_t* confusion, void* stream);

/* Pixel-level threshold epilogue of the anomaly branch (src/test.py:79-106 evaluate_results, src/train_utils.py:232-245
 * validate_epoch): for each of k <= 8 thresholds the confusion counts {tp, fp, fn, tn} of (pred > t) against
 * (truth > 0.5) over the images with select[n] != 0 (NULL: all).  counts[k][4] (int64, device) is ADDED to: zero it first.
 * pred / truth: fp32 [n_images][per_image].  Integer atomics: exact and order-independent. */
int32_t unet_threshold_confusion(const float* pred, const float* truth, const uint8_t* select, int64_t n_images,
                                 int64_t per_image, const float* thresholds, int32_t k, int64_t* counts, void* stream);

/* ---- nn.Dropout2d of SegmentationUNet's bottleneck (src/model.py:129,146): y = x * scale[n][c] on dense NHWC; the
 * caller draws scale = bernoulli(1-p)/(1-p) per (image, channel); the same call is the backward (dx = dy * scale). */
int32_t unet_channel_scale(int32_t dtype, const void* x, const float* scale, int32_t n, int64_t hw, int32_t c, void* y,
                           void* stream);
/* ---- compute_anomaly_score (src/utils.py:205-215): score[n][hw] = mean_c (recon - image)^2 (l1 != 0: mean_c |.|)
 * over fp32 NCHW planes, image_score[n] = mean over pixels (ordered partials). */
size_t unet_anomaly_score_workspace(int32_t n, int64_t hw);
int32_t unet_anomaly_score(const float* recon, const float* image, int32_t n, int32_t c, int64_t hw, int32_t l1,
                           float* score, float* image_score, void* workspace, size_t workspace_bytes, void* stream);

/* ---- input pipeline on the GPU: transforms.ToTensor() + Normalize(mean, std) (+ the training
 * RandomHorizontalFlip) of src/dataset.py:134-146 / src/kolektorsdd_dataset.py:133-150 for a batch of decoded uint8
 * images [n][h][w][3] (device memory): out[n][c][y][x] = (u8 / 255 - mean[c]) / std[c], mirrored in x where flip[n] != 0
 * (flip may be NULL; mean3 / std3 are HOST arrays).  The same fp32 operations in the same order: bit-identical. */
int32_t unet_preprocess_u8(const uint8_t* images_hwc, const uint8_t* flip, float* out_nchw, int32_t n, int32_t h,
                           int32_t w, const float* mean3, const float* std3, void* stream);

/* ---- the rest of the loaders' image transform on the GPU (round 4).  The reference runs torchvision on
 * PIL images (src/dataset.py:134-141: Resize -> RandomHorizontalFlip -> RandomRotation(10) -> ColorJitter(0.1, 0.1,
 * 0.1, 0.05) -> ToTensor -> Normalize; src/kolektorsdd_dataset.py:133-150: the same with RandomRotation(5), masks
 * Resize(NEAREST)); each entry point below restates the arithmetic of the Pillow C kernel that transform ends in and is
 * bit-exact against fixtures produced by PIL (tests/golden/aug_*.npz).  Batches are decoded uint8 images [n][h][w][c]
 * in device memory; the random draws stay with the host and come in as per-image parameters.
 *
 * transforms.Resize on a PIL image = Image.resize(BILINEAR) = ImagingResample (Resample.c): a separable triangle filter
 * whose support grows with the down-scale factor, 22-bit fixed-point coefficients, 8-bit intermediate image between the
 * horizontal and the vertical pass.  unet_resize_bilinear_coeffs (HOST function, host arrays) writes PIL's tables for
 * one axis: bounds[out][2] = (first source index, count), kk[out][ksize] with ksize = unet_resize_bilinear_ksize(..).
 * unet_resize_bilinear_u8 takes those tables in DEVICE memory (x tables for w -> out_w, y tables for h -> out_h; a pass
 * whose size does not change is skipped like PIL skips it and its tables may be NULL), tmp = [n][h][out_w][c] bytes
 * (needed when both passes run), c = 1 or 3. */
int32_t unet_resize_bilinear_ksize(int32_t in_size, int32_t out_size);
int32_t unet_resize_bilinear_coeffs(int32_t in_size, int32_t out_size, int32_t* bounds, int32_t* kk);
int32_t unet_resize_bilinear_u8(const uint8_t* src, int32_t n, int32_t h, int32_t w, int32_t c, int32_t out_h,
                                int32_t out_w, const int32_t* xbounds, const int32_t* xkk, int32_t xksize,
                                const int32_t* ybounds, const int32_t* ykk, int32_t yksize, uint8_t* tmp, uint8_t* dst,
                                void* stream);
/* Image.resize(NEAREST) (Geometry.c ImagingScaleAffine; the mask transform of src/kolektorsdd_dataset.py:147-150 and
 * :116-117): unet_resize_nearest_index (HOST) writes the source index of every output position of one axis (-1 =
 * outside); unet_resize_nearest_u8 gathers with the y / x tables in DEVICE memory. */
int32_t unet_resize_nearest_index(int32_t in_size, int32_t out_size, int32_t* idx);
int32_t unet_resize_nearest_u8(const uint8_t* src, int32_t n, int32_t h, int32_t w, int32_t c, int32_t out_h,
                               int32_t out_w, const int32_t* yidx, const int32_t* xidx, uint8_t* dst, void* stream);
/* RandomHorizontalFlip + RandomRotation: dst = rotate(flip[n] ? mirror(src) : src) with Image.rotate(angle, NEAREST,
 * expand=False, fill 0) = Geometry.c affine_fixed in 16.16 fixed point.  matrices[n][6] (DEVICE, may be NULL: no
 * rotation) = {a0, a1, a2, a3, a4, a5} ALREADY in fixed point, a2 / a5 including the half-pixel terms (FIX(a[2] +
 * a[0]*0.5 + a[1]*0.5)); flip[n] DEVICE bytes, may be NULL.  src != dst; sides < 32768; c <= 4. */
int32_t unet_flip_rotate_u8(const uint8_t* src, int32_t n, int32_t h, int32_t w, int32_t c, const uint8_t* flip,
                            const int32_t* matrices, uint8_t* dst, void* stream);
/* ColorJitter + ToTensor + Normalize of RGB batches.  Per image: `order` = the permutation torchvision draws (entries:
 * 0 brightness, 1 contrast, 2 saturation, 3 hue, -1 = skip), the three ImageEnhance factors (Blend.c: float32, truncating)
 * and hue_shift = uint8(hue_factor * 255) added to H of Convert.c's integer HSV.  The contrast operation needs the mean
 * L of the image as it is at that point of the list: one integer-sum pass, then one pass that applies the list and
 * writes out[n][c][y][x] = (u8 / 255 - mean[c]) / std[c] (fp32 NCHW, as unet_preprocess_u8).  desc: DEVICE array of n
 * (NULL: no jitter, no workspace needed); mean3 / std3: HOST arrays. */
typedef struct unet_jitter_desc {
  int32_t order[4];
  float brightness, contrast, saturation;
  int32_t hue_shift;
} unet_jitter_desc;
size_t unet_color_jitter_workspace(int32_t n);
int32_t unet_color_jitter_normalize_u8(const uint8_t* images_hwc, int32_t n, int32_t h, int32_t w,
                                       const unet_jitter_desc* desc, const float* mean3, const float* std3,
                                       float* out_nchw, void* workspace, size_t workspace_bytes, void* stream);

/* One fused step over a flat fp32 parameter arena: L2-coupled weight decay, bias correction,
 * gradient pre-scale (1/world_size under data parallelism). step is 1-based.  The betas are doubles so that 1 - beta is
 * formed in double before rounding to fp32, as torch.optim.Adam's `value=1 - beta2` is. */
int32_t unet_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                       float lr, double beta1, double beta2, float eps, float weight_decay,
                       float grad_scale, int32_t step, void* stream);

/* The optimiser step of get_optimizer's Adam / AdamW (src/train_utils.py:263-270) for ALL parameter tensors of a model in
 * ONE launch (the reference's torch.optim.Adam walks the 98 tensors): descs (DEVICE array) = one entry per tensor,
 * chunks (DEVICE array) = one entry per block: block b updates elements [first, first + unet_adam_chunk_elems()) of
 * tensor `tensor`.  Same arithmetic as unet_adam_step; decoupled != 0 = AdamW (p *= 1 - lr*wd instead of g += wd*p);
 * grad_scale folds the 1/world of data parallelism into the step. */
typedef struct unet_adam_desc {
  float* p;
  const float* g;
  float* m;
  float* v;
  int64_t n;
} unet_adam_desc;
typedef struct unet_adam_chunk {
  int32_t tensor;
  int32_t reserved;
  int64_t first;
} unet_adam_chunk;
int32_t unet_adam_chunk_elems(void);
int32_t unet_adam_multi(const unet_adam_desc* descs, const unet_adam_chunk* chunks, int32_t n_chunks, float lr, double beta1,
                        double beta2, float eps, float weight_decay, float grad_scale, int32_t step, int32_t decoupled,
                        void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UNET_HIP_H_ */
