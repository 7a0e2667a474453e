// Multi-class segmentation head of the Gear / Kolektor trainers ("next" row f-1):
//   CombinedSegmentationLoss.forward  /root/reference/src/metrics.py:300-335
//     cross entropy (class weights, ignore_index)  :312-320
//     Dice of softmax(pred) vs one-hot target      :323-326 -> dice_loss :233-261
//     focal                                        :329-331 -> focal_loss :264-282
//   SegmentationMetrics.update                     :22-45  (argmax :31, confusion matrix :43)
//
// Logits are NCHW fp32 with C <= 8 classes: a lane owns a pixel and reads its C logits from C planes (each plane
// read is 256 contiguous bytes per wave).  HBM-bound streaming kernels, no MFMA:
//   seg_reduce_kernel    softmax per pixel; per-block partial sums of the CE numerator / denominator, the focal sum
//                        and the per-(image, class) Dice sums (intersection, sum p, sum one-hot)
//   seg_finalize_kernel  ordered fp64 reduction of the partials -> the four loss values and the per-(image, class)
//                        Dice gradient coefficients
//   seg_grad_kernel      recomputes the softmax and writes d loss / d logits (CE + focal through (p - onehot), Dice
//                        through the softmax Jacobian)
//   seg_confusion_kernel first-maximum argmax (torch.argmax tie rule), optional label map, confusion matrix counted
//                        with integer atomics (exact, order-independent)
// Everything is deterministic: fixed pixel -> block mapping, ordered reductions, integer atomics only.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int MAXC = 8;
constexpr int SEG_THREADS = 256;
constexpr int NSUM = 3 * MAXC + 3;       // per block: I[c], P[c], T[c], ce_num, ce_den, focal_sum

struct SegParams {
  const float* logits; const long long* target; const float* cw;
  int N, C; long long hw; long long ignore; int is_prob;
  float alpha, gamma;
  int bpi;                               // blocks per image
};

__device__ __forceinline__ void softmax_c(const float (&z)[MAXC], int C, float (&p)[MAXC]) {
  float m = z[0];
#pragma unroll
  for (int c = 1; c < MAXC; ++c) if (c < C) m = fmaxf(m, z[c]);
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) { p[c] = c < C ? expf(z[c] - m) : 0.f; s += p[c]; }
  const float inv = 1.f / s;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) p[c] *= inv;
}

__global__ __launch_bounds__(SEG_THREADS) void seg_reduce_kernel(const SegParams P, float* __restrict__ part) {
  __shared__ float red[SEG_THREADS / 64][NSUM];
  const int n = blockIdx.y, b = blockIdx.x;
  const long long per = cdiv64(P.hw, P.bpi);
  const long long q0 = b * per, q1 = min(q0 + per, P.hw);
  float acc[NSUM];
#pragma unroll
  for (int i = 0; i < NSUM; ++i) acc[i] = 0.f;
  for (long long q = q0 + threadIdx.x; q < q1; q += SEG_THREADS) {
    float z[MAXC], p[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) z[c] = c < P.C ? P.logits[((long long)n * P.C + c) * P.hw + q] : 0.f;
    if (P.is_prob) {
#pragma unroll
      for (int c = 0; c < MAXC; ++c) p[c] = z[c];
    } else {
      softmax_c(z, P.C, p);
    }
    const long long t = P.target[(long long)n * P.hw + q];
    const bool valid = t != P.ignore && t >= 0 && t < P.C;
    float pt = 1.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const bool hit = valid && t == c;
      acc[c] += hit ? p[c] : 0.f;                // intersection
      acc[MAXC + c] += p[c];                     // sum p
      acc[2 * MAXC + c] += hit ? 1.f : 0.f;      // sum one-hot
      if (hit) pt = p[c];
    }
    if (valid) {
      const float ce = -logf(fmaxf(pt, 1e-38f));
      float w = 1.f;
      if (P.cw) w = P.cw[t];
      acc[3 * MAXC] += w * ce;
      acc[3 * MAXC + 1] += w;
      acc[3 * MAXC + 2] += P.alpha * powf(1.f - pt, P.gamma) * ce;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NSUM; ++i) {
    float v = acc[i];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    if (lane == 0) red[wave][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < NSUM) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < SEG_THREADS / 64; ++w) t += red[w][threadIdx.x];
    part[((size_t)n * P.bpi + b) * NSUM + threadIdx.x] = t;
  }
}

// coef layout (floats): [0..3] loss total/ce/dice/focal, [4] ce scale, [5] focal scale, then A[n][c], B[n][c]
__global__ __launch_bounds__(256) void seg_finalize_kernel(const float* __restrict__ part, int N, int C, int bpi,
                                                           long long hw, float ce_w, float dice_w, float focal_w,
                                                           float* __restrict__ loss_out, float* __restrict__ coef) {
  __shared__ double tot[3];
  __shared__ double dsum[256];
  const int tid = threadIdx.x;
  if (tid < 3) {                                   // ce_num, ce_den, focal over all images / blocks, fixed order
    double s = 0.0;
    for (int i = 0; i < N * bpi; ++i) s += (double)part[(size_t)i * NSUM + 3 * MAXC + tid];
    tot[tid] = s;
  }
  // Dice per (image, class): thread = n*C + c (N*C <= 256 per pass)
  double dloss = 0.0;
  const double smooth = 1e-8, inv_nc = 1.0 / ((double)N * C);
  for (int i = tid; i < N * C; i += 256) {
    const int n = i / C, c = i - n * C;
    double I = 0.0, Pp = 0.0, T = 0.0;
    for (int b = 0; b < bpi; ++b) {
      const float* q = part + ((size_t)n * bpi + b) * NSUM;
      I += (double)q[c]; Pp += (double)q[MAXC + c]; T += (double)q[2 * MAXC + c];
    }
    const double U = Pp + T + smooth;
    dloss += (2.0 * I + smooth) / U;
    // d(1 - mean dice)/dp = -(1/NC) * [2*onehot*U - (2I+s)] / U^2 = A*onehot + B
    coef[6 + i] = (float)(-dice_w * inv_nc * 2.0 / U);
    coef[6 + N * C + i] = (float)(dice_w * inv_nc * (2.0 * I + smooth) / (U * U));
  }
  dsum[tid] = dloss;
  __syncthreads();
  if (tid == 0) {
    double d = 0.0;
    for (int i = 0; i < 256; ++i) d += dsum[i];
    const double ce = tot[1] > 0.0 ? tot[0] / tot[1] : 0.0;
    const double dice = 1.0 - d * inv_nc;
    const double focal = tot[2] / ((double)N * (double)hw);
    const double total = (ce_w > 0 ? ce_w * ce : 0.0) + (dice_w > 0 ? dice_w * dice : 0.0) +
                         (focal_w > 0 ? focal_w * focal : 0.0);
    loss_out[0] = (float)total; loss_out[1] = (float)ce; loss_out[2] = (float)dice; loss_out[3] = (float)focal;
    coef[4] = (float)(ce_w > 0 && tot[1] > 0.0 ? ce_w / tot[1] : 0.0);
    coef[5] = (float)(focal_w > 0 ? focal_w / ((double)N * (double)hw) : 0.0);
  }
}

__global__ __launch_bounds__(SEG_THREADS) void seg_grad_kernel(const SegParams P, const float* __restrict__ coef,
                                                                int dice_on, float* __restrict__ dlogits) {
  const int n = blockIdx.y;
  const float ce_s = coef[4], fo_s = coef[5];
  float A[MAXC], B[MAXC];
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    A[c] = (dice_on && c < P.C) ? coef[6 + n * P.C + c] : 0.f;
    B[c] = (dice_on && c < P.C) ? coef[6 + P.N * P.C + n * P.C + c] : 0.f;
  }
  for (long long q = blockIdx.x * (long long)SEG_THREADS + threadIdx.x; q < P.hw; q += (long long)gridDim.x * SEG_THREADS) {
    float z[MAXC], p[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) z[c] = c < P.C ? P.logits[((long long)n * P.C + c) * P.hw + q] : 0.f;
    if (P.is_prob) {
#pragma unroll
      for (int c = 0; c < MAXC; ++c) p[c] = z[c];
    } else {
      softmax_c(z, P.C, p);
    }
    const long long t = P.target[(long long)n * P.hw + q];
    const bool valid = t != P.ignore && t >= 0 && t < P.C;
    float g[MAXC], gp = 0.f, pt = 1.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const bool hit = valid && t == c;
      g[c] = (hit ? A[c] : 0.f) + B[c];          // d loss / d p_c (Dice)
      gp += g[c] * p[c];
      if (hit) pt = p[c];
    }
    float k = 0.f;                                // CE + focal both act through (p - onehot)
    if (valid) {
      const float ce = -logf(fmaxf(pt, 1e-38f));
      const float w = P.cw ? P.cw[t] : 1.f;
      const float om = 1.f - pt;
      k = ce_s * w + fo_s * P.alpha * (powf(om, P.gamma) + P.gamma * powf(om, P.gamma - 1.f) * pt * ce);
    }
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      if (c < P.C) {
        const bool hit = valid && t == c;
        float d;
        if (P.is_prob) d = g[c];                                  // input was already a probability map
        else d = p[c] * (g[c] - gp) + k * (p[c] - (hit ? 1.f : 0.f));
        dlogits[((long long)n * P.C + c) * P.hw + q] = d;
      }
    }
  }
}

__global__ __launch_bounds__(SEG_THREADS) void seg_confusion_kernel(const float* __restrict__ logits,
                                                                     const long long* __restrict__ target, int N, int C,
                                                                     long long hw, long long ignore,
                                                                     long long* __restrict__ labels,
                                                                     unsigned long long* __restrict__ cm) {
  __shared__ unsigned int cnt[MAXC * MAXC];
  if (threadIdx.x < MAXC * MAXC) cnt[threadIdx.x] = 0;
  __syncthreads();
  const long long total = (long long)N * hw;
  for (long long i = blockIdx.x * (long long)SEG_THREADS + threadIdx.x; i < total; i += (long long)gridDim.x * SEG_THREADS) {
    const long long n = i / hw, q = i - n * hw;
    float best = logits[(n * C) * hw + q];
    int arg = 0;
    for (int c = 1; c < C; ++c) {
      const float v = logits[(n * C + c) * hw + q];
      if (v > best) { best = v; arg = c; }       // strict: the FIRST maximum wins (torch.argmax)
    }
    if (labels) labels[i] = arg;
    if (cm && target) {
      const long long t = target[i];
      if (t != ignore && t >= 0 && t < C) atomicAdd(&cnt[(int)t * C + arg], 1u);
    }
  }
  __syncthreads();
  if (cm && threadIdx.x < C * C && cnt[threadIdx.x])
    atomicAdd(&cm[threadIdx.x], (unsigned long long)cnt[threadIdx.x]);
}

inline int seg_bpi(int n, long long hw) {
  long long b = cdiv64(hw, 4096);                 // >= 16 pixels per thread
  const long long cap = 2048 / (n > 0 ? n : 1) > 0 ? 2048 / n : 1;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" size_t unet_seg_loss_workspace(int32_t n, int32_t c, int64_t hw) {
  return ((size_t)n * seg_bpi(n, hw) * NSUM + 6 + (size_t)2 * n * c) * sizeof(float);
}

extern "C" int32_t unet_seg_loss(const float* logits, const int64_t* target, int32_t n, int32_t c, int64_t hw,
                                 const float* class_weights, int64_t ignore_index, int32_t input_is_prob,
                                 float ce_weight, float dice_weight, float focal_weight, float focal_alpha,
                                 float focal_gamma, float* loss, float* dlogits, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  UNET_REQUIRE(logits && target && loss && workspace, UNET_ERR_BAD_ARG, "unet_seg_loss: null pointer");
  UNET_REQUIRE(n > 0 && hw > 0 && c >= 1 && c <= MAXC, UNET_ERR_UNSUPPORTED, "unet_seg_loss: n=%d c=%d (1..8 classes)", n, c);
  UNET_REQUIRE((long long)n * c <= 4096, UNET_ERR_UNSUPPORTED, "unet_seg_loss: n*c = %lld too large", (long long)n * c);
  UNET_REQUIRE(workspace_bytes >= unet_seg_loss_workspace(n, c, hw), UNET_ERR_WORKSPACE, "unet_seg_loss: workspace too small");
  UNET_REQUIRE(!input_is_prob || (ce_weight == 0.f && focal_weight == 0.f), UNET_ERR_BAD_ARG,
               "unet_seg_loss: a probability map only supports the Dice term");
  hipStream_t s = (hipStream_t)stream;
  const int bpi = seg_bpi(n, hw);
  float* part = (float*)workspace;
  float* coef = part + (size_t)n * bpi * NSUM;
  SegParams P{logits, (const long long*)target, class_weights, n, c, (long long)hw, (long long)ignore_index,
              input_is_prob, focal_alpha, focal_gamma, bpi};
  ProfScope prof(UNET_K_LOSS, 0.0, s);
  hipLaunchKernelGGL(seg_reduce_kernel, dim3(bpi, n), dim3(SEG_THREADS), 0, s, P, part);
  int32_t rc = unet_check_launch("seg_reduce_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(seg_finalize_kernel, dim3(1), dim3(256), 0, s, (const float*)part, n, c, bpi, (long long)hw,
                     ce_weight, dice_weight, focal_weight, loss, coef);
  rc = unet_check_launch("seg_finalize_kernel");
  if (rc || !dlogits) return rc;
  long long gb = cdiv64(hw, SEG_THREADS * 4);
  if (gb > 1024) gb = 1024;
  hipLaunchKernelGGL(seg_grad_kernel, dim3((unsigned)gb, n), dim3(SEG_THREADS), 0, s, P, (const float*)coef,
                     dice_weight > 0.f ? 1 : 0, dlogits);
  return unet_check_launch("seg_grad_kernel");
}

extern "C" int32_t unet_seg_confusion(const float* logits, const int64_t* target, int32_t n, int32_t c, int64_t hw,
                                      int64_t ignore_index, int64_t* labels, int64_t* confusion, void* stream) {
  UNET_REQUIRE(logits && (labels || (confusion && target)), UNET_ERR_BAD_ARG, "unet_seg_confusion: null pointer");
  UNET_REQUIRE(n > 0 && hw > 0 && c >= 1 && c <= MAXC, UNET_ERR_UNSUPPORTED, "unet_seg_confusion: n=%d c=%d (1..8 classes)", n, c);
  hipStream_t s = (hipStream_t)stream;
  long long gb = cdiv64((long long)n * hw, SEG_THREADS * 8);
  if (gb > 2048) gb = 2048;
  if (gb < 1) gb = 1;
  ProfScope prof(UNET_K_LOSS, 0.0, s);
  hipLaunchKernelGGL(seg_confusion_kernel, dim3((unsigned)gb), dim3(SEG_THREADS), 0, s, logits, (const long long*)target, n,
                     c, (long long)hw, (long long)ignore_index, (long long*)labels, (unsigned long long*)confusion);
  return unet_check_launch("seg_confusion_kernel");
}

// ------------------------------------------------------------------------------------------------------
// Pixel-level threshold epilogue of the anomaly branch (/root/reference/src/test.py:79-106 evaluate_results,
// src/train_utils.py:232-245 validate_epoch): for each of K thresholds, the confusion counts of (anomaly_map > t)
// against (mask > 0.5) over the selected images.  counts[k] = {tp, fp, fn, tn}; integer atomics: exact.
namespace {
__global__ __launch_bounds__(256) void threshold_confusion_kernel(const float* __restrict__ pred, const float* __restrict__ truth,
                                                                  const unsigned char* __restrict__ select, long long per_image,
                                                                  long long total, const float* __restrict__ thr, int K,
                                                                  unsigned long long* __restrict__ counts) {
  __shared__ unsigned int cnt[8 * 4];
  if (threadIdx.x < 32) cnt[threadIdx.x] = 0;
  __syncthreads();
  float t[8];
  for (int k = 0; k < 8; ++k) t[k] = k < K ? thr[k] : 0.f;
  unsigned int loc[8][4] = {};
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    if (select && !select[i / per_image]) continue;
    const float p = pred[i];
    const bool pos = truth[i] > 0.5f;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k < K) loc[k][(p > t[k]) ? (pos ? 0 : 1) : (pos ? 2 : 3)] += 1;
  }
  for (int k = 0; k < K; ++k)
    for (int j = 0; j < 4; ++j)
      if (loc[k][j]) atomicAdd(&cnt[k * 4 + j], loc[k][j]);
  __syncthreads();
  if (threadIdx.x < K * 4 && cnt[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)cnt[threadIdx.x]);
}
}  // namespace

extern "C" int32_t unet_threshold_confusion(const float* pred, const float* truth, const uint8_t* select, int64_t n_images,
                                            int64_t per_image, const float* thresholds, int32_t k, int64_t* counts,
                                            void* stream) {
  UNET_REQUIRE(pred && truth && thresholds && counts, UNET_ERR_BAD_ARG, "unet_threshold_confusion: null pointer");
  UNET_REQUIRE(n_images > 0 && per_image > 0 && k >= 1 && k <= 8, UNET_ERR_UNSUPPORTED,
               "unet_threshold_confusion: %lld images, %d thresholds (1..8)", (long long)n_images, k);
  const long long total = n_images * per_image;
  long long gb = cdiv64(total, 256 * 8);
  if (gb > 2048) gb = 2048;
  hipLaunchKernelGGL(threshold_confusion_kernel, dim3((unsigned)gb), dim3(256), 0, (hipStream_t)stream, pred, truth, select,
                     (long long)per_image, total, thresholds, k, (unsigned long long*)counts);
  return unet_check_launch("threshold_confusion_kernel");
}
