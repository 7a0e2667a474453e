// BatchNorm2d (+ReLU) statistics, apply and backward for NHWC tensors on gfx950.
// (nn.BatchNorm2d / nn.ReLU at /root/reference/src/model.py:15-16,18-19; semantics pinned in
//  biased variance for normalisation, unbiased for the running estimate.)
//
// All kernels are HBM-bound streaming passes with 16-byte vector accesses.  Per-channel reductions are
// two-stage and ordered (block partials -> fixed-order finalize in fp64), so results are bitwise
// reproducible -- no float atomics.
#include "common.h"

namespace {

constexpr int MAX_PARTS = 512;

// MODE 0: sum y, sum y^2       MODE 1: sum dz, sum dz*yhat (dz = da*[fma(y,scale,shift) > 0])     MODE 2: sum x
template <typename T, int MODE>
__global__ __launch_bounds__(256) void colreduce_kernel(const T* __restrict__ y, const T* __restrict__ da,
                                                        long long pixels, int C, int rows_per_block,
                                                        const float* __restrict__ scale,
                                                        const float* __restrict__ shift,
                                                        const float* __restrict__ mean,
                                                        const float* __restrict__ istd,
                                                        float* __restrict__ part) {
  constexpr int PIECE = ET<T>::PIECE;        // channels per thread
  constexpr int TPR = 64 / PIECE;            // threads per 64-channel row slice
  constexpr int RPI = 256 / TPR;             // rows per iteration
  __shared__ float red[2][RPI][64 + 1];
  const int tc = threadIdx.x % TPR, tr = threadIdx.x / TPR;
  const int c0 = blockIdx.x * 64 + tc * PIECE;
  const long long r_begin = (long long)blockIdx.y * rows_per_block;
  long long r_end = r_begin + rows_per_block;
  if (r_end > pixels) r_end = pixels;

  float s0[PIECE], s1[PIECE], sc[PIECE], sh[PIECE], mu[PIECE], is[PIECE];
#pragma unroll
  for (int j = 0; j < PIECE; ++j) { s0[j] = 0.f; s1[j] = 0.f; }
  if (MODE == 1) {
#pragma unroll
    for (int j = 0; j < PIECE; ++j) { sc[j] = scale[c0 + j]; sh[j] = shift[c0 + j]; mu[j] = mean[c0 + j]; is[j] = istd[c0 + j]; }
  }
  auto accum = [&](const float (&v)[PIECE], const float (&g)[PIECE]) {
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < PIECE; ++j) { s0[j] += v[j]; s1[j] = fmaf(v[j], v[j], s1[j]); }
    } else if (MODE == 1) {
#pragma unroll
      for (int j = 0; j < PIECE; ++j) {
        const float z = fmaf(v[j], sc[j], sh[j]);
        const float dz = z > 0.f ? g[j] : 0.f;
        s0[j] += dz;
        s1[j] = fmaf(dz, (v[j] - mu[j]) * is[j], s1[j]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < PIECE; ++j) s0[j] += v[j];
    }
  };
  long long r = r_begin + tr;
  for (; r + 3 * RPI < r_end; r += 4 * RPI) {      // 4 independent 16-byte loads in flight per tensor
    float v[4][PIECE], g[4][PIECE];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      Vec<T>::load(y + (r + u * RPI) * C + c0, v[u]);
      if (MODE == 1) Vec<T>::load(da + (r + u * RPI) * C + c0, g[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) accum(v[u], g[u]);
  }
  for (; r < r_end; r += RPI) {
    float v[PIECE], g[PIECE];
    Vec<T>::load(y + r * C + c0, v);
    if (MODE == 1) Vec<T>::load(da + r * C + c0, g);
    accum(v, g);
  }
#pragma unroll
  for (int j = 0; j < PIECE; ++j) { red[0][tr][tc * PIECE + j] = s0[j]; red[1][tr][tc * PIECE + j] = s1[j]; }
  __syncthreads();
  if (threadIdx.x < 128) {
    const int q = threadIdx.x >> 6, c = threadIdx.x & 63;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < RPI; ++i) s += red[q][i][c];
    part[((size_t)blockIdx.y * 2 + q) * C + blockIdx.x * 64 + c] = s;
  }
}

// ordered fp64 sum of the block partials: block = FC channels x FL row-lanes (FL*FC = 256)
constexpr int FL = 16, FC = 16;
template <int TL, int TC>
__device__ inline void sum_parts_t(const float* part, int nparts, int C, int c, int rl, double (&red)[2][TL][TC],
                                   double& a, double& b) {
  // lane rl takes partials rl, rl+TL, ... (independent loads), then a FIXED binary tree over the TL lanes through LDS
  // (log2(TL) steps instead of one thread walking TL doubles: these kernels are pure latency)
  double s0 = 0.0, s1 = 0.0;
  int k = rl;
  for (; k + 3 * TL < nparts; k += 4 * TL) {
    const float a0 = part[((size_t)k * 2 + 0) * C + c], b0 = part[((size_t)k * 2 + 1) * C + c];
    const float a1 = part[((size_t)(k + TL) * 2 + 0) * C + c], b1 = part[((size_t)(k + TL) * 2 + 1) * C + c];
    const float a2 = part[((size_t)(k + 2 * TL) * 2 + 0) * C + c], b2 = part[((size_t)(k + 2 * TL) * 2 + 1) * C + c];
    const float a3 = part[((size_t)(k + 3 * TL) * 2 + 0) * C + c], b3 = part[((size_t)(k + 3 * TL) * 2 + 1) * C + c];
    s0 += (double)a0; s0 += (double)a1; s0 += (double)a2; s0 += (double)a3;
    s1 += (double)b0; s1 += (double)b1; s1 += (double)b2; s1 += (double)b3;
  }
  for (; k < nparts; k += TL) {
    s0 += (double)part[((size_t)k * 2 + 0) * C + c];
    s1 += (double)part[((size_t)k * 2 + 1) * C + c];
  }
  const int cc = c % TC;
  red[0][rl][cc] = s0;
  red[1][rl][cc] = s1;
  __syncthreads();
#pragma unroll
  for (int st = TL / 2; st >= 1; st >>= 1) {
    if (rl < st) {
      red[0][rl][cc] += red[0][rl + st][cc];
      red[1][rl][cc] += red[1][rl + st][cc];
    }
    __syncthreads();
  }
  a = red[0][0][cc];
  b = red[1][0][cc];
}
__device__ inline void sum_parts(const float* part, int nparts, int C, int c, int rl, double (&red)[2][FL][FC],
                                 double& a, double& b) {
  sum_parts_t<FL, FC>(part, nparts, C, c, rl, red, a, b);
}

// TL x TC = 256: (16,16) for few partials, (64,4) when a conv epilogue produced thousands of them
template <int TL, int TC>
__global__ __launch_bounds__(256) void bn_finalize_train_kernel(
    const float* __restrict__ part, int nparts, int C, double count, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* running_mean, float* running_var, float momentum, float eps,
    float* __restrict__ save_mean, float* __restrict__ save_istd, float* __restrict__ scale,
    float* __restrict__ shift) {
  __shared__ double red[2][TL][TC];
  const int c = blockIdx.x * TC + (threadIdx.x % TC), rl = threadIdx.x / TC;
  double s, ss;
  sum_parts_t<TL, TC>(part, nparts, C, c, rl, red, s, ss);
  if (rl != 0) return;
  const double mean = s / count;
  double var = ss / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float istd = (float)(1.0 / sqrt(var + (double)eps));
  const float m = (float)mean;
  save_mean[c] = m;
  save_istd[c] = istd;
  const float sc = gamma[c] * istd;
  scale[c] = sc;
  shift[c] = fmaf(-m, sc, beta[c]);
  if (running_mean) {
    const double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

// dgamma, dbeta and the three per-channel coefficients of  dy = A*dz + B*y + K
template <int TL, int TC>
__global__ __launch_bounds__(256) void bn_finalize_bwd_kernel(const float* __restrict__ part, int nparts, int C,
                                                              float inv_count, const float* __restrict__ gamma,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ istd,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              float* __restrict__ coefs, int raw) {
  __shared__ double red[2][TL][TC];
  const int c = blockIdx.x * TC + (threadIdx.x % TC), rl = threadIdx.x / TC;
  double s, ss;
  sum_parts_t<TL, TC>(part, nparts, C, c, rl, red, s, ss);
  if (rl != 0) return;
  if (raw & 1) ss *= (double)istd[c];      // partials hold sum dz * (y - mean): normalise here
  const float db = (float)s, dg = (float)ss;
  dbeta[c] = db;
  dgamma[c] = dg;
  const float A = gamma[c] * istd[c];
  // bit 1: frozen statistics (BatchNorm in eval mode inside a training graph): mean / istd are constants, the
  // two batch-statistics terms of the input gradient vanish
  const float B = (raw & 2) ? 0.f : -A * istd[c] * dg * inv_count;
  coefs[c] = A;
  coefs[C + c] = B;
  coefs[2 * C + c] = (raw & 2) ? 0.f : -A * db * inv_count - B * mean[c];
}

__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* __restrict__ part, int nparts, int C,
                                                              float* __restrict__ out) {
  __shared__ double red[2][FL][FC];
  const int c = blockIdx.x * FC + (threadIdx.x % FC), rl = threadIdx.x / FC;
  double s, ss;
  sum_parts(part, nparts, C, c, rl, red, s, ss);
  if (rl == 0) out[c] = (float)s;
}

__global__ void bn_eval_coeffs_kernel(int C, const float* gamma, const float* beta, const float* rm,
                                      const float* rv, float eps, float* scale, float* shift, float* mean_out,
                                      float* istd_out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float istd = (float)(1.0 / sqrt((double)rv[c] + (double)eps));
  const float sc = gamma[c] * istd;
  scale[c] = sc;
  shift[c] = fmaf(-rm[c], sc, beta[c]);
  if (mean_out) { mean_out[c] = rm[c]; istd_out[c] = istd; }
}

// FAST: (256*PIECE) % C == 0, so a thread's channel group never changes along the grid-stride loop and its
// coefficients live in registers; 2 independent 16-byte accesses in flight.  Otherwise a generic modulo path.
template <typename T, bool FAST>
__global__ __launch_bounds__(256) void bn_relu_apply_kernel(const T* __restrict__ y, long long pieces, int C,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift, T* __restrict__ a, int mode) {
  constexpr int PIECE = ET<T>::PIECE;
  const long long stride = (long long)gridDim.x * 256;
  long long i = blockIdx.x * 256LL + threadIdx.x;
  if (FAST) {
    const int c0 = (threadIdx.x * PIECE) % C;
    float sc[PIECE], sh[PIECE];
#pragma unroll
    for (int j = 0; j < PIECE; ++j) { sc[j] = scale[c0 + j]; sh[j] = shift[c0 + j]; }
    if (mode >= 2) {
      // block-contiguous walk, four 4-KiB rows of the block in flight (see bn_bwd_apply_premasked_kernel)
      const long long per = (pieces + gridDim.x - 1) / gridDim.x;
      const long long chunk = (per + 1023) / 1024 * 1024;
      long long b0 = (long long)blockIdx.x * chunk, b1 = b0 + chunk;
      if (b1 > pieces) b1 = pieces;
      for (long long q = b0 + threadIdx.x; q < b1; q += 1024) {
        float v[4][PIECE];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (q + u * 256 < b1) {
            if (mode == 3) Vec<T>::load_nt(y + (q + u * 256) * PIECE, v[u]);
            else Vec<T>::load(y + (q + u * 256) * PIECE, v[u]);
          }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (q + u * 256 < b1) {
#pragma unroll
            for (int j = 0; j < PIECE; ++j) v[u][j] = fmaxf(fmaf(v[u][j], sc[j], sh[j]), 0.f);
            Vec<T>::store(a + (q + u * 256) * PIECE, v[u]);
          }
      }
      return;
    }
    for (; i + stride < pieces; i += 2 * stride) {
      float v0[PIECE], v1[PIECE];
      if (mode == 1) {
        Vec<T>::load_nt(y + i * PIECE, v0);
        Vec<T>::load_nt(y + (i + stride) * PIECE, v1);
      } else {
        Vec<T>::load(y + i * PIECE, v0);
        Vec<T>::load(y + (i + stride) * PIECE, v1);
      }
#pragma unroll
      for (int j = 0; j < PIECE; ++j) { v0[j] = fmaxf(fmaf(v0[j], sc[j], sh[j]), 0.f); v1[j] = fmaxf(fmaf(v1[j], sc[j], sh[j]), 0.f); }
      Vec<T>::store(a + i * PIECE, v0);
      Vec<T>::store(a + (i + stride) * PIECE, v1);
    }
    for (; i < pieces; i += stride) {
      float v[PIECE];
      Vec<T>::load(y + i * PIECE, v);
#pragma unroll
      for (int j = 0; j < PIECE; ++j) v[j] = fmaxf(fmaf(v[j], sc[j], sh[j]), 0.f);
      Vec<T>::store(a + i * PIECE, v);
    }
  } else {
    const int groups = C / PIECE;
    for (; i < pieces; i += stride) {
      const int c0 = (int)(i % groups) * PIECE;
      float v[PIECE];
      Vec<T>::load(y + i * PIECE, v);
#pragma unroll
      for (int j = 0; j < PIECE; ++j) v[j] = fmaxf(fmaf(v[j], scale[c0 + j], shift[c0 + j]), 0.f);
      Vec<T>::store(a + i * PIECE, v);
    }
  }
}

// dy = A*dz + B*y + K   (== gamma*istd * (dz - dbeta/M - yhat*dgamma/M)),  dz = da*[fma(y,scale,shift) > 0]
template <typename T, bool FAST>
__global__ __launch_bounds__(256) void bn_relu_bwd_apply_kernel(
    const T* __restrict__ da, const T* __restrict__ y, long long pieces, int C, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ coefs, T* __restrict__ dy) {
  constexpr int PIECE = ET<T>::PIECE;
  const long long stride = (long long)gridDim.x * 256;
  long long i = blockIdx.x * 256LL + threadIdx.x;
  if (FAST) {
    const int c0 = (threadIdx.x * PIECE) % C;
    float sc[PIECE], sh[PIECE], A[PIECE], B[PIECE], K[PIECE];
#pragma unroll
    for (int j = 0; j < PIECE; ++j) {
      sc[j] = scale[c0 + j]; sh[j] = shift[c0 + j];
      A[j] = coefs[c0 + j]; B[j] = coefs[C + c0 + j]; K[j] = coefs[2 * C + c0 + j];
    }
    for (; i + stride < pieces; i += 2 * stride) {
      float v0[PIECE], g0[PIECE], v1[PIECE], g1[PIECE];
      Vec<T>::load(y + i * PIECE, v0);
      Vec<T>::load(da + i * PIECE, g0);
      Vec<T>::load(y + (i + stride) * PIECE, v1);
      Vec<T>::load(da + (i + stride) * PIECE, g1);
#pragma unroll
      for (int j = 0; j < PIECE; ++j) {
        const float d0 = fmaf(v0[j], sc[j], sh[j]) > 0.f ? g0[j] : 0.f;
        const float d1 = fmaf(v1[j], sc[j], sh[j]) > 0.f ? g1[j] : 0.f;
        g0[j] = fmaf(A[j], d0, fmaf(B[j], v0[j], K[j]));
        g1[j] = fmaf(A[j], d1, fmaf(B[j], v1[j], K[j]));
      }
      Vec<T>::store(dy + i * PIECE, g0);
      Vec<T>::store(dy + (i + stride) * PIECE, g1);
    }
    for (; i < pieces; i += stride) {
      float v[PIECE], g[PIECE];
      Vec<T>::load(y + i * PIECE, v);
      Vec<T>::load(da + i * PIECE, g);
#pragma unroll
      for (int j = 0; j < PIECE; ++j) {
        const float d = fmaf(v[j], sc[j], sh[j]) > 0.f ? g[j] : 0.f;
        g[j] = fmaf(A[j], d, fmaf(B[j], v[j], K[j]));
      }
      Vec<T>::store(dy + i * PIECE, g);
    }
  } else {
    const int groups = C / PIECE;
    for (; i < pieces; i += stride) {
      const int c0 = (int)(i % groups) * PIECE;
      float v[PIECE], g[PIECE];
      Vec<T>::load(y + i * PIECE, v);
      Vec<T>::load(da + i * PIECE, g);
#pragma unroll
      for (int j = 0; j < PIECE; ++j) {
        const int c = c0 + j;
        const float d = fmaf(v[j], scale[c], shift[c]) > 0.f ? g[j] : 0.f;
        g[j] = fmaf(coefs[c], d, fmaf(coefs[C + c], v[j], coefs[2 * C + c]));
      }
      Vec<T>::store(dy + i * PIECE, g);
    }
  }
}

// dy = A*dz + B*y + K for a gradient that already carries the ReLU mask (dz written by the producer's epilogue:
// head_bwd_kernel<BN>, conv dgrad with fused mask).  dy may alias dz (element-wise in place).
template <typename T, bool FAST>
__global__ __launch_bounds__(256) void bn_bwd_apply_premasked_kernel(const T* dz, const T* __restrict__ y,
                                                                     long long pieces, int C,
                                                                     const float* __restrict__ coefs, T* dy, int mode) {
  constexpr int PIECE = ET<T>::PIECE;
  const long long stride = (long long)gridDim.x * 256;
  long long i = blockIdx.x * 256LL + threadIdx.x;
  if (FAST) {
    const int c0 = (threadIdx.x * PIECE) % C;
    float A[PIECE], B[PIECE], K[PIECE];
#pragma unroll
    for (int j = 0; j < PIECE; ++j) { A[j] = coefs[c0 + j]; B[j] = coefs[C + c0 + j]; K[j] = coefs[2 * C + c0 + j]; }
    if (mode >= 2) {
      // block-contiguous walk: a block streams 4 x 4 KiB of each tensor per iteration from ONE region
      const long long per = (pieces + gridDim.x - 1) / gridDim.x;
      const long long chunk = (per + 1023) / 1024 * 1024;
      long long b0 = (long long)blockIdx.x * chunk, b1 = b0 + chunk;
      if (b1 > pieces) b1 = pieces;
      for (long long q = b0 + threadIdx.x; q < b1; q += 1024) {
        float v[4][PIECE], g[4][PIECE];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (q + u * 256 < b1) {
            if (mode == 3) Vec<T>::load_nt(y + (q + u * 256) * PIECE, v[u]);
            else Vec<T>::load(y + (q + u * 256) * PIECE, v[u]);
            Vec<T>::load(dz + (q + u * 256) * PIECE, g[u]);
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (q + u * 256 < b1) {
#pragma unroll
            for (int j = 0; j < PIECE; ++j) g[u][j] = fmaf(A[j], g[u][j], fmaf(B[j], v[u][j], K[j]));
            Vec<T>::store(dy + (q + u * 256) * PIECE, g[u]);
          }
        }
      }
      return;
    }
    for (; i + 3 * stride < pieces; i += 4 * stride) {
      float v[4][PIECE], g[4][PIECE];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (mode == 1) Vec<T>::load_nt(y + (i + u * stride) * PIECE, v[u]);
        else Vec<T>::load(y + (i + u * stride) * PIECE, v[u]);
        Vec<T>::load(dz + (i + u * stride) * PIECE, g[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int j = 0; j < PIECE; ++j) g[u][j] = fmaf(A[j], g[u][j], fmaf(B[j], v[u][j], K[j]));
        Vec<T>::store(dy + (i + u * stride) * PIECE, g[u]);
      }
    }
    for (; i < pieces; i += stride) {
      float v[PIECE], g[PIECE];
      Vec<T>::load(y + i * PIECE, v);
      Vec<T>::load(dz + i * PIECE, g);
#pragma unroll
      for (int j = 0; j < PIECE; ++j) g[j] = fmaf(A[j], g[j], fmaf(B[j], v[j], K[j]));
      Vec<T>::store(dy + i * PIECE, g);
    }
  } else {
    const int groups = C / PIECE;
    for (; i < pieces; i += stride) {
      const int c0 = (int)(i % groups) * PIECE;
      float v[PIECE], g[PIECE];
      Vec<T>::load(y + i * PIECE, v);
      Vec<T>::load(dz + i * PIECE, g);
#pragma unroll
      for (int j = 0; j < PIECE; ++j) {
        const int c = c0 + j;
        g[j] = fmaf(coefs[c], g[j], fmaf(coefs[C + c], v[j], coefs[2 * C + c]));
      }
      Vec<T>::store(dy + i * PIECE, g);
    }
  }
}

struct RedPlan { int nparts, rows_per_block; };
RedPlan red_plan(long long pixels, int C) {
  long long want = 1024 / (C / 64);
  if (want < 1) want = 1;
  if (want > MAX_PARTS) want = MAX_PARTS;
  long long rpb = cdiv64(pixels, want);
  rpb = cdiv64(rpb, 32) * 32;
  RedPlan p;
  p.rows_per_block = (int)rpb;
  p.nparts = (int)cdiv64(pixels, rpb);
  return p;
}

template <typename T, int MODE>
int32_t launch_reduce(const void* y, const void* da, long long pixels, int C, const float* scale,
                      const float* shift, const float* mean, const float* istd, float* part, RedPlan pl,
                      hipStream_t s) {
  hipLaunchKernelGGL((colreduce_kernel<T, MODE>), dim3(C / 64, pl.nparts), dim3(256), 0, s, (const T*)y,
                     (const T*)da, pixels, C, pl.rows_per_block, scale, shift, mean, istd, part);
  return unet_check_launch("colreduce_kernel");
}

inline int ew_blocks(long long pieces) { return (int)std::min<long long>(cdiv64(pieces, 512), 256 * 8); }

// Walk of the apply passes: UNET_EW_VAR = 1 (streaming / nontemporal loads of y), 2 (every block streams ONE contiguous
// region, four 4-KiB rows in flight), 3 (both); default 0 = grid-stride walk, plain loads.  Kept as hooks with a negative
// result (round 4, profiles/r04_experiments.txt): looping ONE kernel over the same three 256-MiB tensors, mode 3 reads
// 6.6 instead of 4.6 TB/s -- because the gradient tensor then stays resident in the 256-MB Infinity Cache from one
// iteration to the next; inside the step, where the operands arrive from other kernels, the same-box A/B is +-0
// (bn class 3.10 vs 3.10 ms).  A single-kernel loop is not evidence for an HBM-bound pass whose working set is near
// the cache size.
inline int ew_mode(long long elements) {
  (void)elements;
  const char v = unet_tuning().ew_var;
  return (v >= '1' && v <= '3') ? v - '0' : 0;
}

}  // namespace

extern "C" size_t unet_bn_workspace(int64_t pixels, int32_t c) {
  (void)pixels;
  return ((size_t)MAX_PARTS * 2 + 3) * (size_t)c * sizeof(float);
}

int32_t unet_internal_colsum(int dtype, const void* x, int64_t pixels, int C, float* out, float* ws,
                             size_t ws_bytes, hipStream_t s) {
  UNET_REQUIRE(C % 64 == 0, UNET_ERR_UNSUPPORTED, "colsum: channels %d not a multiple of 64", C);
  RedPlan pl = red_plan(pixels, C);
  const size_t per = (size_t)2 * C * sizeof(float);
  if ((size_t)pl.nparts * per > ws_bytes) {
    const long long fit = (long long)(ws_bytes / per);
    UNET_REQUIRE(fit >= 1, UNET_ERR_WORKSPACE, "colsum: workspace %zu too small", ws_bytes);
    pl.rows_per_block = (int)(cdiv64(cdiv64(pixels, fit), 32) * 32);
    pl.nparts = (int)cdiv64(pixels, pl.rows_per_block);
  }
  int32_t rc = dtype == UNET_BF16
                   ? launch_reduce<bf16_t, 2>(x, nullptr, pixels, C, nullptr, nullptr, nullptr, nullptr, ws, pl, s)
                   : launch_reduce<float, 2>(x, nullptr, pixels, C, nullptr, nullptr, nullptr, nullptr, ws, pl, s);
  if (rc) return rc;
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3(C / FC), dim3(256), 0, s, ws, pl.nparts, C, out);
  return unet_check_launch("colsum_finalize_kernel");
}

int32_t unet_internal_bn_partials(int dtype, const void* y, int64_t pixels, int C, float* part, int* n_parts,
                                  hipStream_t s) {
  UNET_REQUIRE(C % 64 == 0, UNET_ERR_UNSUPPORTED, "bn partials: channels %d not a multiple of 64", C);
  const RedPlan pl = red_plan(pixels, C);
  ProfScope prof(UNET_K_BN, 0.0, s);
  *n_parts = pl.nparts;
  return dtype == UNET_BF16
             ? launch_reduce<bf16_t, 0>(y, nullptr, pixels, C, nullptr, nullptr, nullptr, nullptr, part, pl, s)
             : launch_reduce<float, 0>(y, nullptr, pixels, C, nullptr, nullptr, nullptr, nullptr, part, pl, s);
}

extern "C" int32_t unet_bn_finalize_partials(const float* partial, int32_t n_parts, int64_t pixels, int32_t c,
                                             const float* gamma, const float* beta, float* running_mean,
                                             float* running_var, float momentum, float eps, float* save_mean,
                                             float* save_istd, float* scale, float* shift, void* stream) {
  UNET_REQUIRE(partial && gamma && beta && save_mean && save_istd && scale && shift, UNET_ERR_BAD_ARG,
               "unet_bn_finalize_partials: null pointer");
  UNET_REQUIRE(n_parts > 0 && pixels > 0 && c > 0 && c % FC == 0, UNET_ERR_BAD_ARG, "unet_bn_finalize_partials: dims");
  UNET_REQUIRE((running_mean == nullptr) == (running_var == nullptr), UNET_ERR_BAD_ARG,
               "unet_bn_finalize_partials: running_mean/var must both be given or both NULL");
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(UNET_K_BN, 0.0, s);
  if (n_parts >= 128)      // 64 partial-lanes per channel: fewer dependent loads per thread on this latency-bound kernel
    hipLaunchKernelGGL((bn_finalize_train_kernel<64, 4>), dim3(c / 4), dim3(256), 0, s, partial, n_parts, c,
                       (double)pixels, gamma, beta, running_mean, running_var, momentum, eps, save_mean, save_istd,
                       scale, shift);
  else
    hipLaunchKernelGGL((bn_finalize_train_kernel<FL, FC>), dim3(c / FC), dim3(256), 0, s, partial, n_parts, c,
                       (double)pixels, gamma, beta, running_mean, running_var, momentum, eps, save_mean, save_istd,
                       scale, shift);
  return unet_check_launch("bn_finalize_train_kernel");
}

extern "C" int32_t unet_bn_train_stats(int32_t dtype, const void* y, int64_t pixels, int32_t c,
                                       const float* gamma, const float* beta, float* running_mean,
                                       float* running_var, float momentum, float eps, float* save_mean,
                                       float* save_istd, float* scale, float* shift, void* workspace,
                                       size_t workspace_bytes, void* stream) {
  UNET_REQUIRE(y && gamma && beta && save_mean && save_istd && scale && shift && workspace, UNET_ERR_BAD_ARG,
               "unet_bn_train_stats: null pointer");
  UNET_REQUIRE(pixels > 0 && c > 0 && c % 64 == 0, UNET_ERR_UNSUPPORTED, "unet_bn_train_stats: c=%d pixels=%lld",
               c, (long long)pixels);
  UNET_REQUIRE((running_mean == nullptr) == (running_var == nullptr), UNET_ERR_BAD_ARG,
               "unet_bn_train_stats: running_mean/var must both be given or both NULL");
  const RedPlan pl = red_plan(pixels, c);
  UNET_REQUIRE(workspace_bytes >= (size_t)pl.nparts * 2 * c * sizeof(float), UNET_ERR_WORKSPACE,
               "unet_bn_train_stats: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(UNET_K_BN, 0.0, s);
  float* part = (float*)workspace;
  int32_t rc = dtype == UNET_BF16
                   ? launch_reduce<bf16_t, 0>(y, nullptr, pixels, c, nullptr, nullptr, nullptr, nullptr, part, pl, s)
                   : launch_reduce<float, 0>(y, nullptr, pixels, c, nullptr, nullptr, nullptr, nullptr, part, pl, s);
  if (rc) return rc;
  hipLaunchKernelGGL((bn_finalize_train_kernel<FL, FC>), dim3(c / FC), dim3(256), 0, s, part, pl.nparts, c,
                     (double)pixels, gamma, beta, running_mean, running_var, momentum, eps, save_mean, save_istd, scale,
                     shift);
  return unet_check_launch("bn_finalize_train_kernel");
}

extern "C" int32_t unet_bn_eval_coeffs(int32_t c, const float* gamma, const float* beta,
                                       const float* running_mean, const float* running_var, float eps,
                                       float* scale, float* shift, void* stream) {
  UNET_REQUIRE(gamma && beta && running_mean && running_var && scale && shift && c > 0, UNET_ERR_BAD_ARG,
               "unet_bn_eval_coeffs: bad argument");
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(cdiv(c, 256)), dim3(256), 0, (hipStream_t)stream, c, gamma, beta,
                     running_mean, running_var, eps, scale, shift, (float*)nullptr, (float*)nullptr);
  return unet_check_launch("bn_eval_coeffs_kernel");
}

extern "C" int32_t unet_bn_eval_coeffs4(int32_t c, const float* gamma, const float* beta, const float* running_mean,
                                        const float* running_var, float eps, float* mean, float* istd, float* scale,
                                        float* shift, void* stream) {
  UNET_REQUIRE(gamma && beta && running_mean && running_var && mean && istd && scale && shift && c > 0, UNET_ERR_BAD_ARG,
               "unet_bn_eval_coeffs4: bad argument");
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(cdiv(c, 256)), dim3(256), 0, (hipStream_t)stream, c, gamma, beta,
                     running_mean, running_var, eps, scale, shift, mean, istd);
  return unet_check_launch("bn_eval_coeffs_kernel");
}

extern "C" int32_t unet_bn_relu_apply(int32_t dtype, const void* y, int64_t pixels, int32_t c,
                                      const float* scale, const float* shift, void* a, void* stream) {
  UNET_REQUIRE(y && scale && shift && a, UNET_ERR_BAD_ARG, "unet_bn_relu_apply: null pointer");
  UNET_REQUIRE(pixels > 0 && c > 0 && c % 8 == 0, UNET_ERR_UNSUPPORTED, "unet_bn_relu_apply: c=%d", c);
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(UNET_K_BN, 0.0, s);
  const int ewm = ew_mode(pixels * (long long)c);
  if (dtype == UNET_BF16) {
    const long long pieces = pixels * c / 8;
    if ((256 * 8) % c == 0)
      hipLaunchKernelGGL((bn_relu_apply_kernel<bf16_t, true>), dim3(ew_blocks(pieces)), dim3(256), 0, s,
                         (const bf16_t*)y, pieces, c, scale, shift, (bf16_t*)a, ewm);
    else
      hipLaunchKernelGGL((bn_relu_apply_kernel<bf16_t, false>), dim3(ew_blocks(pieces)), dim3(256), 0, s,
                         (const bf16_t*)y, pieces, c, scale, shift, (bf16_t*)a, ewm);
  } else {
    const long long pieces = pixels * c / 4;
    if ((256 * 4) % c == 0)
      hipLaunchKernelGGL((bn_relu_apply_kernel<float, true>), dim3(ew_blocks(pieces)), dim3(256), 0, s,
                         (const float*)y, pieces, c, scale, shift, (float*)a, ewm);
    else
      hipLaunchKernelGGL((bn_relu_apply_kernel<float, false>), dim3(ew_blocks(pieces)), dim3(256), 0, s,
                         (const float*)y, pieces, c, scale, shift, (float*)a, ewm);
  }
  return unet_check_launch("bn_relu_apply_kernel");
}

static int32_t bn_relu_bwd_impl(int32_t dtype, const void* da, const void* y, int64_t pixels, int32_t c,
                                const float* gamma, const float* save_mean, const float* save_istd,
                                const float* scale, const float* shift, float* dgamma, float* dbeta,
                                void* dy, void* workspace, size_t workspace_bytes, void* stream, int frozen) {
  UNET_REQUIRE(da && y && gamma && save_mean && save_istd && scale && shift && dgamma && dbeta && dy && workspace,
               UNET_ERR_BAD_ARG, "unet_bn_relu_bwd: null pointer");
  UNET_REQUIRE(pixels > 0 && c > 0 && c % 64 == 0, UNET_ERR_UNSUPPORTED, "unet_bn_relu_bwd: c=%d", c);
  const RedPlan pl = red_plan(pixels, c);
  const size_t part_bytes = (size_t)pl.nparts * 2 * c * sizeof(float);
  UNET_REQUIRE(workspace_bytes >= part_bytes + (size_t)3 * c * sizeof(float), UNET_ERR_WORKSPACE,
               "unet_bn_relu_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(UNET_K_BN, 0.0, s);
  float* part = (float*)workspace;
  float* coefs = part + (size_t)pl.nparts * 2 * c;
  int32_t rc = dtype == UNET_BF16
                   ? launch_reduce<bf16_t, 1>(y, da, pixels, c, scale, shift, save_mean, save_istd, part, pl, s)
                   : launch_reduce<float, 1>(y, da, pixels, c, scale, shift, save_mean, save_istd, part, pl, s);
  if (rc) return rc;
  const float inv = (float)(1.0 / (double)pixels);
  if (pl.nparts >= 128)      // latency-bound: 64 partial-lanes per channel
    hipLaunchKernelGGL((bn_finalize_bwd_kernel<64, 4>), dim3(c / 4), dim3(256), 0, s, part, pl.nparts, c, inv, gamma,
                       save_mean, save_istd, dgamma, dbeta, coefs, frozen ? 2 : 0);
  else
    hipLaunchKernelGGL((bn_finalize_bwd_kernel<FL, FC>), dim3(c / FC), dim3(256), 0, s, part, pl.nparts, c, inv, gamma,
                       save_mean, save_istd, dgamma, dbeta, coefs, frozen ? 2 : 0);
  rc = unet_check_launch("bn_finalize_bwd_kernel");
  if (rc) return rc;
  if (dtype == UNET_BF16) {
    const long long pieces = pixels * c / 8;
    if ((256 * 8) % c == 0)
      hipLaunchKernelGGL((bn_relu_bwd_apply_kernel<bf16_t, true>), dim3(ew_blocks(pieces)), dim3(256), 0, s,
                         (const bf16_t*)da, (const bf16_t*)y, pieces, c, scale, shift, coefs, (bf16_t*)dy);
    else
      hipLaunchKernelGGL((bn_relu_bwd_apply_kernel<bf16_t, false>), dim3(ew_blocks(pieces)), dim3(256), 0, s,
                         (const bf16_t*)da, (const bf16_t*)y, pieces, c, scale, shift, coefs, (bf16_t*)dy);
  } else {
    const long long pieces = pixels * c / 4;
    if ((256 * 4) % c == 0)
      hipLaunchKernelGGL((bn_relu_bwd_apply_kernel<float, true>), dim3(ew_blocks(pieces)), dim3(256), 0, s,
                         (const float*)da, (const float*)y, pieces, c, scale, shift, coefs, (float*)dy);
    else
      hipLaunchKernelGGL((bn_relu_bwd_apply_kernel<float, false>), dim3(ew_blocks(pieces)), dim3(256), 0, s,
                         (const float*)da, (const float*)y, pieces, c, scale, shift, coefs, (float*)dy);
  }
  return unet_check_launch("bn_relu_bwd_apply_kernel");
}

extern "C" int32_t unet_bn_relu_bwd(int32_t dtype, const void* da, const void* y, int64_t pixels, int32_t c,
                                    const float* gamma, const float* save_mean, const float* save_istd,
                                    const float* scale, const float* shift, float* dgamma, float* dbeta,
                                    void* dy, void* workspace, size_t workspace_bytes, void* stream) {
  return bn_relu_bwd_impl(dtype, da, y, pixels, c, gamma, save_mean, save_istd, scale, shift, dgamma, dbeta, dy, workspace,
                          workspace_bytes, stream, 0);
}

extern "C" int32_t unet_bn_relu_bwd_frozen(int32_t dtype, const void* da, const void* y, int64_t pixels, int32_t c,
                                           const float* gamma, const float* mean, const float* istd, const float* scale,
                                           const float* shift, float* dgamma, float* dbeta, void* dy, void* workspace,
                                           size_t workspace_bytes, void* stream) {
  return bn_relu_bwd_impl(dtype, da, y, pixels, c, gamma, mean, istd, scale, shift, dgamma, dbeta, dy, workspace,
                          workspace_bytes, stream, 1);
}

extern "C" int32_t unet_bn_bwd_premasked(int32_t dtype, const void* dz, const void* y, int64_t pixels, int32_t c,
                                         const float* gamma, const float* save_mean, const float* save_istd,
                                         const float* partial, int32_t n_parts, float* dgamma, float* dbeta, void* dy,
                                         void* workspace, size_t workspace_bytes, void* stream) {
  UNET_REQUIRE(gamma && save_mean && save_istd && partial && dgamma && dbeta && workspace && (!dy || (dz && y)),
               UNET_ERR_BAD_ARG, "unet_bn_bwd_premasked: null pointer");
  UNET_REQUIRE(pixels > 0 && c > 0 && c % FC == 0 && n_parts > 0, UNET_ERR_UNSUPPORTED, "unet_bn_bwd_premasked: c=%d", c);
  UNET_REQUIRE(workspace_bytes >= (size_t)3 * c * sizeof(float), UNET_ERR_WORKSPACE,
               "unet_bn_bwd_premasked: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(UNET_K_BN, 0.0, s);
  float* coefs = (float*)workspace;
  const float inv = (float)(1.0 / (double)pixels);
  if (n_parts >= 128)
    hipLaunchKernelGGL((bn_finalize_bwd_kernel<64, 4>), dim3(c / 4), dim3(256), 0, s, partial, n_parts, c, inv, gamma,
                       save_mean, save_istd, dgamma, dbeta, coefs, 1);
  else
    hipLaunchKernelGGL((bn_finalize_bwd_kernel<FL, FC>), dim3(c / FC), dim3(256), 0, s, partial, n_parts, c, inv, gamma,
                       save_mean, save_istd, dgamma, dbeta, coefs, 1);
  int32_t rc = unet_check_launch("bn_finalize_bwd_kernel");
  if (rc) return rc;
  if (!dy) return UNET_OK;       // coefficients only: the consumer of dy applies them itself (unet_conv3x3_first_wgrad_bn)
  const int ewm = ew_mode(pixels * (long long)c);
  if (dtype == UNET_BF16) {
    const long long pieces = pixels * c / 8;
    if ((256 * 8) % c == 0)
      hipLaunchKernelGGL((bn_bwd_apply_premasked_kernel<bf16_t, true>), dim3(ew_blocks(pieces)), dim3(256), 0, s,
                         (const bf16_t*)dz, (const bf16_t*)y, pieces, c, coefs, (bf16_t*)dy, ewm);
    else
      hipLaunchKernelGGL((bn_bwd_apply_premasked_kernel<bf16_t, false>), dim3(ew_blocks(pieces)), dim3(256), 0, s,
                         (const bf16_t*)dz, (const bf16_t*)y, pieces, c, coefs, (bf16_t*)dy, ewm);
  } else if (dtype == UNET_F32) {
    const long long pieces = pixels * c / 4;
    if ((256 * 4) % c == 0)
      hipLaunchKernelGGL((bn_bwd_apply_premasked_kernel<float, true>), dim3(ew_blocks(pieces)), dim3(256), 0, s,
                         (const float*)dz, (const float*)y, pieces, c, coefs, (float*)dy, ewm);
    else
      hipLaunchKernelGGL((bn_bwd_apply_premasked_kernel<float, false>), dim3(ew_blocks(pieces)), dim3(256), 0, s,
                         (const float*)dz, (const float*)y, pieces, c, coefs, (float*)dy, ewm);
  } else {
    unet_set_error("unet_bn_bwd_premasked: dtype %d", dtype);
    return UNET_ERR_BAD_ARG;
  }
  return unet_check_launch("bn_bwd_apply_premasked_kernel");
}
