// Loss heads of the hot path on gfx950 (fp32 NCHW planes, HBM-bound / tiny):
//   * CombinedLoss (/root/reference/src/train_utils.py:30-44): MSE + focal(alpha, gamma) on probabilities,
//     values AND gradients in one pass (the gradients are what total_loss.backward() would produce).
//   * SSIMLoss (src/train_utils.py:67-104): separable 11-tap Gaussian (sigma 1.5) through LDS tiles with
//     zero padding; forward also emits the four adjoint maps, backward blurs them (the window is symmetric,
//     so the adjoint of the zero-padded correlation is the same correlation).
// Reductions: block partials -> ordered fp64 finalize (deterministic, no atomics).
#include "common.h"

namespace {

constexpr int LOSS_BLOCKS = 1024;

__device__ inline float block_sum(float v, float* red) {
  // wave reduce then 4-wave combine, fixed order
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void mse_focal_kernel(const float* __restrict__ recon,
                                                        const float* __restrict__ image, long long nr,
                                                        const float* __restrict__ amap,
                                                        const float* __restrict__ mask, long long na, float alpha,
                                                        float gamma, float* __restrict__ d_recon,
                                                        float* __restrict__ d_amap, float* __restrict__ part) {
  __shared__ float red[4];
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long t0 = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  float s_mse = 0.f, s_foc = 0.f;
  const float inv_nr = 1.f / (float)nr, inv_na = 1.f / (float)na;
  for (long long i = t0; i < nr; i += stride) {
    const float d = recon[i] - image[i];
    s_mse = fmaf(d, d, s_mse);
    d_recon[i] = 2.f * d * inv_nr;
  }
  for (long long i = t0; i < na; i += stride) {
    const float p = amap[i], t = mask[i];
    const float lp = fmaxf(logf(p), -100.f), l1p = fmaxf(logf(1.f - p), -100.f);   // ATen BCE clamp
    const float bce = -(t * lp + (1.f - t) * l1p);
    const float pt = expf(-bce);
    const float om = 1.f - pt;
    const float omg = (gamma == 2.f) ? om * om : powf(om, gamma);
    const float omg1 = (gamma == 2.f) ? om : powf(om, gamma - 1.f);
    s_foc += alpha * omg * bce;
    const float df_dbce = alpha * (omg + gamma * omg1 * pt * bce);
    const float dbce_dp = (p - t) / fmaxf((1.f - p) * p, 1e-12f);                   // ATen BCE backward clamp
    d_amap[i] = df_dbce * dbce_dp * inv_na;
  }
  const float a = block_sum(s_mse, red);
  const float b = block_sum(s_foc, red);
  if (threadIdx.x == 0) { part[2 * blockIdx.x] = a; part[2 * blockIdx.x + 1] = b; }
}

// 2 waves: wave q sums the block partials of loss q (64 lanes, then a fixed-order butterfly) in fp64
__global__ __launch_bounds__(128) void mse_focal_finalize_kernel(const float* __restrict__ part, int nblocks,
                                                                 double nr, double na, float* __restrict__ losses) {
  const int q = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double s = 0.0;
  for (int k = lane; k < nblocks; k += 64) s += (double)part[2 * k + q];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
  if (lane == 0) losses[q] = (float)(s / (q == 0 ? nr : na));
}

// ------------------------------------------------------------------------------------ SSIM
constexpr int TS = 32;          // output tile
constexpr int MAXW = 15;        // max window
struct Win { float g[MAXW]; int n; };

// stage a (TS+2p)x(TS+2p) zero-padded patch of one plane into LDS
__device__ inline void load_patch(const float* __restrict__ plane, int H, int W, int y0, int x0, int p,
                                  float* dst, int PW) {
  const int n = PW * PW;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int r = i / PW, c = i - r * PW;
    const int y = y0 - p + r, x = x0 - p + c;
    dst[i] = (y >= 0 && y < H && x >= 0 && x < W) ? plane[(long long)y * W + x] : 0.f;
  }
}

__global__ __launch_bounds__(256) void ssim_fwd_kernel(const float* __restrict__ img1,
                                                       const float* __restrict__ img2, int H, int W, Win win,
                                                       float dl_ds, float* __restrict__ maps,
                                                       long long map_stride, float* __restrict__ part) {
  extern __shared__ float sm[];
  const int p = win.n / 2, PW = TS + 2 * p;
  float* sx = sm;                     // [PW][PW]
  float* sy = sx + PW * PW;
  float* sh = sy + PW * PW;           // [5][PW][TS]
  __shared__ float red[4];
  const int tiles_x = (W + TS - 1) / TS, tiles_y = (H + TS - 1) / TS;
  int b = blockIdx.x;
  const int txi = b % tiles_x;  b /= tiles_x;
  const int tyi = b % tiles_y;
  const long long plane = b / tiles_y;
  const int y0 = tyi * TS, x0 = txi * TS;
  const float* p1 = img1 + plane * H * (long long)W;
  const float* p2 = img2 + plane * H * (long long)W;
  load_patch(p1, H, W, y0, x0, p, sx, PW);
  load_patch(p2, H, W, y0, x0, p, sy, PW);
  __syncthreads();
  // horizontal pass
  for (int i = threadIdx.x; i < PW * TS; i += 256) {
    const int r = i / TS, c = i - r * TS;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
    for (int k = 0; k < win.n; ++k) {
      const float g = win.g[k], x = sx[r * PW + c + k], y = sy[r * PW + c + k];
      a0 = fmaf(g, x, a0); a1 = fmaf(g, y, a1); a2 = fmaf(g, x * x, a2); a3 = fmaf(g, y * y, a3);
      a4 = fmaf(g, x * y, a4);
    }
    sh[0 * PW * TS + i] = a0; sh[1 * PW * TS + i] = a1; sh[2 * PW * TS + i] = a2; sh[3 * PW * TS + i] = a3;
    sh[4 * PW * TS + i] = a4;
  }
  __syncthreads();
  float local = 0.f;
  const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
  for (int i = threadIdx.x; i < TS * TS; i += 256) {
    const int r = i / TS, c = i - r * TS;
    const int y = y0 + r, x = x0 + c;
    if (y >= H || x >= W) continue;
    float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
    for (int k = 0; k < win.n; ++k) {
      const float g = win.g[k];
      const int o = (r + k) * TS + c;
      m1 = fmaf(g, sh[o], m1); m2 = fmaf(g, sh[PW * TS + o], m2); e11 = fmaf(g, sh[2 * PW * TS + o], e11);
      e22 = fmaf(g, sh[3 * PW * TS + o], e22); e12 = fmaf(g, sh[4 * PW * TS + o], e12);
    }
    const float s11 = e11 - m1 * m1, s22 = e22 - m2 * m2, s12 = e12 - m1 * m2;
    const float A1 = 2.f * m1 * m2 + C1, A2 = 2.f * s12 + C2, B1 = m1 * m1 + m2 * m2 + C1, B2 = s11 + s22 + C2;
    const float inv = 1.f / (B1 * B2);
    const float S = A1 * A2 * inv;
    local += S;
    if (maps) {
      const long long o = plane * H * (long long)W + (long long)y * W + x;
      const float dmu1 = (2.f * m2 * (A2 - A1)) * inv - S * (2.f * m1 / B1 - 2.f * m1 / B2);
      const float dmu2 = (2.f * m1 * (A2 - A1)) * inv - S * (2.f * m2 / B1 - 2.f * m2 / B2);
      maps[o] = dl_ds * dmu1;
      maps[map_stride + o] = dl_ds * dmu2;
      maps[2 * map_stride + o] = dl_ds * (-S / B2);          // d/d e11 == d/d e22
      maps[3 * map_stride + o] = dl_ds * (2.f * A1 * inv);   // d/d e12
    }
  }
  const float tot = block_sum(local, red);
  if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

__global__ void ssim_finalize_kernel(const float* __restrict__ part, int nblocks, double total,
                                     float* __restrict__ loss) {
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int k = 0; k < nblocks; ++k) s += (double)part[k];
    loss[0] = (float)(1.0 - s / total);
  }
}

// d_img1 = G*(gmu1) + 2*img1*G*(ge) + img2*G*(g12);  d_img2 = G*(gmu2) + 2*img2*G*(ge) + img1*G*(g12)
__global__ __launch_bounds__(256) void ssim_bwd_kernel(const float* __restrict__ img1,
                                                       const float* __restrict__ img2, int H, int W, Win win,
                                                       const float* __restrict__ maps, long long map_stride,
                                                       float* __restrict__ d1, float* __restrict__ d2) {
  extern __shared__ float sm[];
  const int p = win.n / 2, PW = TS + 2 * p;
  float* sp = sm;                      // [4][PW][PW]
  float* sh = sp + 4 * PW * PW;        // [4][PW][TS]
  const int tiles_x = (W + TS - 1) / TS, tiles_y = (H + TS - 1) / TS;
  int b = blockIdx.x;
  const int txi = b % tiles_x;  b /= tiles_x;
  const int tyi = b % tiles_y;
  const long long plane = b / tiles_y;
  const int y0 = tyi * TS, x0 = txi * TS;
  for (int m = 0; m < 4; ++m)
    load_patch(maps + m * map_stride + plane * H * (long long)W, H, W, y0, x0, p, sp + m * PW * PW, PW);
  __syncthreads();
  for (int i = threadIdx.x; i < PW * TS; i += 256) {
    const int r = i / TS, c = i - r * TS;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < win.n; ++k) {
      const float g = win.g[k];
#pragma unroll
      for (int m = 0; m < 4; ++m) a[m] = fmaf(g, sp[m * PW * PW + r * PW + c + k], a[m]);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) sh[m * PW * TS + i] = a[m];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < TS * TS; i += 256) {
    const int r = i / TS, c = i - r * TS;
    const int y = y0 + r, x = x0 + c;
    if (y >= H || x >= W) continue;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < win.n; ++k) {
      const float g = win.g[k];
#pragma unroll
      for (int m = 0; m < 4; ++m) a[m] = fmaf(g, sh[m * PW * TS + (r + k) * TS + c], a[m]);
    }
    const long long o = plane * H * (long long)W + (long long)y * W + x;
    const float u = img1[o], v = img2[o];
    if (d1) d1[o] = a[0] + 2.f * u * a[2] + v * a[3];
    if (d2) d2[o] = a[1] + 2.f * v * a[2] + u * a[3];
  }
}

Win make_window(int n) {
  // SSIMLoss.gaussian (train_utils.py:57-59): fp32 taps of exp(-(x-n//2)^2 / (2*1.5^2)), divided by their fp32 sum
  Win w;
  w.n = n;
  float sum = 0.f;
  for (int i = 0; i < n; ++i) {
    const double d = (double)(i - n / 2);
    w.g[i] = (float)exp(-(d * d) / (2.0 * 1.5 * 1.5));
    sum += w.g[i];
  }
  for (int i = 0; i < n; ++i) w.g[i] /= sum;
  for (int i = n; i < MAXW; ++i) w.g[i] = 0.f;
  return w;
}

}  // namespace

extern "C" size_t unet_loss_workspace(int64_t elems) {
  (void)elems;
  return (size_t)LOSS_BLOCKS * 2 * sizeof(float);
}

extern "C" int32_t unet_loss_mse_focal(const float* recon, const float* image, int64_t n_recon,
                                       const float* amap, const float* mask, int64_t n_amap, float alpha,
                                       float gamma, float* losses, float* d_recon, float* d_amap,
                                       void* workspace, size_t workspace_bytes, void* stream) {
  UNET_REQUIRE(recon && image && amap && mask && losses && d_recon && d_amap && workspace, UNET_ERR_BAD_ARG,
               "unet_loss_mse_focal: null pointer");
  UNET_REQUIRE(n_recon > 0 && n_amap > 0, UNET_ERR_BAD_ARG, "unet_loss_mse_focal: empty input");
  UNET_REQUIRE(workspace_bytes >= unet_loss_workspace(0), UNET_ERR_WORKSPACE, "unet_loss_mse_focal: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(UNET_K_LOSS, 0.0, s);
  const long long m = n_recon > n_amap ? n_recon : n_amap;
  const int nb = (int)std::min<long long>(cdiv64(m, 256), LOSS_BLOCKS);
  hipLaunchKernelGGL(mse_focal_kernel, dim3(nb), dim3(256), 0, s, recon, image, (long long)n_recon, amap, mask,
                     (long long)n_amap, alpha, gamma, d_recon, d_amap, (float*)workspace);
  int32_t rc = unet_check_launch("mse_focal_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(mse_focal_finalize_kernel, dim3(1), dim3(128), 0, s, (const float*)workspace, nb,
                     (double)n_recon, (double)n_amap, losses);
  return unet_check_launch("mse_focal_finalize_kernel");
}

static long long ssim_blocks(int planes, int h, int w) {
  return (long long)planes * ((h + TS - 1) / TS) * ((w + TS - 1) / TS);
}

extern "C" size_t unet_ssim_workspace(int32_t planes, int32_t h, int32_t w) {
  return (size_t)4 * planes * h * w * sizeof(float) + (size_t)ssim_blocks(planes, h, w) * sizeof(float);
}

extern "C" int32_t unet_ssim_loss(const float* img1, const float* img2, int32_t planes, int32_t h, int32_t w,
                                  int32_t window, float* loss, float* d_img1, float* d_img2, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  UNET_REQUIRE(img1 && img2 && loss && workspace, UNET_ERR_BAD_ARG, "unet_ssim_loss: null pointer");
  UNET_REQUIRE(planes > 0 && h > 0 && w > 0, UNET_ERR_BAD_ARG, "unet_ssim_loss: bad dims");
  UNET_REQUIRE(window >= 1 && window <= MAXW && (window & 1), UNET_ERR_UNSUPPORTED,
               "unet_ssim_loss: window %d (odd, <= %d)", window, MAXW);
  UNET_REQUIRE(workspace_bytes >= unet_ssim_workspace(planes, h, w), UNET_ERR_WORKSPACE,
               "unet_ssim_loss: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(UNET_K_LOSS, 0.0, s);
  const Win win = make_window(window);
  const long long nb = ssim_blocks(planes, h, w);
  const long long map_stride = (long long)planes * h * w;
  float* maps = (float*)workspace;
  float* part = maps + 4 * map_stride;
  const int PW = TS + 2 * (window / 2);
  const double total = (double)planes * h * w;
  const bool need_grad = d_img1 || d_img2;
  const size_t lds_f = (size_t)(2 * PW * PW + 5 * PW * TS) * sizeof(float);
  hipLaunchKernelGGL(ssim_fwd_kernel, dim3((unsigned)nb), dim3(256), lds_f, s, img1, img2, h, w, win,
                     (float)(-1.0 / total), need_grad ? maps : nullptr, map_stride, part);
  int32_t rc = unet_check_launch("ssim_fwd_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(ssim_finalize_kernel, dim3(1), dim3(64), 0, s, (const float*)part, (int)nb, total, loss);
  rc = unet_check_launch("ssim_finalize_kernel");
  if (rc || !need_grad) return rc;
  const size_t lds_b = (size_t)(4 * PW * PW + 4 * PW * TS) * sizeof(float);
  hipLaunchKernelGGL(ssim_bwd_kernel, dim3((unsigned)nb), dim3(256), lds_b, s, img1, img2, h, w, win,
                     (const float*)maps, map_stride, d_img1, d_img2);
  return unet_check_launch("ssim_bwd_kernel");
}

// SSIMLoss(size_average=False) (train_utils.py:84-87): loss[i] = 1 - mean over (C, H, W) of image i's SSIM map;
// d_img1 / d_img2 hold d loss[i] / d img for image i (the caller scales image i's slice by its upstream gradient).
// One pass of the same kernels per image (planes = channels): the workspace of ONE image is reused in stream order.
extern "C" int32_t unet_ssim_loss_per_image(const float* img1, const float* img2, int32_t n, int32_t c, int32_t h, int32_t w,
                                            int32_t window, float* loss, float* d_img1, float* d_img2, void* workspace,
                                            size_t workspace_bytes, void* stream) {
  UNET_REQUIRE(n > 0 && c > 0, UNET_ERR_BAD_ARG, "unet_ssim_loss_per_image: bad dims");
  const size_t img = (size_t)c * h * w;
  for (int i = 0; i < n; ++i) {
    const int32_t rc = unet_ssim_loss(img1 + i * img, img2 + i * img, c, h, w, window, loss + i,
                                      d_img1 ? d_img1 + i * img : nullptr, d_img2 ? d_img2 + i * img : nullptr, workspace,
                                      workspace_bytes, stream);
    if (rc) return rc;
  }
  return UNET_OK;
}
