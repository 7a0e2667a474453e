// Input pipeline on the GPU: the image transforms of the reference's loaders
//   /root/reference/src/dataset.py:134-146            Resize -> RandomHorizontalFlip -> RandomRotation(10) -> ColorJitter -> ToTensor -> Normalize
//   /root/reference/src/kolektorsdd_dataset.py:133-150 the same with RandomRotation(5); masks Resize(NEAREST)
// for batches of DECODED uint8 images resident in HBM.  The reference runs torchvision on PIL images, so every stage here
// restates the integer / float arithmetic of Pillow's C kernels (Resample.c ImagingResample*_8bpc, Geometry.c
// affine_fixed / ImagingScaleAffine, Blend.c, Convert.c rgb2l / rgb2hsv / hsv2rgb) and is bit-exact against fixtures
// produced by PIL itself (tests/golden/aug_*.npz, tools/make_goldens_aug.py).  The random DRAWS (flip, angle, jitter
// factors and order) stay with the host, which passes them in as per-image parameters.
//
// All kernels are tiny HBM streams next to a training step (a 32 x 1024 x 1024 x 3 batch is 100 MB in, 6 MB out): one
// thread per output pixel, no LDS.  Floating-point stages are written with explicit non-contracted operations
// (__fmul_rn / __fadd_rn ...): PIL's C code is compiled without fused multiply-adds and a contracted a*b+c rounds once
// instead of twice.
#include <math.h>

#include <algorithm>

#include "common.h"

// no fused multiply-adds anywhere in this file, host or device: PIL's arithmetic rounds every product and every sum
#pragma clang fp contract(off)

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;      // Resample.c

// ---------------------------------------------------------------------------------------------- bilinear resize
// One axis of ImagingResample for 8-bit pixels: out = clip8((2^21 + sum_k in[xmin + k] * kk[k]) >> 22).
// AXIS 1: along x (src [n][h][w][c] -> dst [n][h][ow][c]); AXIS 0: along y (src [n][h][w][c] -> dst [n][oh][w][c]).
template <int AXIS>
__global__ __launch_bounds__(256) void resample_u8_kernel(const unsigned char* __restrict__ src,
                                                          unsigned char* __restrict__ dst, int N, int H, int W, int C,
                                                          int OUT, const int* __restrict__ bounds,
                                                          const int* __restrict__ kk, int ksize) {
  const int OH = AXIS == 0 ? OUT : H, OW = AXIS == 1 ? OUT : W;
  const long long total = (long long)N * OH * OW * C;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C);
    long long t = i / C;
    const int x = (int)(t % OW);  t /= OW;
    const int y = (int)(t % OH);
    const long long n = t / OH;
    const int o = AXIS == 0 ? y : x;
    const int lo = bounds[2 * o], cnt = bounds[2 * o + 1];
    const int* k = kk + (long long)o * ksize;
    int ss = 1 << (PRECISION_BITS - 1);
    if (AXIS == 1) {
      const unsigned char* p = src + ((n * H + y) * W + lo) * C + c;
      for (int j = 0; j < cnt; ++j) ss += (int)p[(long long)j * C] * k[j];
    } else {
      const unsigned char* p = src + ((n * H + lo) * W + x) * C + c;
      for (int j = 0; j < cnt; ++j) ss += (int)p[(long long)j * W * C] * k[j];
    }
    ss >>= PRECISION_BITS;
    dst[i] = (unsigned char)(ss < 0 ? 0 : (ss > 255 ? 255 : ss));
  }
}

// Image.resize(NEAREST) (ImagingScaleAffine): source indices per output row / column from the host's tables, -1 = outside
__global__ __launch_bounds__(256) void resize_nearest_u8_kernel(const unsigned char* __restrict__ src,
                                                                unsigned char* __restrict__ dst, int N, int H, int W, int C,
                                                                int OH, int OW, const int* __restrict__ yidx,
                                                                const int* __restrict__ xidx) {
  const long long total = (long long)N * OH * OW * C;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C);
    long long t = i / C;
    const int x = (int)(t % OW);  t /= OW;
    const int y = (int)(t % OH);
    const long long n = t / OH;
    const int sy = yidx[y], sx = xidx[x];
    dst[i] = (sy >= 0 && sx >= 0) ? src[((n * H + sy) * W + sx) * C + c] : (unsigned char)0;
  }
}

// ---------------------------------------------------------------------------------------------- flip + rotation
// RandomHorizontalFlip then Image.rotate(angle, NEAREST, fill 0): Geometry.c affine_fixed in 16.16 fixed point.  m[n] =
// {a0, a1, a2, a3, a4, a5} already in fixed point (FIX(a) = floor(a * 65536 + 0.5), a2 / a5 with the half-pixel terms
// folded in, exactly as affine_fixed forms them -- the host computes them in double like PIL's Python + C do).
__global__ __launch_bounds__(256) void flip_rotate_u8_kernel(const unsigned char* __restrict__ src,
                                                             unsigned char* __restrict__ dst, int N, int H, int W, int C,
                                                             const unsigned char* __restrict__ flip,
                                                             const int* __restrict__ m) {
  const long long total = (long long)N * H * W;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int x = (int)(i % W);
    const long long r = i / W;
    const int y = (int)(r % H);
    const long long n = r / H;
    int xin = x, yin = y;
    bool ok = true;
    if (m) {
      const int* a = m + n * 6;
      // (affine_fixed accumulates xx += a0 per pixel and a2 += a1 per row in 32-bit ints: sums of exact integers)
      const int xx = a[2] + a[1] * y + a[0] * x;
      const int yy = a[5] + a[4] * y + a[3] * x;
      xin = xx >> 16;
      yin = yy >> 16;
      ok = xin >= 0 && xin < W && yin >= 0 && yin < H;
    }
    if (ok && flip && flip[n]) xin = W - 1 - xin;       // the rotation reads the FLIPPED image
    unsigned char* o = dst + i * C;
    if (ok) {
      const unsigned char* p = src + ((n * H + yin) * W + xin) * C;
      for (int c = 0; c < C; ++c) o[c] = p[c];
    } else {
      for (int c = 0; c < C; ++c) o[c] = 0;
    }
  }
}

// ---------------------------------------------------------------------------------------------- ColorJitter
struct RGB { int r, g, b; };

__device__ __forceinline__ int l24(const RGB& p) {            // Convert.c rgb2l
  return (p.r * 19595 + p.g * 38470 + p.b * 7471 + 0x8000) >> 16;
}

// Blend.c ImagingBlend(degenerate, image, alpha) for one band value: float32 arithmetic, two roundings
__device__ __forceinline__ int blend1(int deg, int v, float alpha) {
  if (alpha == 1.0f) return v;
  if (alpha == 0.0f) return deg;
  const float t = __fadd_rn((float)deg, __fmul_rn(alpha, (float)(v - deg)));
  if (alpha >= 0.0f && alpha <= 1.0f) return (int)t & 255;                  // (UINT8) of an in-range value
  return t <= 0.0f ? 0 : (t >= 255.0f ? 255 : (int)t);
}

__device__ __forceinline__ RGB blend3(int dr, int dg, int db, const RGB& p, float alpha) {
  RGB o;
  o.r = blend1(dr, p.r, alpha);
  o.g = blend1(dg, p.g, alpha);
  o.b = blend1(db, p.b, alpha);
  return o;
}

// adjust_hue: RGB -> HSV (Convert.c rgb2hsv_row), h += shift (uint8 wrap), HSV -> RGB (hsv2rgb_row)
__device__ __forceinline__ RGB hue_shift(const RGB& p, int shift) {
  const int maxc = max(p.r, max(p.g, p.b)), minc = min(p.r, min(p.g, p.b));
  int uh = 0, us = 0;
  const int uv = maxc;
  if (minc != maxc) {
    const float cr = (float)(maxc - minc);
    const float s = __fdiv_rn(cr, (float)maxc);
    const float rc = __fdiv_rn((float)(maxc - p.r), cr);
    const float gc = __fdiv_rn((float)(maxc - p.g), cr);
    const float bc = __fdiv_rn((float)(maxc - p.b), cr);
    float h;
    if (p.r == maxc) h = (float)__dsub_rn((double)bc, (double)gc);
    else if (p.g == maxc) h = (float)__dsub_rn(__dadd_rn(2.0, (double)rc), (double)bc);
    else h = (float)__dsub_rn(__dadd_rn(4.0, (double)gc), (double)rc);
    h = (float)fmod(__dadd_rn(__ddiv_rn((double)h, 6.0), 1.0), 1.0);
    const int ih = (int)__dmul_rn((double)h, 255.0), is = (int)__dmul_rn((double)s, 255.0);
    uh = ih < 0 ? 0 : (ih > 255 ? 255 : ih);
    us = is < 0 ? 0 : (is > 255 ? 255 : is);
  }
  uh = (uh + shift) & 255;
  RGB o;
  if (us == 0) { o.r = o.g = o.b = uv; return o; }
  const float hf = __fdiv_rn(__fmul_rn((float)uh, 6.0f), 255.0f);
  const float fi = floorf(hf);
  const float f = __fsub_rn(hf, fi);
  const float fs = __fdiv_rn((float)us, 255.0f);
  const float vf = (float)uv;
  auto rnd = [](float x) { const int q = (int)floorf(__fadd_rn(x, 0.5f)); return q < 0 ? 0 : (q > 255 ? 255 : q); };
  const int pp = rnd(__fmul_rn(vf, __fsub_rn(1.0f, fs)));
  const int qq = rnd(__fmul_rn(vf, __fsub_rn(1.0f, __fmul_rn(fs, f))));
  const int tt = rnd(__fmul_rn(vf, __fsub_rn(1.0f, __fmul_rn(fs, __fsub_rn(1.0f, f)))));
  switch (((int)fi) % 6) {
    case 0: o.r = uv; o.g = tt; o.b = pp; break;
    case 1: o.r = qq; o.g = uv; o.b = pp; break;
    case 2: o.r = pp; o.g = uv; o.b = tt; break;
    case 3: o.r = pp; o.g = qq; o.b = uv; break;
    case 4: o.r = tt; o.g = pp; o.b = uv; break;
    default: o.r = uv; o.g = pp; o.b = qq; break;
  }
  return o;
}

// ops [first, last) of the image's jitter list applied to one pixel; `cmean`: the contrast operation's grey level
__device__ __forceinline__ RGB jitter_ops(RGB p, const unet_jitter_desc& d, int first, int last, int cmean) {
  for (int k = first; k < last; ++k) {
    const int op = d.order[k];
    if (op == 0) p = blend3(0, 0, 0, p, d.brightness);
    else if (op == 1) p = blend3(cmean, cmean, cmean, p, d.contrast);
    else if (op == 2) { const int l = l24(p); p = blend3(l, l, l, p, d.saturation); }
    else if (op == 3) p = hue_shift(p, d.hue_shift);
  }
  return p;
}

__device__ __forceinline__ int contrast_pos(const unet_jitter_desc& d) {
  for (int k = 0; k < 4; ++k)
    if (d.order[k] == 1) return k;
  return -1;
}

// ImageEnhance.Contrast: mean of the L image of the picture AS IT IS when the contrast operation runs: per-image
// integer sum of L over the pixels with the preceding operations applied (integer atomics: exact, order-free)
__global__ __launch_bounds__(256) void jitter_lsum_kernel(const unsigned char* __restrict__ src, int N, int HW,
                                                          const unet_jitter_desc* __restrict__ desc,
                                                          unsigned long long* __restrict__ lsum) {
  const int n = blockIdx.y;
  const unet_jitter_desc d = desc[n];
  const int cp = contrast_pos(d);
  if (cp < 0) return;
  unsigned int acc = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
    const unsigned char* q = src + ((long long)n * HW + i) * 3;
    RGB p = {q[0], q[1], q[2]};
    p = jitter_ops(p, d, 0, cp, 0);
    acc += (unsigned)l24(p);
  }
  __shared__ unsigned int red[256];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(lsum + n, (unsigned long long)red[0]);
}

__global__ __launch_bounds__(256) void jitter_normalize_kernel(const unsigned char* __restrict__ src,
                                                               float* __restrict__ dst, int N, int H, int W,
                                                               const unet_jitter_desc* __restrict__ desc,
                                                               const unsigned long long* __restrict__ lsum, float m0,
                                                               float m1, float m2, float s0, float s1, float s2) {
  const long long plane = (long long)H * W, total = (long long)N * plane;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long n = i / plane, q = i - n * plane;
    const unsigned char* s = src + i * 3;
    RGB p = {s[0], s[1], s[2]};
    if (desc) {
      const unet_jitter_desc d = desc[n];
      // int(ImageStat.Stat(L).mean[0] + 0.5): Python floats are doubles
      const int cmean = (int)__dadd_rn(__ddiv_rn((double)lsum[n], (double)plane), 0.5);
      p = jitter_ops(p, d, 0, 4, cmean);
    }
    float* o = dst + n * 3 * plane + q;
    o[0] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)p.r, 255.f), m0), s0);
    o[plane] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)p.g, 255.f), m1), s1);
    o[2 * plane] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)p.b, 255.f), m2), s2);
  }
}

inline int aug_blocks(long long total) { return (int)std::min<long long>(cdiv64(total, 256), 256 * 16); }

}  // namespace

// ---------------------------------------------------------------------------------------------- host: PIL's tables
extern "C" int32_t unet_resize_bilinear_ksize(int32_t in_size, int32_t out_size) {
  if (in_size <= 0 || out_size <= 0) return 0;
  double filterscale = (double)in_size / (double)out_size;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 1.0 * filterscale;
  return (int32_t)ceil(support) * 2 + 1;
}

// Resample.c precompute_coeffs (bilinear: support 1, filter 1 - |x|) + normalize_coeffs_8bpc, in host doubles.
extern "C" int32_t unet_resize_bilinear_coeffs(int32_t in_size, int32_t out_size, int32_t* bounds, int32_t* kk) {
  UNET_REQUIRE(in_size > 0 && out_size > 0 && bounds && kk, UNET_ERR_BAD_ARG, "unet_resize_bilinear_coeffs: bad argument");
  const double scale = (double)in_size / (double)out_size;
  double filterscale = scale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 1.0 * filterscale;
  const int ksize = (int)ceil(support) * 2 + 1;
  const double ss = 1.0 / filterscale;
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = 0.0 + (xx + 0.5) * scale;
    double ww = 0.0;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double k[64];
    double* kd = k;
    double* big = nullptr;
    if (ksize > 64) { big = new double[ksize]; kd = big; }
    int x = 0;
    for (; x < xmax; ++x) {
      double a = (x + xmin - center + 0.5) * ss;
      if (a < 0.0) a = -a;
      const double w = a < 1.0 ? 1.0 - a : 0.0;
      kd[x] = w;
      ww += w;
    }
    for (x = 0; x < xmax; ++x)
      if (ww != 0.0) kd[x] /= ww;
    for (; x < ksize; ++x) kd[x] = 0.0;
    bounds[2 * xx] = xmin;
    bounds[2 * xx + 1] = xmax;
    for (x = 0; x < ksize; ++x)
      kk[(size_t)xx * ksize + x] = kd[x] < 0.0 ? (int)(-0.5 + kd[x] * (double)(1 << PRECISION_BITS))
                                               : (int)(0.5 + kd[x] * (double)(1 << PRECISION_BITS));
    delete[] big;
  }
  return UNET_OK;
}

// Geometry.c ImagingScaleAffine (Image.resize(NEAREST) over the whole image): xo = a0 * 0.5, then xo += a0 per output
// position -- the accumulated double, not a product -- and COORD(xo) = (int)xo for xo >= 0
extern "C" int32_t unet_resize_nearest_index(int32_t in_size, int32_t out_size, int32_t* idx) {
  UNET_REQUIRE(in_size > 0 && out_size > 0 && idx, UNET_ERR_BAD_ARG, "unet_resize_nearest_index: bad argument");
  const double a0 = (double)in_size / (double)out_size;
  double xo = 0.0 + a0 * 0.5;
  for (int x = 0; x < out_size; ++x) {
    const int xin = xo >= 0.0 ? (int)xo : -1;
    idx[x] = (xin >= 0 && xin < in_size) ? xin : -1;
    xo += a0;
  }
  return UNET_OK;
}

// ---------------------------------------------------------------------------------------------- device entry points
extern "C" int32_t unet_resize_bilinear_u8(const uint8_t* src, int32_t n, int32_t h, int32_t w, int32_t c, int32_t out_h,
                                           int32_t out_w, const int32_t* xbounds, const int32_t* xkk, int32_t xksize,
                                           const int32_t* ybounds, const int32_t* ykk, int32_t yksize, uint8_t* tmp,
                                           uint8_t* dst, void* stream) {
  UNET_REQUIRE(src && dst && n > 0 && h > 0 && w > 0 && (c == 1 || c == 3) && out_h > 0 && out_w > 0, UNET_ERR_BAD_ARG,
               "unet_resize_bilinear_u8: bad argument");
  hipStream_t s = (hipStream_t)stream;
  const bool need_x = out_w != w, need_y = out_h != h;          // (ImagingResample skips a pass that changes nothing)
  UNET_REQUIRE(!need_x || (xbounds && xkk && xksize > 0), UNET_ERR_BAD_ARG, "unet_resize_bilinear_u8: x tables missing");
  UNET_REQUIRE(!need_y || (ybounds && ykk && yksize > 0), UNET_ERR_BAD_ARG, "unet_resize_bilinear_u8: y tables missing");
  UNET_REQUIRE(!(need_x && need_y) || tmp, UNET_ERR_BAD_ARG, "unet_resize_bilinear_u8: tmp [n][h][out_w][c] missing");
  if (!need_x && !need_y) {
    hipError_t e = hipMemcpyAsync(dst, src, (size_t)n * h * w * c, hipMemcpyDeviceToDevice, s);
    UNET_REQUIRE(e == hipSuccess, UNET_ERR_LAUNCH, "unet_resize_bilinear_u8: copy: %s", hipGetErrorString(e));
    return UNET_OK;
  }
  // horizontal pass first (into tmp, 8-bit like PIL's intermediate image), then vertical
  if (need_x) {
    uint8_t* o = need_y ? tmp : dst;
    hipLaunchKernelGGL((resample_u8_kernel<1>), dim3(aug_blocks((long long)n * h * out_w * c)), dim3(256), 0, s, src, o, n, h, w,
                       c, out_w, xbounds, xkk, xksize);
    int32_t rc = unet_check_launch("resample_u8_kernel<x>");
    if (rc) return rc;
  }
  if (need_y) {
    const uint8_t* in = need_x ? tmp : src;
    hipLaunchKernelGGL((resample_u8_kernel<0>), dim3(aug_blocks((long long)n * out_h * out_w * c)), dim3(256), 0, s, in, dst, n,
                       h, out_w, c, out_h, ybounds, ykk, yksize);
    return unet_check_launch("resample_u8_kernel<y>");
  }
  return UNET_OK;
}

extern "C" int32_t unet_resize_nearest_u8(const uint8_t* src, int32_t n, int32_t h, int32_t w, int32_t c, int32_t out_h,
                                          int32_t out_w, const int32_t* yidx, const int32_t* xidx, uint8_t* dst,
                                          void* stream) {
  UNET_REQUIRE(src && dst && yidx && xidx && n > 0 && h > 0 && w > 0 && c > 0 && out_h > 0 && out_w > 0, UNET_ERR_BAD_ARG,
               "unet_resize_nearest_u8: bad argument");
  hipLaunchKernelGGL(resize_nearest_u8_kernel, dim3(aug_blocks((long long)n * out_h * out_w * c)), dim3(256), 0,
                     (hipStream_t)stream, src, dst, n, h, w, c, out_h, out_w, yidx, xidx);
  return unet_check_launch("resize_nearest_u8_kernel");
}

extern "C" int32_t unet_flip_rotate_u8(const uint8_t* src, int32_t n, int32_t h, int32_t w, int32_t c, const uint8_t* flip,
                                       const int32_t* matrices, uint8_t* dst, void* stream) {
  UNET_REQUIRE(src && dst && src != dst && n > 0 && h > 0 && w > 0 && c > 0 && c <= 4, UNET_ERR_BAD_ARG,
               "unet_flip_rotate_u8: bad argument");
  UNET_REQUIRE(h < 32768 && w < 32768, UNET_ERR_UNSUPPORTED, "unet_flip_rotate_u8: 16.16 fixed point needs sides < 32768");
  hipLaunchKernelGGL(flip_rotate_u8_kernel, dim3(aug_blocks((long long)n * h * w)), dim3(256), 0, (hipStream_t)stream, src, dst,
                     n, h, w, c, flip, matrices);
  return unet_check_launch("flip_rotate_u8_kernel");
}

extern "C" size_t unet_color_jitter_workspace(int32_t n) { return (size_t)(n > 0 ? n : 0) * sizeof(unsigned long long); }

extern "C" int32_t unet_color_jitter_normalize_u8(const uint8_t* images_hwc, int32_t n, int32_t h, int32_t w,
                                                  const unet_jitter_desc* desc, const float* mean3, const float* std3,
                                                  float* out_nchw, void* workspace, size_t workspace_bytes, void* stream) {
  UNET_REQUIRE(images_hwc && out_nchw && mean3 && std3 && n > 0 && h > 0 && w > 0, UNET_ERR_BAD_ARG,
               "unet_color_jitter_normalize_u8: bad argument");
  UNET_REQUIRE((long long)h * w < (1LL << 24), UNET_ERR_UNSUPPORTED, "unet_color_jitter_normalize_u8: image too large");
  hipStream_t s = (hipStream_t)stream;
  unsigned long long* lsum = nullptr;
  if (desc) {
    UNET_REQUIRE(workspace && workspace_bytes >= unet_color_jitter_workspace(n), UNET_ERR_WORKSPACE,
                 "unet_color_jitter_normalize_u8: workspace too small");
    lsum = (unsigned long long*)workspace;
    hipError_t e = hipMemsetAsync(lsum, 0, (size_t)n * sizeof(unsigned long long), s);
    UNET_REQUIRE(e == hipSuccess, UNET_ERR_LAUNCH, "unet_color_jitter_normalize_u8: memset: %s", hipGetErrorString(e));
    const int hw = h * w;
    hipLaunchKernelGGL(jitter_lsum_kernel, dim3(std::min(cdiv(hw, 256), 64), n), dim3(256), 0, s, images_hwc, n, hw, desc, lsum);
    int32_t rc = unet_check_launch("jitter_lsum_kernel");
    if (rc) return rc;
  }
  hipLaunchKernelGGL(jitter_normalize_kernel, dim3(aug_blocks((long long)n * h * w)), dim3(256), 0, s, images_hwc, out_nchw, n, h,
                     w, desc, lsum, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
  return unet_check_launch("jitter_normalize_kernel");
}
