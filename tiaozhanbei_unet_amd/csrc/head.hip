// 1x1 output head (OutConv, /root/reference/src/model.py:72) fused with bias and the optional sigmoid
// of AnomalyUNet.forward (src/model.py:201,208).  K = c_in (64), N = c_out (1..8): arithmetic intensity
// ~1-3 FLOP/B, HBM-bound, so this is a VALU streaming kernel, not an MFMA one.
//
// TPP = c_in/PIECE lanes cooperate on one pixel (each loads 16 B of its channels), partial dot products
// are combined with wave shuffles.  Output is written NCHW fp32 (what the loss heads and callers read).
// Backward recomputes nothing: dlogit = dout * out * (1 - out); dx is written NHWC in the compute dtype,
// dW/db are block partials summed in a fixed order (deterministic).
#include "common.h"

namespace {

constexpr int MAXCO = 8;

// (image, pixel-in-image) of a flat pixel index: a 64-bit division costs more than the pixel's arithmetic, so the
// common case (fewer than 2^31 pixels) divides in 32 bits
__device__ __forceinline__ void split_pixel(long long p, long long hw, long long& n, long long& q) {
  if (p < (1LL << 31) && hw < (1LL << 31)) {
    const unsigned nn = (unsigned)p / (unsigned)hw;
    n = nn;
    q = (unsigned)p - nn * (unsigned)hw;
  } else {
    n = p / hw;
    q = p - n * hw;
  }
}

template <typename T, int TPP, int CO>
__global__ __launch_bounds__(256) void head_fwd_kernel(const T* __restrict__ x, long long pixels, long long hw,
                                                       int Cin, const float* __restrict__ w,
                                                       const float* __restrict__ b, int sigm,
                                                       float* __restrict__ out) {
  constexpr int PIECE = ET<T>::PIECE;
  constexpr int U = 4;                  // pixels in flight per thread (memory-level parallelism)
  const int g = threadIdx.x % TPP;
  float wr[CO][PIECE];                  // this lane's slice of the CO filters, in registers
#pragma unroll
  for (int co = 0; co < CO; ++co)
#pragma unroll
    for (int j = 0; j < PIECE; ++j) wr[co][j] = w[co * Cin + g * PIECE + j];
  const long long slot = (blockIdx.x * 256LL + threadIdx.x) / TPP;
  const long long nslots = (long long)gridDim.x * 256 / TPP;
  // every lane of a wave runs the same number of iterations (shuffles need all lanes)
  const long long iters = cdiv64(pixels, nslots * U);
  for (long long it = 0; it < iters; ++it) {
    float v[U][PIECE];
    long long p[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      p[u] = slot + (it * U + u) * nslots;
#pragma unroll
      for (int j = 0; j < PIECE; ++j) v[u][j] = 0.f;
      if (p[u] < pixels) Vec<T>::load(x + p[u] * Cin + g * PIECE, v[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float acc[CO];
#pragma unroll
      for (int co = 0; co < CO; ++co) {
        acc[co] = 0.f;
#pragma unroll
        for (int j = 0; j < PIECE; ++j) acc[co] = fmaf(v[u][j], wr[co][j], acc[co]);
      }
#pragma unroll
      for (int m = 1; m < TPP; m <<= 1)
#pragma unroll
        for (int co = 0; co < CO; ++co) acc[co] += __shfl_xor(acc[co], m);
      if (p[u] < pixels && g < CO) {
        float r = 0.f;
#pragma unroll
        for (int co = 0; co < CO; ++co) if (co == g) r = acc[co];
        r += b[g];
        if (sigm) r = 1.f / (1.f + expf(-r));
        long long n, q;
        split_pixel(p[u], hw, n, q);
        out[(n * CO + g) * hw + q] = r;
      }
    }
  }
}

template <typename T, int TPP, int CO>
__global__ __launch_bounds__(256) void head_bwd_kernel(const T* __restrict__ x, const float* __restrict__ out,
                                                       const float* __restrict__ dout, long long pixels,
                                                       long long hw, int Cin, const float* __restrict__ w,
                                                       int sigm, T* __restrict__ dx, float* __restrict__ part) {
  constexpr int PIECE = ET<T>::PIECE;
  constexpr int U = 4;
  __shared__ float red[4][MAXCO * 129];
  const int g = threadIdx.x % TPP;
  float wr[CO][PIECE];
#pragma unroll
  for (int co = 0; co < CO; ++co)
#pragma unroll
    for (int j = 0; j < PIECE; ++j) wr[co][j] = w[co * Cin + g * PIECE + j];
  const long long slot = (blockIdx.x * 256LL + threadIdx.x) / TPP;
  const long long nslots = (long long)gridDim.x * 256 / TPP;
  const long long iters = cdiv64(pixels, nslots * U);
  float dwacc[CO][PIECE], dbacc[CO];
#pragma unroll
  for (int co = 0; co < CO; ++co) {
    dbacc[co] = 0.f;
#pragma unroll
    for (int j = 0; j < PIECE; ++j) dwacc[co][j] = 0.f;
  }
  for (long long it = 0; it < iters; ++it) {
    float v[U][PIECE], dl[U][CO];
    long long p[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      p[u] = slot + (it * U + u) * nslots;
#pragma unroll
      for (int j = 0; j < PIECE; ++j) v[u][j] = 0.f;
#pragma unroll
      for (int co = 0; co < CO; ++co) dl[u][co] = 0.f;
      if (p[u] < pixels) {
        Vec<T>::load(x + p[u] * Cin + g * PIECE, v[u]);
        long long n, q;
        split_pixel(p[u], hw, n, q);
#pragma unroll
        for (int co = 0; co < CO; ++co) {
          float d = dout[(n * CO + co) * hw + q];
          if (sigm) { const float o = out[(n * CO + co) * hw + q]; d *= o * (1.f - o); }
          dl[u][co] = d;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (p[u] >= pixels) continue;
      float d[PIECE];
#pragma unroll
      for (int j = 0; j < PIECE; ++j) d[j] = 0.f;
#pragma unroll
      for (int co = 0; co < CO; ++co) {
        if (g == 0) dbacc[co] += dl[u][co];
#pragma unroll
        for (int j = 0; j < PIECE; ++j) {
          d[j] = fmaf(dl[u][co], wr[co][j], d[j]);
          dwacc[co][j] = fmaf(dl[u][co], v[u][j], dwacc[co][j]);
        }
      }
      Vec<T>::store(dx + p[u] * Cin + g * PIECE, d);
    }
  }
  // lanes with equal g (stride TPP) hold partials of the same channels: combine across the wave
#pragma unroll
  for (int m = TPP; m < 64; m <<= 1) {
#pragma unroll
    for (int co = 0; co < CO; ++co) {
      dbacc[co] += __shfl_xor(dbacc[co], m);
#pragma unroll
      for (int j = 0; j < PIECE; ++j) dwacc[co][j] += __shfl_xor(dwacc[co][j], m);
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int stride = Cin + 1;
  if (lane < TPP) {
#pragma unroll
    for (int co = 0; co < CO; ++co) {
#pragma unroll
      for (int j = 0; j < PIECE; ++j) red[wave][co * stride + lane * PIECE + j] = dwacc[co][j];
      if (lane == 0) red[wave][co * stride + Cin] = dbacc[co];
    }
  }
  __syncthreads();
  // part[block][co][Cin+1]  (last column = bias gradient)
  for (int i = threadIdx.x; i < CO * stride; i += 256)
    part[(size_t)blockIdx.x * CO * stride + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}

// one block per output element group: 256 threads sum the block partials of 4 outputs (64 lanes each)
__global__ __launch_bounds__(256) void head_bwd_finalize_kernel(const float* __restrict__ part, int nblocks, int CO,
                                                                int Cin, float* __restrict__ dw,
                                                                float* __restrict__ db) {
  const int stride = Cin + 1;
  const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= CO * stride) return;
  double s = 0.0;
  for (int k = lane; k < nblocks; k += 64) s += (double)part[(size_t)k * CO * stride + i];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
  if (lane == 0) {
    const int co = i / stride, c = i % stride;
    if (c < Cin) dw[co * Cin + c] = (float)s;
    else db[co] = (float)s;
  }
}

inline int head_blocks(long long pixels, int tpp) {
  const long long per_block = 256 / tpp;
  long long b = cdiv64(pixels, per_block * 8);   // ~8 pixels per slot
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  return (int)b;
}

template <typename T>
int tpp_of(int c_in) { return c_in / ET<T>::PIECE; }

}  // namespace

#define HEAD_CO_SWITCH(...)                                  \
  switch (c_out) {                                          \
    case 1: { constexpr int CO = 1; __VA_ARGS__; } break;          \
    case 2: { constexpr int CO = 2; __VA_ARGS__; } break;          \
    case 3: { constexpr int CO = 3; __VA_ARGS__; } break;          \
    case 4: { constexpr int CO = 4; __VA_ARGS__; } break;          \
    default: { constexpr int CO = 8; __VA_ARGS__; } break;         \
  }
#define HEAD_TPP_SWITCH(T, tpp, ...)                        \
  switch (tpp) {                                            \
    case 4: { constexpr int TPP = 4; HEAD_CO_SWITCH(__VA_ARGS__); } break;   \
    case 8: { constexpr int TPP = 8; HEAD_CO_SWITCH(__VA_ARGS__); } break;   \
    case 16: { constexpr int TPP = 16; HEAD_CO_SWITCH(__VA_ARGS__); } break; \
    case 32: { constexpr int TPP = 32; HEAD_CO_SWITCH(__VA_ARGS__); } break; \
    default:                                                \
      unet_set_error("head: c_in %d unsupported", c_in);    \
      return UNET_ERR_UNSUPPORTED;                          \
  }

extern "C" int32_t unet_head_fwd(int32_t dtype, const void* x, int32_t n, int32_t h, int32_t w, int32_t c_in,
                                 const float* weight, const float* bias, int32_t c_out, int32_t sigmoid,
                                 float* out, void* stream) {
  UNET_REQUIRE(x && weight && bias && out, UNET_ERR_BAD_ARG, "unet_head_fwd: null pointer");
  UNET_REQUIRE(n > 0 && h > 0 && w > 0, UNET_ERR_BAD_ARG, "unet_head_fwd: bad dims");
  UNET_REQUIRE(c_out >= 1 && c_out <= MAXCO && c_in <= 128 && (c_in == 32 || c_in == 64 || c_in == 128), UNET_ERR_UNSUPPORTED,
               "unet_head_fwd: %d -> %d channels unsupported (c_out <= 8, c_in in {32,64,128})", c_in, c_out);
  UNET_REQUIRE(c_out <= c_in / (dtype == UNET_BF16 ? 8 : 4), UNET_ERR_UNSUPPORTED, "unet_head_fwd: c_out %d too wide for c_in %d", c_out, c_in);
  hipStream_t s = (hipStream_t)stream;
  const long long pixels = (long long)n * h * w, hw = (long long)h * w;
  ProfScope prof(UNET_K_HEAD, 2.0 * pixels * c_in * c_out, s);
  if (dtype == UNET_BF16) {
    const int tpp = tpp_of<bf16_t>(c_in);
    HEAD_TPP_SWITCH(bf16_t, tpp, hipLaunchKernelGGL((head_fwd_kernel<bf16_t, TPP, CO>), dim3(head_blocks(pixels, TPP)), dim3(256), 0, s,
                    (const bf16_t*)x, pixels, hw, c_in, weight, bias, sigmoid, out));
  } else {
    const int tpp = tpp_of<float>(c_in);
    HEAD_TPP_SWITCH(float, tpp, hipLaunchKernelGGL((head_fwd_kernel<float, TPP, CO>), dim3(head_blocks(pixels, TPP)), dim3(256), 0, s,
                    (const float*)x, pixels, hw, c_in, weight, bias, sigmoid, out));
  }
  return unet_check_launch("head_fwd_kernel");
}

extern "C" size_t unet_head_bwd_workspace(int32_t n, int32_t h, int32_t w, int32_t c_in, int32_t c_out) {
  (void)n; (void)h; (void)w;
  return (size_t)2048 * c_out * (c_in + 1) * sizeof(float);
}

extern "C" int32_t unet_head_bwd(int32_t dtype, const void* x, const float* out, const float* dout, int32_t n,
                                 int32_t h, int32_t w, int32_t c_in, const float* weight, int32_t c_out,
                                 int32_t sigmoid, void* dx, float* dweight, float* dbias, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  UNET_REQUIRE(x && dout && weight && dx && dweight && dbias && workspace && (out || !sigmoid), UNET_ERR_BAD_ARG,
               "unet_head_bwd: null pointer");
  UNET_REQUIRE(n > 0 && h > 0 && w > 0, UNET_ERR_BAD_ARG, "unet_head_bwd: bad dims");
  UNET_REQUIRE(c_out >= 1 && c_out <= MAXCO && c_in <= 128 && (c_in == 32 || c_in == 64 || c_in == 128), UNET_ERR_UNSUPPORTED,
               "unet_head_bwd: %d -> %d channels unsupported", c_in, c_out);
  UNET_REQUIRE(workspace_bytes >= unet_head_bwd_workspace(n, h, w, c_in, c_out), UNET_ERR_WORKSPACE,
               "unet_head_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const long long pixels = (long long)n * h * w, hw = (long long)h * w;
  ProfScope prof(UNET_K_HEAD, 4.0 * pixels * c_in * c_out, s);
  int nb = 0;
  if (dtype == UNET_BF16) {
    const int tpp = tpp_of<bf16_t>(c_in);
    nb = head_blocks(pixels, tpp);
    HEAD_TPP_SWITCH(bf16_t, tpp, hipLaunchKernelGGL((head_bwd_kernel<bf16_t, TPP, CO>), dim3(nb), dim3(256), 0, s,
                    (const bf16_t*)x, out, dout, pixels, hw, c_in, weight, sigmoid, (bf16_t*)dx, (float*)workspace));
  } else {
    const int tpp = tpp_of<float>(c_in);
    nb = head_blocks(pixels, tpp);
    HEAD_TPP_SWITCH(float, tpp, hipLaunchKernelGGL((head_bwd_kernel<float, TPP, CO>), dim3(nb), dim3(256), 0, s,
                    (const float*)x, out, dout, pixels, hw, c_in, weight, sigmoid, (float*)dx, (float*)workspace));
  }
  int32_t rc = unet_check_launch("head_bwd_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(head_bwd_finalize_kernel, dim3(cdiv(c_out * (c_in + 1), 4)), dim3(256), 0, s,
                     (const float*)workspace, nb, c_out, c_in, dweight, dbias);
  return unet_check_launch("head_bwd_finalize_kernel");
}
