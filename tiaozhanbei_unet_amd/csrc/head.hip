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

// A wave works on 64 CONSECUTIVE pixels per iteration.  Phase 1: TPP lanes cooperate on a pixel (16 B of channels
// each, a pixel's NHWC row is one contiguous read / write), S = TPP sub-steps cover the 64 pixels and their loads
// are all in flight together.  Phase 2: lane = pixel, so the NCHW fp32 planes (out, dout) are read and written as
// 256 contiguous bytes per wave -- lanes trade values between the two layouts with wave shuffles.
// BN: x is the RAW convolution output y of the preceding conv-BatchNorm-ReLU layer; a = max(fma(y, scale, shift), 0),
// rounded to T like the stored activation would be, is formed on load (the activation tensor is never written).
template <typename T, int TPP, int CO, bool BN = false>
__global__ __launch_bounds__(256) void head_fwd_kernel(const T* __restrict__ x, long long pixels, long long hw,
                                                       int Cin, const float* __restrict__ w,
                                                       const float* __restrict__ b, int sigm,
                                                       float* __restrict__ out,
                                                       const float* __restrict__ bn_scale = nullptr,
                                                       const float* __restrict__ bn_shift = nullptr) {
  constexpr int PIECE = ET<T>::PIECE;
  constexpr int PPW = 64 / TPP, S = TPP;
  constexpr int SB = S < 8 ? S : 8;              // sub-steps whose loads are in flight together (register budget)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane % TPP, sub = lane / TPP;
  float wr[CO][PIECE];                  // this lane's slice of the CO filters, in registers
#pragma unroll
  for (int co = 0; co < CO; ++co)
#pragma unroll
    for (int j = 0; j < PIECE; ++j) wr[co][j] = w[co * Cin + g * PIECE + j];
  float sc[BN ? PIECE : 1], sh[BN ? PIECE : 1];
  if constexpr (BN) {
#pragma unroll
    for (int j = 0; j < PIECE; ++j) { sc[j] = bn_scale[g * PIECE + j]; sh[j] = bn_shift[g * PIECE + j]; }
  }
  const long long nw = (long long)gridDim.x * 4;
  for (long long c = blockIdx.x * 4LL + wave; c * 64 < pixels; c += nw) {
    const long long base = c * 64;
    float res[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) res[co] = 0.f;
#pragma unroll 1
    for (int s0 = 0; s0 < S; s0 += SB) {
      float v[SB][PIECE];
#pragma unroll
      for (int i = 0; i < SB; ++i) {
        const long long p = base + (s0 + i) * PPW + sub;
#pragma unroll
        for (int j = 0; j < PIECE; ++j) v[i][j] = 0.f;
        if (p < pixels) Vec<T>::load(x + p * Cin + g * PIECE, v[i]);
      }
      if constexpr (BN) {
#pragma unroll
        for (int i = 0; i < SB; ++i)
#pragma unroll
          for (int j = 0; j < PIECE; ++j)
            v[i][j] = ET<T>::to_f(ET<T>::from_f(fmaxf(fmaf(v[i][j], sc[j], sh[j]), 0.f)));
      }
      if constexpr (TPP == 8) {
        // 64 input channels in bf16 (the benchmark's heads): the 8 lanes of a pixel group hold 8 sub-steps x CO partial
        // dot products.  A reduce-scatter over the lane bits (7 exchanges per filter instead of 24 butterfly steps + 8
        // gathers) leaves lane g with the finished sum of sub-step g -- the same pairwise tree as the butterfly, bit for
        // bit -- and the wave's 64 results are its 64 consecutive pixels in a permuted lane order: the NCHW planes are
        // still written as 256 contiguous bytes.
        static_assert(SB == 8 && S == 8, "one batch");
        const bool b0 = g & 1, b1 = g & 2, b2 = g & 4;
#pragma unroll
        for (int co = 0; co < CO; ++co) {
          float a[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            a[i] = 0.f;
#pragma unroll
            for (int j = 0; j < PIECE; ++j) a[i] = fmaf(v[i][j], wr[co][j], a[i]);
          }
          float h4[4], h2[2];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float keep = b0 ? a[2 * k + 1] : a[2 * k], send = b0 ? a[2 * k] : a[2 * k + 1];
            h4[k] = keep + __shfl_xor(send, 1);
          }
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const float keep = b1 ? h4[2 * k + 1] : h4[2 * k], send = b1 ? h4[2 * k] : h4[2 * k + 1];
            h2[k] = keep + __shfl_xor(send, 2);
          }
          const float keep = b2 ? h2[1] : h2[0], send = b2 ? h2[0] : h2[1];
          res[co] = keep + __shfl_xor(send, 4);
        }
      } else {
#pragma unroll
      for (int i = 0; i < SB; ++i) {
        const int s = s0 + i;
        float acc[CO];
#pragma unroll
        for (int co = 0; co < CO; ++co) {
          acc[co] = 0.f;
#pragma unroll
          for (int j = 0; j < PIECE; ++j) acc[co] = fmaf(v[i][j], wr[co][j], acc[co]);
        }
#pragma unroll
        for (int m = 1; m < TPP; m <<= 1)
#pragma unroll
          for (int co = 0; co < CO; ++co) acc[co] += __shfl_xor(acc[co], m);
        // pixel (s, sub') of the chunk is lane s*PPW + sub' of phase 2: fetch it from lane group sub'
#pragma unroll
        for (int co = 0; co < CO; ++co) {
          const float t = __shfl(acc[co], (lane % PPW) * TPP);
          if (lane / PPW == s) res[co] = t;
        }
      }
      }
    }
    const long long p = base + (TPP == 8 ? g * PPW + sub : lane);
    if (p < pixels) {
      long long n, q;
      split_pixel(p, hw, n, q);
#pragma unroll
      for (int co = 0; co < CO; ++co) {
        float r = res[co] + b[co];
        if (sigm) r = 1.f / (1.f + expf(-r));
        out[(n * CO + co) * hw + q] = r;
      }
    }
  }
}

// BN: x is the raw convolution output y (see head_fwd_kernel): the activation is recomputed for the weight gradient,
// the gradient w.r.t. the activation gets the ReLU mask at once (dx = dz = da * [fma(y, scale, shift) > 0], rounded to
// T) and the two per-channel sums the BatchNorm backward needs -- sum dz and sum dz * (y - mean) of the ROUNDED dz --
// leave as one ordered partial per block (bn_part[block][2][Cin]): no separate reduction pass over (y, da).
template <typename T, int TPP, int CO, bool BN = false>
__global__ __launch_bounds__(256) void head_bwd_kernel(const T* __restrict__ x, const float* __restrict__ out,
                                                       const float* __restrict__ dout, long long pixels,
                                                       long long hw, int Cin, const float* __restrict__ w,
                                                       int sigm, T* __restrict__ dx, float* __restrict__ part,
                                                       const float* __restrict__ bn_scale = nullptr,
                                                       const float* __restrict__ bn_shift = nullptr,
                                                       const float* __restrict__ bn_mean = nullptr,
                                                       float* __restrict__ bn_part = nullptr) {
  constexpr int PIECE = ET<T>::PIECE;
  constexpr int PPW = 64 / TPP, S = TPP;
  __shared__ float red[4][MAXCO * 129];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane % TPP, sub = lane / TPP;
  float wr[CO][PIECE];
#pragma unroll
  for (int co = 0; co < CO; ++co)
#pragma unroll
    for (int j = 0; j < PIECE; ++j) wr[co][j] = w[co * Cin + g * PIECE + j];
  float dwacc[CO][PIECE], dbacc[CO];
#pragma unroll
  for (int co = 0; co < CO; ++co) {
    dbacc[co] = 0.f;
#pragma unroll
    for (int j = 0; j < PIECE; ++j) dwacc[co][j] = 0.f;
  }
  float sc[BN ? PIECE : 1], sh[BN ? PIECE : 1], mu[BN ? PIECE : 1], bs0[BN ? PIECE : 1], bs1[BN ? PIECE : 1];
  if constexpr (BN) {
#pragma unroll
    for (int j = 0; j < PIECE; ++j) {
      sc[j] = bn_scale[g * PIECE + j]; sh[j] = bn_shift[g * PIECE + j]; mu[j] = bn_mean[g * PIECE + j];
      bs0[j] = 0.f; bs1[j] = 0.f;
    }
  }
  constexpr int SB = S < 8 ? S : 8;              // sub-steps whose loads are in flight together (register budget)
  const long long nw = (long long)gridDim.x * 4;
  for (long long c = blockIdx.x * 4LL + wave; c * 64 < pixels; c += nw) {
    const long long base = c * 64;
    // phase 2 layout first: lane = pixel reads the NCHW planes coalesced
    float dl[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) dl[co] = 0.f;
    {
      const long long p = base + lane;
      if (p < pixels) {
        long long n, q;
        split_pixel(p, hw, n, q);
#pragma unroll
        for (int co = 0; co < CO; ++co) {
          float d = dout[(n * CO + co) * hw + q];
          if (sigm) { const float o = out[(n * CO + co) * hw + q]; d *= o * (1.f - o); }
          dl[co] = d;
          dbacc[co] += d;
        }
      }
    }
#pragma unroll 1
    for (int s0 = 0; s0 < S; s0 += SB) {
      float v[SB][PIECE];
#pragma unroll
      for (int i = 0; i < SB; ++i) {
        const long long p = base + (s0 + i) * PPW + sub;
#pragma unroll
        for (int j = 0; j < PIECE; ++j) v[i][j] = 0.f;
        if (p < pixels) Vec<T>::load(x + p * Cin + g * PIECE, v[i]);
      }
#pragma unroll
      for (int i = 0; i < SB; ++i) {
        const int s = s0 + i;
        const long long p = base + s * PPW + sub;
        float ds[CO], d[PIECE];
#pragma unroll
        for (int co = 0; co < CO; ++co) ds[co] = __shfl(dl[co], s * PPW + sub);
#pragma unroll
        for (int j = 0; j < PIECE; ++j) d[j] = 0.f;
        float av[PIECE];                       // the activation the 1x1 filter saw
        bool on[PIECE];
#pragma unroll
        for (int j = 0; j < PIECE; ++j) {
          if constexpr (BN) {
            const float z = fmaf(v[i][j], sc[j], sh[j]);
            on[j] = z > 0.f;
            av[j] = ET<T>::to_f(ET<T>::from_f(fmaxf(z, 0.f)));
          } else {
            on[j] = true;
            av[j] = v[i][j];
          }
        }
#pragma unroll
        for (int co = 0; co < CO; ++co)
#pragma unroll
          for (int j = 0; j < PIECE; ++j) {
            d[j] = fmaf(ds[co], wr[co][j], d[j]);
            dwacc[co][j] = fmaf(ds[co], av[j], dwacc[co][j]);        // ds is 0 past the end
          }
        if constexpr (BN) {
#pragma unroll
          for (int j = 0; j < PIECE; ++j) {
            d[j] = ET<T>::to_f(ET<T>::from_f(on[j] ? d[j] : 0.f));     // dz as stored
            bs0[j] += d[j];
            bs1[j] = fmaf(d[j], v[i][j] - mu[j], bs1[j]);
          }
        }
        if (p < pixels) Vec<T>::store(dx + p * Cin + g * PIECE, d);
      }
    }
  }
  // dW: lanes with equal g (stride TPP) hold partials of the same channels; db: every lane holds distinct pixels
#pragma unroll
  for (int m = TPP; m < 64; m <<= 1)
#pragma unroll
    for (int co = 0; co < CO; ++co)
#pragma unroll
      for (int j = 0; j < PIECE; ++j) dwacc[co][j] += __shfl_xor(dwacc[co][j], m);
#pragma unroll
  for (int m = 1; m < 64; m <<= 1)
#pragma unroll
    for (int co = 0; co < CO; ++co) dbacc[co] += __shfl_xor(dbacc[co], m);
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
  if constexpr (BN) {
#pragma unroll
    for (int m = TPP; m < 64; m <<= 1)
#pragma unroll
      for (int j = 0; j < PIECE; ++j) {
        bs0[j] += __shfl_xor(bs0[j], m);
        bs1[j] += __shfl_xor(bs1[j], m);
      }
    __syncthreads();                               // the weight-gradient partials above are consumed
    if (lane < TPP) {
#pragma unroll
      for (int j = 0; j < PIECE; ++j) {
        red[wave][lane * PIECE + j] = bs0[j];
        red[wave][Cin + lane * PIECE + j] = bs1[j];
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * Cin; i += 256)   // bn_part[block][2][Cin], fixed order
      bn_part[(size_t)blockIdx.x * 2 * Cin + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
  }
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

// ------------------------------------------------------------------------------------------------------
// head_bwd_tile_kernel: the backward of the 64-input-channel head, re-laid so that a lane's state is a handful of
// registers (the kernel above keeps 8 channels x (CO filters + CO weight-gradient sums + 5 BatchNorm values) per lane:
// 200-256 VGPRs, one wave per SIMD, 2.8 TB/s).  A wave takes a tile of 64 consecutive pixels x 64 channels:
//   1. the tile of x (or of the raw conv output y when BN) travels global -> LDS as eight fully coalesced 1-KiB
//      wave loads; lane = pixel reads the NCHW fp32 planes (out, dout) coalesced and leaves dlogit[co][pixel] in LDS;
//   2. lane = (channel pair, pixel parity) walks the tile's 32 pixels of its parity: reads its two channels of a
//      pixel (one 4-byte LDS read: conflict-free, a row is 128 B), the pixel's CO dlogits (LDS broadcast), forms the
//      activation, dx (with the ReLU mask when BN), the weight-gradient and BatchNorm-backward sums of ITS two
//      channels (2*CO + 4 accumulators) and writes dx back over the tile in place;
//   3. the tile goes LDS -> global with eight coalesced 16-byte stores per lane.
// ~50 VGPRs, 33 KB LDS per block: four blocks per CU keep the memory pipes full.  Same outputs and partial-sum
// formats as head_bwd_kernel (the finalize kernels are shared).
template <typename T, int CO, bool BN>
__global__ __launch_bounds__(256) void head_bwd_tile_kernel(const T* __restrict__ x, const float* __restrict__ out,
                                                            const float* __restrict__ dout, long long pixels,
                                                            long long hw, const float* __restrict__ w, int sigm,
                                                            T* __restrict__ dx, float* __restrict__ part,
                                                            const float* __restrict__ bn_scale,
                                                            const float* __restrict__ bn_shift,
                                                            const float* __restrict__ bn_mean,
                                                            float* __restrict__ bn_part) {
  constexpr int CIN = 64, ES = ET<T>::ES;
  constexpr int ROW = CIN * ES;                       // bytes per pixel row
  constexpr int TILE = 64 * ROW;                      // 8 KiB (bf16) / 16 KiB (fp32) per wave
  constexpr int NLD = TILE / 1024;                    // 16-byte pieces per lane
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  char* const tile = smem + wave * (TILE + CO * 256);
  float* const dlt = reinterpret_cast<float*>(tile + TILE);           // [CO][64]
  const int c2 = lane & 31, ph = lane >> 5;           // pixels of parity ph; this lane's two channels:
  // bf16: 2*c2, 2*c2+1 (one packed 4-byte LDS access); fp32: c2, c2+32 (two 4-byte accesses, each conflict-free --
  // an 8-byte vector access here is split and mis-merged across the unrolled iterations by hipcc 7.2's LDS load combiner)
  const int ch0 = sizeof(T) == 2 ? 2 * c2 : c2, ch1 = sizeof(T) == 2 ? 2 * c2 + 1 : c2 + 32;
  float wr[CO][2], dwacc[CO][2], dbacc[CO];
#pragma unroll
  for (int co = 0; co < CO; ++co) {
    wr[co][0] = w[co * CIN + ch0]; wr[co][1] = w[co * CIN + ch1];
    dwacc[co][0] = dwacc[co][1] = 0.f; dbacc[co] = 0.f;
  }
  float sc[2] = {1.f, 1.f}, sh[2] = {0.f, 0.f}, mu[2] = {0.f, 0.f}, bs0[2] = {0.f, 0.f}, bs1[2] = {0.f, 0.f};
  if constexpr (BN) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ch = j ? ch1 : ch0;
      sc[j] = bn_scale[ch]; sh[j] = bn_shift[ch]; mu[j] = bn_mean[ch];
    }
  }
  const long long nw = (long long)gridDim.x * 4;
  for (long long c = blockIdx.x * 4LL + wave; c * 64 < pixels; c += nw) {
    const long long base = c * 64;
    // ---- 1. tile -> LDS (contiguous in NHWC), dlogits -> LDS
    u32x4 ld[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int q = i * 64 + lane;                    // 16-byte piece index inside the tile
      const long long p = base + q / (ROW / 16);
      ld[i] = u32x4{0u, 0u, 0u, 0u};
      if (p < pixels) ld[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(x) + base * ROW + (size_t)q * 16);
    }
    {
      const long long p = base + lane;
      float dl[CO];
#pragma unroll
      for (int co = 0; co < CO; ++co) dl[co] = 0.f;
      if (p < pixels) {
        long long n, q;
        split_pixel(p, hw, n, q);
#pragma unroll
        for (int co = 0; co < CO; ++co) {
          float d = dout[(n * CO + co) * hw + q];
          if (sigm) { const float o = out[(n * CO + co) * hw + q]; d *= o * (1.f - o); }
          dl[co] = d;
          dbacc[co] += d;
        }
      }
#pragma unroll
      for (int co = 0; co < CO; ++co) dlt[co * 64 + lane] = dl[co];
    }
#pragma unroll
    for (int i = 0; i < NLD; ++i) *reinterpret_cast<u32x4*>(tile + (i * 64 + lane) * 16) = ld[i];
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the wave's own LDS writes are done (one wave owns a tile)
    // ---- 2. lane = (channel pair, pixel parity)
#pragma unroll 4
    for (int k = 0; k < 32; ++k) {
      const int p = 2 * k + ph;
      float v[2];
      if constexpr (sizeof(T) == 2) {
        const unsigned u = *reinterpret_cast<const unsigned*>(tile + p * ROW + c2 * 4);
        v[0] = __builtin_bit_cast(float, u << 16);
        v[1] = __builtin_bit_cast(float, u & 0xFFFF0000u);
      } else {
        v[0] = *reinterpret_cast<const float*>(tile + p * ROW + ch0 * 4);
        v[1] = *reinterpret_cast<const float*>(tile + p * ROW + ch1 * 4);
      }
      float ds[CO];
#pragma unroll
      for (int co = 0; co < CO; ++co) ds[co] = dlt[co * 64 + p];
      float d[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float av = v[j];
        bool on = true;
        if constexpr (BN) {
          const float z = fmaf(v[j], sc[j], sh[j]);
          on = z > 0.f;
          av = ET<T>::to_f(ET<T>::from_f(fmaxf(z, 0.f)));       // the activation the 1x1 filter saw
        }
        float t = 0.f;
#pragma unroll
        for (int co = 0; co < CO; ++co) {
          t = fmaf(ds[co], wr[co][j], t);
          dwacc[co][j] = fmaf(ds[co], av, dwacc[co][j]);        // a pixel past the end has ds = 0
        }
        if constexpr (BN) {
          t = ET<T>::to_f(ET<T>::from_f(on ? t : 0.f));         // dz as stored
          bs0[j] += t;
          bs1[j] = fmaf(t, v[j] - mu[j], bs1[j]);
        }
        d[j] = t;
      }
      if constexpr (sizeof(T) == 2) {
        const bf16_t d0 = (bf16_t)d[0], d1 = (bf16_t)d[1];
        const unsigned u = (unsigned)__builtin_bit_cast(unsigned short, d0) | ((unsigned)__builtin_bit_cast(unsigned short, d1) << 16);
        *reinterpret_cast<unsigned*>(tile + p * ROW + c2 * 4) = u;
      } else {
        *reinterpret_cast<float*>(tile + p * ROW + ch0 * 4) = d[0];
        *reinterpret_cast<float*>(tile + p * ROW + ch1 * 4) = d[1];
      }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // ---- 3. tile -> global
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int q = i * 64 + lane;
      const long long p = base + q / (ROW / 16);
      const u32x4 r = *reinterpret_cast<const u32x4*>(tile + q * 16);
      if (p < pixels) *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(dx) + base * ROW + (size_t)q * 16) = r;
    }
    __builtin_amdgcn_wave_barrier();                 // (the next tile's LDS writes must not pass these reads)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  // ---- block partials.  dW / BatchNorm sums: the two parity halves hold the same channels; db: lane = pixel partials
#pragma unroll
  for (int co = 0; co < CO; ++co) {
#pragma unroll
    for (int j = 0; j < 2; ++j) dwacc[co][j] += __shfl_xor(dwacc[co][j], 32);
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) dbacc[co] += __shfl_xor(dbacc[co], m);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) { bs0[j] += __shfl_xor(bs0[j], 32); bs1[j] += __shfl_xor(bs1[j], 32); }
  __syncthreads();                                   // every wave is done with its tile area
  float* red = reinterpret_cast<float*>(smem);       // [4 waves][CO * 65 + 128]
  constexpr int stride = CIN + 1, RW = CO * stride + 2 * CIN;
  if (ph == 0) {
#pragma unroll
    for (int co = 0; co < CO; ++co) {
      red[wave * RW + co * stride + ch0] = dwacc[co][0];
      red[wave * RW + co * stride + ch1] = dwacc[co][1];
      if (lane == 0) red[wave * RW + co * stride + CIN] = dbacc[co];
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      red[wave * RW + CO * stride + (j ? ch1 : ch0)] = bs0[j];
      red[wave * RW + CO * stride + CIN + (j ? ch1 : ch0)] = bs1[j];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < CO * stride; i += 256)
    part[(size_t)blockIdx.x * CO * stride + i] = (red[i] + red[RW + i]) + (red[2 * RW + i] + red[3 * RW + i]);
  if constexpr (BN) {
    for (int i = threadIdx.x; i < 2 * CIN; i += 256) {
      const int o = CO * stride + i;
      bn_part[(size_t)blockIdx.x * 2 * CIN + i] = (red[o] + red[RW + o]) + (red[2 * RW + o] + red[3 * RW + o]);
    }
  }
}

template <typename T, int CO, bool BN>
int32_t launch_head_bwd_tile(const void* x, const float* out, const float* dout, long long pixels, long long hw,
                             const float* w, int sigm, void* dx, float* part, const float* bn_scale,
                             const float* bn_shift, const float* bn_mean, float* bn_part, int nb, hipStream_t s) {
  constexpr int TILE = 64 * 64 * ET<T>::ES;
  constexpr int LDS = 4 * (TILE + CO * 256);
  static_assert(LDS >= 4 * (CO * 65 + 128) * 4, "reduction scratch fits the tile area");
  auto kern = head_bwd_tile_kernel<T, CO, BN>;
  unet_set_max_lds(reinterpret_cast<const void*>(kern), LDS);
  hipLaunchKernelGGL(kern, dim3(nb), dim3(256), LDS, s, (const T*)x, out, dout, pixels, hw, w, sigm, (T*)dx, part,
                     bn_scale, bn_shift, bn_mean, bn_part);
  return unet_check_launch("head_bwd_tile_kernel");
}

inline int head_blocks(long long pixels, int tpp) {
  (void)tpp;
  long long b = cdiv64(pixels, 256 * 4);         // a wave takes 64 pixels per iteration, ~4 iterations per wave
  if (b > 1024) b = 1024;                        // (every block leaves a partial for the weight-gradient finalize)
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
  if (c_in == 64 && (dtype == UNET_BF16 || dtype == UNET_F32)) {
    nb = head_blocks(pixels, 8);
    int32_t rc = UNET_OK;
    if (dtype == UNET_BF16) {
      HEAD_CO_SWITCH(rc = (launch_head_bwd_tile<bf16_t, CO, false>(x, out, dout, pixels, hw, weight, sigmoid, dx, (float*)workspace,
                                                                 nullptr, nullptr, nullptr, nullptr, nb, s)));
    } else {
      HEAD_CO_SWITCH(rc = (launch_head_bwd_tile<float, CO, false>(x, out, dout, pixels, hw, weight, sigmoid, dx, (float*)workspace,
                                                                nullptr, nullptr, nullptr, nullptr, nb, s)));
    }
    if (rc) return rc;
  } else if (dtype == UNET_BF16) {
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

// ---- OutConv fed by the RAW convolution output of the last conv-BatchNorm-ReLU layer (src/model.py:17-19 -> :72) ----
extern "C" int32_t unet_head_bnrelu_fwd(int32_t dtype, const void* y, int32_t n, int32_t h, int32_t w, int32_t c_in,
                                        const float* bn_scale, const float* bn_shift, const float* weight,
                                        const float* bias, int32_t c_out, int32_t sigmoid, float* out, void* stream) {
  UNET_REQUIRE(y && bn_scale && bn_shift && weight && bias && out, UNET_ERR_BAD_ARG, "unet_head_bnrelu_fwd: null pointer");
  UNET_REQUIRE(n > 0 && h > 0 && w > 0, UNET_ERR_BAD_ARG, "unet_head_bnrelu_fwd: bad dims");
  UNET_REQUIRE(c_out >= 1 && c_out <= MAXCO && (c_in == 32 || c_in == 64 || c_in == 128), UNET_ERR_UNSUPPORTED,
               "unet_head_bnrelu_fwd: %d -> %d channels unsupported (c_out <= 8, c_in in {32,64,128})", c_in, c_out);
  UNET_REQUIRE(c_out <= c_in / (dtype == UNET_BF16 ? 8 : 4), UNET_ERR_UNSUPPORTED,
               "unet_head_bnrelu_fwd: c_out %d too wide for c_in %d", c_out, c_in);
  hipStream_t s = (hipStream_t)stream;
  const long long pixels = (long long)n * h * w, hw = (long long)h * w;
  ProfScope prof(UNET_K_HEAD, 2.0 * pixels * c_in * c_out, s);
  if (dtype == UNET_BF16) {
    const int tpp = tpp_of<bf16_t>(c_in);
    HEAD_TPP_SWITCH(bf16_t, tpp, hipLaunchKernelGGL((head_fwd_kernel<bf16_t, TPP, CO, true>), dim3(head_blocks(pixels, TPP)),
                    dim3(256), 0, s, (const bf16_t*)y, pixels, hw, c_in, weight, bias, sigmoid, out, bn_scale, bn_shift));
  } else if (dtype == UNET_F32) {
    const int tpp = tpp_of<float>(c_in);
    HEAD_TPP_SWITCH(float, tpp, hipLaunchKernelGGL((head_fwd_kernel<float, TPP, CO, true>), dim3(head_blocks(pixels, TPP)),
                    dim3(256), 0, s, (const float*)y, pixels, hw, c_in, weight, bias, sigmoid, out, bn_scale, bn_shift));
  } else {
    unet_set_error("unet_head_bnrelu_fwd: dtype %d", dtype);
    return UNET_ERR_BAD_ARG;
  }
  return unet_check_launch("head_fwd_kernel<BN>");
}

extern "C" size_t unet_head_bnrelu_max_parts(void) { return 1024; }

extern "C" int32_t unet_head_bnrelu_bwd(int32_t dtype, const void* y, const float* bn_scale, const float* bn_shift,
                                        const float* bn_mean, const float* out, const float* dout, int32_t n, int32_t h,
                                        int32_t w, int32_t c_in, const float* weight, int32_t c_out, int32_t sigmoid,
                                        void* dz, float* dweight, float* dbias, float* bn_partial, int32_t* n_parts,
                                        void* workspace, size_t workspace_bytes, void* stream) {
  UNET_REQUIRE(y && bn_scale && bn_shift && bn_mean && dout && weight && dz && dweight && dbias && bn_partial && n_parts &&
               workspace && (out || !sigmoid), UNET_ERR_BAD_ARG, "unet_head_bnrelu_bwd: null pointer");
  UNET_REQUIRE(n > 0 && h > 0 && w > 0, UNET_ERR_BAD_ARG, "unet_head_bnrelu_bwd: bad dims");
  UNET_REQUIRE(c_out >= 1 && c_out <= MAXCO && (c_in == 32 || c_in == 64 || c_in == 128), UNET_ERR_UNSUPPORTED,
               "unet_head_bnrelu_bwd: %d -> %d channels unsupported", c_in, c_out);
  UNET_REQUIRE(workspace_bytes >= unet_head_bwd_workspace(n, h, w, c_in, c_out), UNET_ERR_WORKSPACE,
               "unet_head_bnrelu_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const long long pixels = (long long)n * h * w, hw = (long long)h * w;
  ProfScope prof(UNET_K_HEAD, 4.0 * pixels * c_in * c_out, s);
  int nb = 0;
  if (c_in == 64 && (dtype == UNET_BF16 || dtype == UNET_F32)) {
    nb = head_blocks(pixels, 8);
    int32_t rc = UNET_OK;
    if (dtype == UNET_BF16) {
      HEAD_CO_SWITCH(rc = (launch_head_bwd_tile<bf16_t, CO, true>(y, out, dout, pixels, hw, weight, sigmoid, dz, (float*)workspace,
                                                                bn_scale, bn_shift, bn_mean, bn_partial, nb, s)));
    } else {
      HEAD_CO_SWITCH(rc = (launch_head_bwd_tile<float, CO, true>(y, out, dout, pixels, hw, weight, sigmoid, dz, (float*)workspace,
                                                               bn_scale, bn_shift, bn_mean, bn_partial, nb, s)));
    }
    if (rc) return rc;
  } else if (dtype == UNET_BF16) {
    const int tpp = tpp_of<bf16_t>(c_in);
    nb = head_blocks(pixels, tpp);
    HEAD_TPP_SWITCH(bf16_t, tpp, hipLaunchKernelGGL((head_bwd_kernel<bf16_t, TPP, CO, true>), dim3(nb), dim3(256), 0, s,
                    (const bf16_t*)y, out, dout, pixels, hw, c_in, weight, sigmoid, (bf16_t*)dz, (float*)workspace,
                    bn_scale, bn_shift, bn_mean, bn_partial));
  } else if (dtype == UNET_F32) {
    const int tpp = tpp_of<float>(c_in);
    nb = head_blocks(pixels, tpp);
    HEAD_TPP_SWITCH(float, tpp, hipLaunchKernelGGL((head_bwd_kernel<float, TPP, CO, true>), dim3(nb), dim3(256), 0, s,
                    (const float*)y, out, dout, pixels, hw, c_in, weight, sigmoid, (float*)dz, (float*)workspace,
                    bn_scale, bn_shift, bn_mean, bn_partial));
  } else {
    unet_set_error("unet_head_bnrelu_bwd: dtype %d", dtype);
    return UNET_ERR_BAD_ARG;
  }
  int32_t rc = unet_check_launch("head_bwd_kernel<BN>");
  if (rc) return rc;
  *n_parts = nb;
  hipLaunchKernelGGL(head_bwd_finalize_kernel, dim3(cdiv(c_out * (c_in + 1), 4)), dim3(256), 0, s,
                     (const float*)workspace, nb, c_out, c_in, dweight, dbias);
  return unet_check_launch("head_bwd_finalize_kernel");
}
