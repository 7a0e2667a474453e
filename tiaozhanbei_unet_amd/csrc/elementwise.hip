// HBM-bound streaming kernels of the hot path (gfx950): layout changes, weight packing, max-pool,
// bilinear x2, fused Adam.  16-byte vector accesses along the channel (fastest) axis, grid-stride loops.
#include "common.h"

namespace {

inline int ew_blocks(long long items) { return (int)std::min<long long>(cdiv64(items, 256), 256 * 16); }

// ------------------------------------------------------------------------------- layout
// One thread per pixel: the C (<= 8 typical) plane reads are coalesced across threads, each thread then
// writes its whole padded channel row with 16-byte stores (zeros beyond C).
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst,
                                                           int N, int C, int H, int W, int Cp) {
  constexpr int PIECE = ET<T>::PIECE;
  const long long hw = (long long)H * W;
  const long long total = (long long)N * hw;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long n = i / hw, p = i - n * hw;
    const float* s = src + n * C * hw + p;
    T* o = dst + i * Cp;
    for (int c0 = 0; c0 < Cp; c0 += PIECE) {
      float v[PIECE];
#pragma unroll
      for (int j = 0; j < PIECE; ++j) v[j] = (c0 + j < C) ? s[(long long)(c0 + j) * hw] : 0.f;
      Vec<T>::store(o + c0, v);
    }
  }
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ src, float* __restrict__ dst, int N, int C, int H,
                                    int W, int Cp) {
  const long long hw = (long long)H * W;
  const long long total = (long long)N * C * hw;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long p = i % hw;
    const long long t = i / hw;
    const int c = (int)(t % C);
    const long long n = t / C;
    dst[i] = ET<T>::to_f(src[(n * hw + p) * Cp + c]);
  }
}

// ------------------------------------------------------------------------------- weight packing
// out index -> source parameter index; zero where the padded GEMM dims exceed the parameter's.
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ w, T* __restrict__ out, int Co, int Ci, int rows,
                                   int K, int mode, long long total, const float* __restrict__ scale) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    float v = 0.f;
    if (mode == UNET_PACK_CONV_FWD) {            // out[tap][co][ci]  <- w[co][ci][tap]
      const int ci = (int)(i % K);
      long long t = i / K;
      const int co = (int)(t % rows);
      const int tap = (int)(t / rows);
      if (co < Co && ci < Ci) v = w[((long long)co * Ci + ci) * 9 + tap] * (scale ? scale[co] : 1.f);   // BN fold
    } else if (mode == UNET_PACK_CONV_DGRAD) {   // out[tap'][ci][co] <- w[co][ci][8 - tap']
      const int co = (int)(i % K);
      long long t = i / K;
      const int ci = (int)(t % rows);
      const int tap = (int)(t / rows);
      if (co < Co && ci < Ci) v = w[((long long)co * Ci + ci) * 9 + (8 - tap)];
    } else if (mode == UNET_PACK_CONVT_FWD) {    // out[z][co][ci]    <- w[ci][co][z]
      const int ci = (int)(i % K);
      long long t = i / K;
      const int co = (int)(t % rows);
      const int z = (int)(t / rows);
      if (co < Co && ci < Ci) v = w[((long long)ci * Co + co) * 4 + z];
    } else {                                     // out[ci][z][co]    <- w[ci][co][z]      (row length 4*K)
      const int co = (int)(i % K);
      long long t = i / K;
      const int z = (int)(t % 4);
      const int ci = (int)(t / 4);
      if (co < Co && ci < Ci) v = w[((long long)ci * Co + co) * 4 + z];
    }
    out[i] = ET<T>::from_f(v);
  }
}

// every weight of the model in one launch: blockIdx.y = descriptor.  The parameter is w[a][b][taps]
// (conv: a=co, b=ci, 9 taps; convT: a=ci, b=co, 4 taps).  A block moves one tile through LDS so that BOTH
// the fp32 reads (runs of TB*taps floats) and the packed writes (runs of 64 elements along the GEMM K axis)
// are coalesced: tile 8(a) x 64(b) when the output is contiguous along b, 64(a) x 8(b) when along a.
template <typename T>
__global__ __launch_bounds__(256) void pack_weights_batched_kernel(const unet_pack_desc* __restrict__ descs) {
  __shared__ float tile[32 * 289];                 // (>= 64 * 8 * 9 + 64 of the single-layout path)
  const unet_pack_desc d = descs[blockIdx.y];
  const int mode = d.mode;
  // PAIRED fast path (bf16, 3x3 conv, unpadded dims that are multiples of 32): the forward layout [tap][co][ci] and the
  // data-gradient layout [8 - tap][ci][co] of one weight are consecutive descriptors; the forward descriptor's blocks
  // write BOTH from one 32 (co) x 32 (ci) x 9 tile -- the fp32 parameter is read once instead of twice (346 -> 173 MB of
  // reads per step); the data-gradient descriptor's blocks have nothing left to do.
  if constexpr (sizeof(T) == 2) {
    auto pair_ok = [](const unet_pack_desc& f, const unet_pack_desc& g) {
      return f.mode == UNET_PACK_CONV_FWD && g.mode == UNET_PACK_CONV_DGRAD && f.w == g.w && f.c_out == g.c_out &&
             f.c_in == g.c_in && f.rows == f.c_out && f.k == f.c_in && g.rows == g.c_in && g.k == g.c_out &&
             (f.c_out & 31) == 0 && (f.c_in & 31) == 0;
    };
    if (mode == UNET_PACK_CONV_DGRAD && blockIdx.y > 0 && pair_ok(descs[blockIdx.y - 1], d)) return;
    if (mode == UNET_PACK_CONV_FWD && blockIdx.y + 1 < gridDim.y && pair_ok(d, descs[blockIdx.y + 1])) {
      const unet_pack_desc g = descs[blockIdx.y + 1];
      const int Co = d.c_out, Ci = d.c_in;
      const int nB = Ci / 32, ntile = (Co / 32) * nB;
      const float* __restrict__ w = d.w;
      bf16_t* __restrict__ of = reinterpret_cast<bf16_t*>(d.out);
      bf16_t* __restrict__ og = reinterpret_cast<bf16_t*>(g.out);
      for (int t = blockIdx.x; t < ntile; t += gridDim.x) {
        const int a0 = (t / nB) * 32, b0 = (t % nB) * 32;
        __syncthreads();
        for (int i = threadIdx.x; i < 32 * 288; i += 256) {      // 32 rows of 288 contiguous floats (32 ci x 9 taps)
          const int al = i / 288, r = i - al * 288;
          tile[al * 289 + r] = w[((size_t)(a0 + al) * Ci + b0) * 9 + r];
        }
        __syncthreads();
        for (int gi = threadIdx.x; gi < 2 * 32 * 9 * 4; gi += 256) {   // 16-byte stores: 8 consecutive ci (fwd) / co (dgrad)
          const int lay = gi / (32 * 9 * 4), rem = gi % (32 * 9 * 4);
          const int q = rem & 3, rest = rem >> 2;
          const int tap = rest % 9, outer = rest / 9;             // outer: co (fwd) / ci (dgrad)
          bf16x8 r8;
          if (lay == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) r8[e] = (bf16_t)tile[outer * 289 + (q * 8 + e) * 9 + tap];
            *reinterpret_cast<bf16x8*>(of + ((size_t)tap * Co + a0 + outer) * Ci + b0 + q * 8) = r8;
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) r8[e] = (bf16_t)tile[(q * 8 + e) * 289 + outer * 9 + tap];
            *reinterpret_cast<bf16x8*>(og + ((size_t)(8 - tap) * Ci + b0 + outer) * Co + a0 + q * 8) = r8;
          }
        }
      }
      return;
    }
  }
  const bool conv = mode <= UNET_PACK_CONV_DGRAD;
  const int taps = conv ? 9 : 4;
  const int As = conv ? d.c_out : d.c_in, Bs = conv ? d.c_in : d.c_out;     // source dims
  const bool along_b = (mode == UNET_PACK_CONV_FWD || mode == UNET_PACK_CONVT_DGRAD);
  // padded extents of a and b in the OUTPUT
  const int Ap = (mode == UNET_PACK_CONV_FWD || mode == UNET_PACK_CONVT_DGRAD) ? d.rows : d.k;
  const int Bp = (mode == UNET_PACK_CONV_FWD || mode == UNET_PACK_CONVT_DGRAD) ? d.k : d.rows;
  const int TA = along_b ? 8 : 64, TB = along_b ? 64 : 8;
  const int nA = (Ap + TA - 1) / TA, nB = (Bp + TB - 1) / TB;
  const float* __restrict__ w = d.w;
  T* __restrict__ out = reinterpret_cast<T*>(d.out);
  const int run = TB * taps;                       // contiguous source floats per a
  for (int t = blockIdx.x; t < nA * nB; t += gridDim.x) {
    const int a0 = (t / nB) * TA, b0 = (t % nB) * TB;
    __syncthreads();
    for (int i = threadIdx.x; i < TA * run; i += 256) {
      const int al = i / run, r = i - al * run;
      const int a = a0 + al, bb = b0 + r / taps;
      tile[i] = (a < As && bb < Bs) ? w[((size_t)a * Bs + b0) * taps + r] : 0.f;
    }
    __syncthreads();
    // TA*TB*taps outputs; innermost = the 64-wide contiguous axis.  Eight consecutive outputs per thread -> one
    // 16-byte store (bf16) when the contiguous extent is a multiple of 8 (it is: GEMM dims are padded to 64)
    if ((((along_b ? Bp : Ap) & 7) == 0) && ((d.k & 7) == 0)) {
      for (int gi = threadIdx.x; gi < TA * TB * taps / 8; gi += 256) {
        const int inner0 = (gi & 7) * 8, rest = gi >> 3;
        const int tap = rest % taps, outer = rest / taps;
        const int al0 = along_b ? outer : inner0, bl0 = along_b ? inner0 : outer;
        const int a = a0 + al0, bb = b0 + bl0;
        if (a >= Ap || bb >= Bp) continue;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = tile[(al0 + (along_b ? 0 : e)) * run + (bl0 + (along_b ? e : 0)) * taps + tap];
        size_t o;
        if (mode == UNET_PACK_CONV_FWD) o = ((size_t)tap * d.rows + a) * d.k + bb;
        else if (mode == UNET_PACK_CONV_DGRAD) o = ((size_t)(8 - tap) * d.rows + bb) * d.k + a;
        else if (mode == UNET_PACK_CONVT_FWD) o = ((size_t)tap * d.rows + bb) * d.k + a;
        else o = ((size_t)a * 4 + tap) * d.k + bb;
        if constexpr (sizeof(T) == 2) {
          bf16x8 r;
#pragma unroll
          for (int e = 0; e < 8; ++e) r[e] = (bf16_t)v[e];
          *reinterpret_cast<bf16x8*>(out + o) = r;
        } else {
          *reinterpret_cast<f32x4*>(out + o) = f32x4{v[0], v[1], v[2], v[3]};
          *reinterpret_cast<f32x4*>(out + o + 4) = f32x4{v[4], v[5], v[6], v[7]};
        }
      }
      continue;
    }
    for (int i = threadIdx.x; i < TA * TB * taps; i += 256) {
      const int inner = i & 63, rest = i >> 6;
      const int tap = rest % taps, outer = rest / taps;          // outer: the 8-wide axis
      const int al = along_b ? outer : inner, bl = along_b ? inner : outer;
      const int a = a0 + al, bb = b0 + bl;
      if (a >= Ap || bb >= Bp) continue;
      const float v = tile[al * run + bl * taps + tap];
      size_t o;
      if (mode == UNET_PACK_CONV_FWD) o = ((size_t)tap * d.rows + a) * d.k + bb;
      else if (mode == UNET_PACK_CONV_DGRAD) o = ((size_t)(8 - tap) * d.rows + bb) * d.k + a;
      else if (mode == UNET_PACK_CONVT_FWD) o = ((size_t)tap * d.rows + bb) * d.k + a;
      else o = ((size_t)a * 4 + tap) * d.k + bb;
      out[o] = ET<T>::from_f(v);
    }
  }
}

// ------------------------------------------------------------------------------- max pool 2x2
template <typename T>
__global__ void maxpool2_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C) {
  constexpr int PIECE = ET<T>::PIECE;
  const int OH = H / 2, OW = W / 2, G = C / PIECE;
  const long long total = (long long)N * OH * OW * G;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(i % G);
    long long t = i / G;
    const int ox = (int)(t % OW);  t /= OW;
    const int oy = (int)(t % OH);
    const long long n = t / OH;
    const T* p = x + ((n * H + 2 * oy) * W + 2 * ox) * (long long)C + g * PIECE;
    float a[PIECE], b[PIECE], c[PIECE], d[PIECE];
    Vec<T>::load(p, a);
    Vec<T>::load(p + C, b);
    Vec<T>::load(p + (long long)W * C, c);
    Vec<T>::load(p + (long long)W * C + C, d);
#pragma unroll
    for (int j = 0; j < PIECE; ++j) a[j] = fmaxf(fmaxf(a[j], b[j]), fmaxf(c[j], d[j]));
    Vec<T>::store(y + i * PIECE, a);
  }
}

// gradient goes to the FIRST maximum in row-major window order; rows/cols dropped by floor get 0.
template <typename T>
__global__ void maxpool2_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx,
                                    int N, int H, int W, int C, int accumulate) {
  constexpr int PIECE = ET<T>::PIECE;
  const int OH = H / 2, OW = W / 2, G = C / PIECE;
  const int WH = (H + 1) / 2, WW = (W + 1) / 2;   // windows incl. the ragged edge
  const long long total = (long long)N * WH * WW * G;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(i % G);
    long long t = i / G;
    const int ox = (int)(t % WW);  t /= WW;
    const int oy = (int)(t % WH);
    const long long n = t / WH;
    const long long base = ((n * H + 2 * oy) * W + 2 * ox) * (long long)C + g * PIECE;
    float z[PIECE];
#pragma unroll
    for (int j = 0; j < PIECE; ++j) z[j] = 0.f;
    if (oy < OH && ox < OW) {
      float a[PIECE], b[PIECE], c[PIECE], d[PIECE], gr[PIECE];
      Vec<T>::load(x + base, a);
      Vec<T>::load(x + base + C, b);
      Vec<T>::load(x + base + (long long)W * C, c);
      Vec<T>::load(x + base + (long long)W * C + C, d);
      Vec<T>::load(dy + (((n * OH + oy) * OW + ox) * (long long)C + g * PIECE), gr);
      float ga[PIECE], gb[PIECE], gc[PIECE], gd[PIECE];
#pragma unroll
      for (int j = 0; j < PIECE; ++j) {
        const float m = fmaxf(fmaxf(a[j], b[j]), fmaxf(c[j], d[j]));
        const bool fa = a[j] == m, fb = !fa && b[j] == m, fc = !fa && !fb && c[j] == m;
        const bool fd = !fa && !fb && !fc;
        ga[j] = fa ? gr[j] : 0.f; gb[j] = fb ? gr[j] : 0.f; gc[j] = fc ? gr[j] : 0.f; gd[j] = fd ? gr[j] : 0.f;
      }
      if (accumulate) {                     // dx += routed gradient (fan-in of a skip connection)
        Vec<T>::load(dx + base, a);
        Vec<T>::load(dx + base + C, b);
        Vec<T>::load(dx + base + (long long)W * C, c);
        Vec<T>::load(dx + base + (long long)W * C + C, d);
#pragma unroll
        for (int j = 0; j < PIECE; ++j) { ga[j] += a[j]; gb[j] += b[j]; gc[j] += c[j]; gd[j] += d[j]; }
      }
      Vec<T>::store(dx + base, ga);
      Vec<T>::store(dx + base + C, gb);
      Vec<T>::store(dx + base + (long long)W * C, gc);
      Vec<T>::store(dx + base + (long long)W * C + C, gd);
    } else if (!accumulate) {
      // ragged edge: whatever exists of this window was not pooled
      const int y0 = 2 * oy, x0 = 2 * ox;
      for (int dyy = 0; dyy < 2; ++dyy)
        for (int dxx = 0; dxx < 2; ++dxx)
          if (y0 + dyy < H && x0 + dxx < W)
            Vec<T>::store(dx + base + ((long long)dyy * W + dxx) * C, z);
    }
  }
}

// ------------------------------------------------------------------------------- BatchNorm-apply + ReLU + max pool 2x2, fused
// Encoder levels (src/model.py:18-19 followed by the next level's nn.MaxPool2d(2), :32): ONE pass reads the raw conv
// output y, writes the activation a (the skip tensor) and its 2x2 max (the next level's input) -- the pool's own read
// of a is gone.  A thread owns one window x 16 bytes of channels; (256 * PIECE) % C == 0, so its channel group -- and its
// coefficients, in registers -- never change along the grid-stride loop.
template <typename T, typename IDX>
__global__ __launch_bounds__(256) void bn_relu_pool_fwd_kernel(const T* __restrict__ y, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, T* __restrict__ a,
                                                               T* __restrict__ pooled, int N, int H, int W, int C, int nt) {
  constexpr int PIECE = ET<T>::PIECE;
  typedef IDX idx_t;                               // int when every element index fits 31 bits (no 64-bit divisions)
  const int OH = H / 2, OW = W / 2, G = C / PIECE;
  const int WH = (H + 1) / 2, WW = (W + 1) / 2;   // windows incl. the ragged edge (its pixels get a, no pooled value)
  const idx_t total = (idx_t)N * WH * WW * G;
  const int g = threadIdx.x % G;
  float sc[PIECE], sh[PIECE];
#pragma unroll
  for (int j = 0; j < PIECE; ++j) { sc[j] = scale[g * PIECE + j]; sh[j] = shift[g * PIECE + j]; }
  for (idx_t i = (idx_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (idx_t)gridDim.x * 256) {
    idx_t t = i / G;
    const int ox = (int)(t % WW);  t /= WW;
    const int oy = (int)(t % WH);
    const idx_t n = t / WH;
    const idx_t base = ((n * H + 2 * oy) * W + 2 * ox) * (idx_t)C + g * PIECE;
    const bool ex = 2 * ox + 1 < W, ey = 2 * oy + 1 < H;
    float v[4][PIECE];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool ok = ((k & 1) == 0 || ex) && ((k >> 1) == 0 || ey);
      const idx_t o = base + ((idx_t)(k >> 1) * W + (k & 1)) * C;
#pragma unroll
      for (int j = 0; j < PIECE; ++j) v[k][j] = 0.f;
      if (ok) {
        if (nt) Vec<T>::load_nt(y + o, v[k]);
        else Vec<T>::load(y + o, v[k]);
#pragma unroll
        for (int j = 0; j < PIECE; ++j) v[k][j] = ET<T>::to_f(ET<T>::from_f(fmaxf(fmaf(v[k][j], sc[j], sh[j]), 0.f)));
        Vec<T>::store(a + o, v[k]);
      }
    }
    if (oy < OH && ox < OW) {
#pragma unroll
      for (int j = 0; j < PIECE; ++j) v[0][j] = fmaxf(fmaxf(v[0][j], v[1][j]), fmaxf(v[2][j], v[3][j]));
      Vec<T>::store(pooled + (((n * OH + oy) * OW + ox) * (idx_t)C + g * PIECE), v[0]);
    }
  }
}

// Backward of the same pair for a skip tensor whose OTHER consumers have already added their gradients into da_old
// (ops.GradSink; may be NULL): da = da_old + route(dpooled) (first maximum wins ties, as torch), then the ReLU mask of
// this layer and the two BatchNorm-backward sums, in one pass: dz = da * [z > 0] (rounded to T exactly like the
// unfused kernels round da), part[block][2][C] = (sum dz, sum dz * (y - mean)) -> unet_bn_bwd_premasked.  The window's
// activations are recomputed from y (rounded like the stored ones), so a itself is not read.
template <typename T, typename IDX>
__global__ __launch_bounds__(256) void bn_relu_pool_bwd_kernel(const T* __restrict__ y, const T* __restrict__ dpooled,
                                                               const T* da_old, const float* __restrict__ scale,
                                                               const float* __restrict__ shift,
                                                               const float* __restrict__ mean, T* dz,
                                                               float* __restrict__ part, int N, int H, int W, int C, int nt) {
  constexpr int PIECE = ET<T>::PIECE;
  typedef IDX idx_t;
  __shared__ float red[2][256][PIECE + 1];
  const int OH = H / 2, OW = W / 2, G = C / PIECE;
  const int WH = (H + 1) / 2, WW = (W + 1) / 2;
  const idx_t total = (idx_t)N * WH * WW * G;
  const int g = threadIdx.x % G;
  float sc[PIECE], sh[PIECE], mu[PIECE], s0[PIECE], s1[PIECE];
#pragma unroll
  for (int j = 0; j < PIECE; ++j) {
    sc[j] = scale[g * PIECE + j]; sh[j] = shift[g * PIECE + j]; mu[j] = mean[g * PIECE + j];
    s0[j] = 0.f; s1[j] = 0.f;
  }
  // The window walk advances by a fixed stride of windows per iteration: its (image, row, column) decomposition is
  // computed once and carried (no divisions in the loop; the element-wise work is what limits this pass).
  const idx_t i0 = (idx_t)blockIdx.x * 256 + threadIdx.x;
  idx_t t0 = i0 / G;
  int ox = (int)(t0 % WW);  t0 /= WW;
  int oy = (int)(t0 % WH);
  idx_t n = t0 / WH;
  idx_t ts = ((idx_t)gridDim.x * 256) / G;       // (256 % G == 0: every thread keeps its channel group)
  const int sx = (int)(ts % WW);  ts /= WW;
  const int sy = (int)(ts % WH);
  const idx_t sn = ts / WH;
  for (idx_t i = i0; i < total; i += (idx_t)gridDim.x * 256) {
    const idx_t base = ((n * H + 2 * oy) * W + 2 * ox) * (idx_t)C + g * PIECE;
    const bool ex = 2 * ox + 1 < W, ey = 2 * oy + 1 < H, full = oy < OH && ox < OW;
    if (ex && ey) {
      // whole window (every window of an even frame): straight-line code, selects instead of branches.  Same values
      // and the same accumulation order (window position 0..3 per channel) as the general path below.
      const idx_t o1 = base + C, o2 = base + (idx_t)W * C, o3 = o2 + C;
      float gr[PIECE], yv[4][PIECE], od[4][PIECE];
      Vec<T>::load(dpooled + (((n * OH + oy) * OW + ox) * (idx_t)C + g * PIECE), gr);
      if (nt) {                                  // y is read for the last time here
        Vec<T>::load_nt(y + base, yv[0]);
        Vec<T>::load_nt(y + o1, yv[1]);
        Vec<T>::load_nt(y + o2, yv[2]);
        Vec<T>::load_nt(y + o3, yv[3]);
      } else {
        Vec<T>::load(y + base, yv[0]);
        Vec<T>::load(y + o1, yv[1]);
        Vec<T>::load(y + o2, yv[2]);
        Vec<T>::load(y + o3, yv[3]);
      }
      if (da_old) {
        Vec<T>::load(da_old + base, od[0]);
        Vec<T>::load(da_old + o1, od[1]);
        Vec<T>::load(da_old + o2, od[2]);
        Vec<T>::load(da_old + o3, od[3]);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int j = 0; j < PIECE; ++j) od[k][j] = 0.f;
      }
#pragma unroll
      for (int j = 0; j < PIECE; ++j) {
        float z[4], a[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          z[k] = fmaf(yv[k][j], sc[j], sh[j]);
          a[k] = ET<T>::to_f(ET<T>::from_f(fmaxf(z[k], 0.f)));
        }
        const float m = fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3]));
        const bool e0 = a[0] == m, e1 = a[1] == m, e2 = a[2] == m;
        const bool f[4] = {e0, !e0 && e1, !(e0 || e1) && e2, !(e0 || e1 || e2)};   // the FIRST maximum takes the gradient
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float d = f[k] ? od[k][j] + gr[j] : od[k][j];
          const float da = ET<T>::to_f(ET<T>::from_f(d));
          const float v = z[k] > 0.f ? da : 0.f;
          od[k][j] = v;
          s0[j] += v;
          s1[j] = fmaf(v, yv[k][j] - mu[j], s1[j]);
        }
      }
      Vec<T>::store(dz + base, od[0]);
      Vec<T>::store(dz + o1, od[1]);
      Vec<T>::store(dz + o2, od[2]);
      Vec<T>::store(dz + o3, od[3]);
    } else {
    float yv[4][PIECE], av[4][PIECE], gr[PIECE];
    bool on[4][PIECE];
#pragma unroll
    for (int j = 0; j < PIECE; ++j) gr[j] = 0.f;
    if (full) Vec<T>::load(dpooled + (((n * OH + oy) * OW + ox) * (idx_t)C + g * PIECE), gr);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool ok = ((k & 1) == 0 || ex) && ((k >> 1) == 0 || ey);
#pragma unroll
      for (int j = 0; j < PIECE; ++j) { yv[k][j] = 0.f; av[k][j] = -1.f; on[k][j] = false; }
      if (ok) {
        Vec<T>::load(y + base + ((idx_t)(k >> 1) * W + (k & 1)) * C, yv[k]);
#pragma unroll
        for (int j = 0; j < PIECE; ++j) {
          const float z = fmaf(yv[k][j], sc[j], sh[j]);
          on[k][j] = z > 0.f;
          av[k][j] = ET<T>::to_f(ET<T>::from_f(fmaxf(z, 0.f)));
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool ok = ((k & 1) == 0 || ex) && ((k >> 1) == 0 || ey);
      if (!ok) continue;
      const idx_t o = base + ((idx_t)(k >> 1) * W + (k & 1)) * C;
      float d[PIECE];
#pragma unroll
      for (int j = 0; j < PIECE; ++j) d[j] = 0.f;
      if (da_old) Vec<T>::load(da_old + o, d);
#pragma unroll
      for (int j = 0; j < PIECE; ++j) {
        if (full) {
          const float m = fmaxf(fmaxf(av[0][j], av[1][j]), fmaxf(av[2][j], av[3][j]));
          bool first = av[k][j] == m;                  // the FIRST maximum in row-major window order takes the gradient
#pragma unroll
          for (int q = 0; q < 4; ++q) first = first && !(q < k && av[q][j] == m);
          if (first) d[j] += gr[j];
        }
        const float da = ET<T>::to_f(ET<T>::from_f(d[j]));                 // da as the unfused kernels store it
        const float v = ET<T>::to_f(ET<T>::from_f(on[k][j] ? da : 0.f));   // dz as stored
        d[j] = v;
        s0[j] += v;
        s1[j] = fmaf(v, yv[k][j] - mu[j], s1[j]);
      }
      Vec<T>::store(dz + o, d);
    }
    }
    // next window of this thread
    ox += sx;
    if (ox >= WW) { ox -= WW; ++oy; }
    oy += sy;
    if (oy >= WH) { oy -= WH; ++n; }
    n += sn;
  }
  // block partial: the 256 / G threads of a channel group, in thread order
#pragma unroll
  for (int j = 0; j < PIECE; ++j) { red[0][threadIdx.x][j] = s0[j]; red[1][threadIdx.x][j] = s1[j]; }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * C; c += 256) {
    const int q = c / C, ch = c - q * C, gg = ch / PIECE, j = ch - gg * PIECE;
    float tsum = 0.f;
    for (int k = gg; k < 256; k += G) tsum += red[q][k][j];
    part[((size_t)blockIdx.x * 2 + q) * C + ch] = tsum;
  }
}

// ------------------------------------------------------------------------------- bilinear x2 (align_corners)
__device__ inline void bil_axis(int o, int n_in, int n_out, int& i0, int& i1, float& f) {
  const float src = (n_out > 1) ? o * ((float)(n_in - 1) / (float)(n_out - 1)) : 0.f;
  i0 = (int)floorf(src);
  if (i0 > n_in - 1) i0 = n_in - 1;
  i1 = i0 + 1 < n_in ? i0 + 1 : n_in - 1;
  f = src - (float)i0;
}

template <typename T>
__global__ void bilinear2x_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C) {
  constexpr int PIECE = ET<T>::PIECE;
  const int OH = 2 * H, OW = 2 * W, G = C / PIECE;
  const long long total = (long long)N * OH * OW * G;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(i % G);
    long long t = i / G;
    const int ox = (int)(t % OW);  t /= OW;
    const int oy = (int)(t % OH);
    const long long n = t / OH;
    int y0, y1, x0, x1; float fy, fx;
    bil_axis(oy, H, OH, y0, y1, fy);
    bil_axis(ox, W, OW, x0, x1, fx);
    float a[PIECE], b[PIECE], c[PIECE], d[PIECE];
    const T* p = x + n * H * (long long)W * C + g * PIECE;
    Vec<T>::load(p + ((long long)y0 * W + x0) * C, a);
    Vec<T>::load(p + ((long long)y0 * W + x1) * C, b);
    Vec<T>::load(p + ((long long)y1 * W + x0) * C, c);
    Vec<T>::load(p + ((long long)y1 * W + x1) * C, d);
#pragma unroll
    for (int j = 0; j < PIECE; ++j) {
      const float top = a[j] * (1.f - fx) + b[j] * fx, bot = c[j] * (1.f - fx) + d[j] * fx;
      a[j] = top * (1.f - fy) + bot * fy;
    }
    Vec<T>::store(y + i * PIECE, a);
  }
}

// gather form of the adjoint: each input pixel sums the (<= 3x3) output pixels that read it.
template <typename T>
__global__ void bilinear2x_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int H, int W, int C) {
  constexpr int PIECE = ET<T>::PIECE;
  const int OH = 2 * H, OW = 2 * W, G = C / PIECE;
  const long long total = (long long)N * H * W * G;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(i % G);
    long long t = i / G;
    const int ix = (int)(t % W);  t /= W;
    const int iy = (int)(t % H);
    const long long n = t / H;
    float acc[PIECE];
#pragma unroll
    for (int j = 0; j < PIECE; ++j) acc[j] = 0.f;
    // output rows whose source coordinate lies in (iy-1, iy+1)
    const int oy_lo = max(0, 2 * iy - 2), oy_hi = min(OH - 1, 2 * iy + 3);
    const int ox_lo = max(0, 2 * ix - 2), ox_hi = min(OW - 1, 2 * ix + 3);
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      int y0, y1; float fy;
      bil_axis(oy, H, OH, y0, y1, fy);
      float wy = 0.f;
      if (y0 == iy) wy += 1.f - fy;
      if (y1 == iy) wy += fy;
      if (wy == 0.f) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        int x0, x1; float fx;
        bil_axis(ox, W, OW, x0, x1, fx);
        float wx = 0.f;
        if (x0 == ix) wx += 1.f - fx;
        if (x1 == ix) wx += fx;
        if (wx == 0.f) continue;
        float v[PIECE];
        Vec<T>::load(dy + (((n * OH + oy) * OW + ox) * (long long)C + g * PIECE), v);
#pragma unroll
        for (int j = 0; j < PIECE; ++j) acc[j] = fmaf(v[j], wy * wx, acc[j]);
      }
    }
    Vec<T>::store(dx + i * PIECE, acc);
  }
}

// ------------------------------------------------------------------------------- Adam
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long long n4, float lr, float b1, float b2, float omb1,
                            float omb2, float eps, float wd, float gscale, float bc1, float bc2_sqrt) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4;
       i += (long long)gridDim.x * blockDim.x) {
    f32x4 pp = reinterpret_cast<f32x4*>(p)[i], gg = reinterpret_cast<const f32x4*>(g)[i];
    f32x4 mm = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gr = fmaf(wd, pp[j], gg[j] * gscale);
      mm[j] = b1 * mm[j] + omb1 * gr;
      vv[j] = b2 * vv[j] + omb2 * gr * gr;
      const float denom = sqrtf(vv[j]) / bc2_sqrt + eps;
      pp[j] -= (lr / bc1) * (mm[j] / denom);
    }
    reinterpret_cast<f32x4*>(p)[i] = pp;
    reinterpret_cast<f32x4*>(m)[i] = mm;
    reinterpret_cast<f32x4*>(v)[i] = vv;
  }
}

}  // namespace

extern "C" int32_t unet_nchw_to_nhwc(const float* src, void* dst, int32_t n, int32_t c, int32_t h, int32_t w,
                                     int32_t c_pad, int32_t dtype, void* stream) {
  UNET_REQUIRE(src && dst, UNET_ERR_BAD_ARG, "unet_nchw_to_nhwc: null pointer");
  UNET_REQUIRE(n > 0 && c > 0 && h > 0 && w > 0 && c_pad >= c && c_pad % 8 == 0, UNET_ERR_BAD_ARG,
               "unet_nchw_to_nhwc: bad dims (c=%d c_pad=%d, c_pad must be a multiple of 8)", c, c_pad);
  hipStream_t s = (hipStream_t)stream;
  const long long total = (long long)n * h * w;
  ProfScope prof(UNET_K_PACK, 0.0, s);
  if (dtype == UNET_BF16)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3(ew_blocks(total)), dim3(256), 0, s, src, (bf16_t*)dst, n, c, h, w, c_pad);
  else
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(ew_blocks(total)), dim3(256), 0, s, src, (float*)dst, n, c, h, w, c_pad);
  return unet_check_launch("nchw_to_nhwc_kernel");
}

extern "C" int32_t unet_nhwc_to_nchw(const void* src, float* dst, int32_t n, int32_t c, int32_t h, int32_t w,
                                     int32_t c_pad, int32_t dtype, void* stream) {
  UNET_REQUIRE(src && dst, UNET_ERR_BAD_ARG, "unet_nhwc_to_nchw: null pointer");
  UNET_REQUIRE(n > 0 && c > 0 && h > 0 && w > 0 && c_pad >= c, UNET_ERR_BAD_ARG, "unet_nhwc_to_nchw: bad dims");
  hipStream_t s = (hipStream_t)stream;
  const long long total = (long long)n * h * w * c;
  ProfScope prof(UNET_K_PACK, 0.0, s);
  if (dtype == UNET_BF16)
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, dim3(ew_blocks(total)), dim3(256), 0, s, (const bf16_t*)src, dst, n, c, h, w, c_pad);
  else
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(ew_blocks(total)), dim3(256), 0, s, (const float*)src, dst, n, c, h, w, c_pad);
  return unet_check_launch("nhwc_to_nchw_kernel");
}

extern "C" int32_t unet_pack_weight(const float* w, void* out, int32_t c_out, int32_t c_in, int32_t rows,
                                    int32_t k, int32_t mode, int32_t dtype, void* stream) {
  UNET_REQUIRE(w && out, UNET_ERR_BAD_ARG, "unet_pack_weight: null pointer");
  UNET_REQUIRE(c_out > 0 && c_in > 0 && rows > 0 && k > 0 && mode >= 0 && mode <= 3, UNET_ERR_BAD_ARG,
               "unet_pack_weight: bad dims/mode");
  const int gemm_rows = (mode == UNET_PACK_CONV_FWD || mode == UNET_PACK_CONVT_FWD) ? c_out : c_in;
  const int gemm_k = (mode == UNET_PACK_CONV_FWD || mode == UNET_PACK_CONVT_FWD) ? c_in : c_out;
  UNET_REQUIRE(rows >= gemm_rows && k >= gemm_k, UNET_ERR_BAD_ARG, "unet_pack_weight: padded dims too small");
  const int taps = (mode <= UNET_PACK_CONV_DGRAD) ? 9 : 4;
  const long long total = (long long)taps * rows * k;
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(UNET_K_PACK, 0.0, s);
  if (dtype == UNET_BF16)
    hipLaunchKernelGGL(pack_weight_kernel<bf16_t>, dim3(ew_blocks(total)), dim3(256), 0, s, w, (bf16_t*)out, c_out, c_in, rows, k, mode, total, (const float*)nullptr);
  else
    hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(ew_blocks(total)), dim3(256), 0, s, w, (float*)out, c_out, c_in, rows, k, mode, total, (const float*)nullptr);
  return unet_check_launch("pack_weight_kernel");
}

extern "C" int32_t unet_pack_conv_weight_folded(const float* w, const float* scale, void* out, int32_t c_out, int32_t c_in,
                                                int32_t rows, int32_t k, int32_t dtype, void* stream) {
  UNET_REQUIRE(w && scale && out, UNET_ERR_BAD_ARG, "unet_pack_conv_weight_folded: null pointer");
  UNET_REQUIRE(c_out > 0 && c_in > 0 && rows >= c_out && k >= c_in, UNET_ERR_BAD_ARG, "unet_pack_conv_weight_folded: bad dims");
  const long long total = 9LL * rows * k;
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(UNET_K_PACK, 0.0, s);
  if (dtype == UNET_BF16)
    hipLaunchKernelGGL(pack_weight_kernel<bf16_t>, dim3(ew_blocks(total)), dim3(256), 0, s, w, (bf16_t*)out, c_out, c_in, rows, k,
                       (int)UNET_PACK_CONV_FWD, total, scale);
  else
    hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(ew_blocks(total)), dim3(256), 0, s, w, (float*)out, c_out, c_in, rows, k,
                       (int)UNET_PACK_CONV_FWD, total, scale);
  return unet_check_launch("pack_weight_kernel");
}

extern "C" int32_t unet_pack_weights_batched(const unet_pack_desc* descs, int32_t n, int32_t dtype, void* stream) {
  UNET_REQUIRE(descs && n > 0 && n <= 65535, UNET_ERR_BAD_ARG, "unet_pack_weights_batched: bad argument");
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(UNET_K_PACK, 0.0, s);
  if (dtype == UNET_BF16)
    hipLaunchKernelGGL(pack_weights_batched_kernel<bf16_t>, dim3(224, n), dim3(256), 0, s, descs);
  else
    hipLaunchKernelGGL(pack_weights_batched_kernel<float>, dim3(224, n), dim3(256), 0, s, descs);
  return unet_check_launch("pack_weights_batched_kernel");
}

#define EW_DISPATCH(NAME, KERN, TOTAL, ...)                                                                 \
  do {                                                                                                        \
    hipStream_t s = (hipStream_t)stream;                                                                      \
    ProfScope prof(UNET_K_POOL, 0.0, s);                                                                      \
    if (dtype == UNET_BF16) {                                                                                 \
      typedef bf16_t T;                                                                                       \
      const long long total = (TOTAL) / 8;                                                                    \
      hipLaunchKernelGGL(KERN<bf16_t>, dim3(ew_blocks(total)), dim3(256), 0, s, __VA_ARGS__);                 \
    } else {                                                                                                  \
      typedef float T;                                                                                        \
      const long long total = (TOTAL) / 4;                                                                    \
      hipLaunchKernelGGL(KERN<float>, dim3(ew_blocks(total)), dim3(256), 0, s, __VA_ARGS__);                  \
    }                                                                                                         \
    return unet_check_launch(NAME);                                                                           \
  } while (0)

extern "C" int32_t unet_maxpool2_fwd(int32_t dtype, const void* x, int32_t n, int32_t h, int32_t w, int32_t c,
                                     void* y, void* stream) {
  UNET_REQUIRE(x && y, UNET_ERR_BAD_ARG, "unet_maxpool2_fwd: null pointer");
  UNET_REQUIRE(n > 0 && h >= 2 && w >= 2 && c > 0 && c % 8 == 0, UNET_ERR_UNSUPPORTED, "unet_maxpool2_fwd: dims");
  EW_DISPATCH("maxpool2_fwd_kernel", maxpool2_fwd_kernel, (long long)n * (h / 2) * (w / 2) * c, (const T*)x, (T*)y, n, h, w, c);
}

extern "C" int32_t unet_maxpool2_bwd(int32_t dtype, const void* x, const void* dy, int32_t n, int32_t h,
                                     int32_t w, int32_t c, void* dx, int32_t accumulate, void* stream) {
  UNET_REQUIRE(x && dy && dx, UNET_ERR_BAD_ARG, "unet_maxpool2_bwd: null pointer");
  UNET_REQUIRE(n > 0 && h >= 2 && w >= 2 && c > 0 && c % 8 == 0, UNET_ERR_UNSUPPORTED, "unet_maxpool2_bwd: dims");
  EW_DISPATCH("maxpool2_bwd_kernel", maxpool2_bwd_kernel, (long long)n * ((h + 1) / 2) * ((w + 1) / 2) * c,
              (const T*)x, (const T*)dy, (T*)dx, n, h, w, c, accumulate);
}


extern "C" int32_t unet_bn_relu_pool_supported(int32_t dtype, int32_t c) {
  const int piece = dtype == UNET_BF16 ? 8 : 4;
  return (c > 0 && c % piece == 0 && 256 % (c / piece) == 0) ? 1 : 0;
}

extern "C" int32_t unet_bn_relu_pool_fwd(int32_t dtype, const void* y, int32_t n, int32_t h, int32_t w, int32_t c,
                                         const float* scale, const float* shift, void* a, void* pooled, void* stream) {
  UNET_REQUIRE(y && scale && shift && a && pooled, UNET_ERR_BAD_ARG, "unet_bn_relu_pool_fwd: null pointer");
  UNET_REQUIRE(n > 0 && h >= 2 && w >= 2 && unet_bn_relu_pool_supported(dtype, c), UNET_ERR_UNSUPPORTED,
               "unet_bn_relu_pool_fwd: c=%d h=%d w=%d", c, h, w);
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(UNET_K_BN, 0.0, s, "bn_relu_pool_fwd_kernel");
  const int ewnt = (unet_tuning().ew_var == '1' || unet_tuning().ew_var == '3') ? 1 : 0;   // streaming loads of y: A/B hook (+-0)
  const long long total = (long long)n * ((h + 1) / 2) * ((w + 1) / 2) * (c / (dtype == UNET_BF16 ? 8 : 4));
  const bool small = (long long)n * h * w * c < 0x7FFFFFFFLL;       // 32-bit element indices (no 64-bit divisions)
  if (dtype == UNET_BF16) {
    if (small)
      hipLaunchKernelGGL((bn_relu_pool_fwd_kernel<bf16_t, int>), dim3(ew_blocks(total)), dim3(256), 0, s, (const bf16_t*)y, scale,
                         shift, (bf16_t*)a, (bf16_t*)pooled, n, h, w, c, ewnt);
    else
      hipLaunchKernelGGL((bn_relu_pool_fwd_kernel<bf16_t, long long>), dim3(ew_blocks(total)), dim3(256), 0, s, (const bf16_t*)y,
                         scale, shift, (bf16_t*)a, (bf16_t*)pooled, n, h, w, c, ewnt);
  } else {
    hipLaunchKernelGGL((bn_relu_pool_fwd_kernel<float, long long>), dim3(ew_blocks(total)), dim3(256), 0, s, (const float*)y, scale,
                       shift, (float*)a, (float*)pooled, n, h, w, c, ewnt);
  }
  return unet_check_launch("bn_relu_pool_fwd_kernel");
}

namespace { constexpr int POOL_BWD_MAX_BLOCKS = 256 * 8; }     // = the partial-sum capacity callers allocate
extern "C" size_t unet_bn_relu_pool_max_parts(void) { return POOL_BWD_MAX_BLOCKS; }

extern "C" int32_t unet_bn_relu_pool_bwd(int32_t dtype, const void* y, const void* dpooled, const void* da_old, int32_t n,
                                         int32_t h, int32_t w, int32_t c, const float* scale, const float* shift,
                                         const float* mean, void* dz, float* partial, int32_t* n_parts, void* stream) {
  UNET_REQUIRE(y && dpooled && scale && shift && mean && dz && partial && n_parts, UNET_ERR_BAD_ARG,
               "unet_bn_relu_pool_bwd: null pointer");
  UNET_REQUIRE(n > 0 && h >= 2 && w >= 2 && unet_bn_relu_pool_supported(dtype, c), UNET_ERR_UNSUPPORTED,
               "unet_bn_relu_pool_bwd: c=%d h=%d w=%d", c, h, w);
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(UNET_K_POOL, 0.0, s, "bn_relu_pool_bwd_kernel");
  const int ewnt = (unet_tuning().ew_var == '1' || unet_tuning().ew_var == '3') ? 1 : 0;   // streaming loads of y: A/B hook (+-0)
  const long long total = (long long)n * ((h + 1) / 2) * ((w + 1) / 2) * (c / (dtype == UNET_BF16 ? 8 : 4));
  const int blocks = (int)std::min<long long>(cdiv64(total, 256), POOL_BWD_MAX_BLOCKS);   // one partial per block
  const bool small = (long long)n * h * w * c < 0x7FFFFFFFLL;
  if (dtype == UNET_BF16) {
    if (small)
      hipLaunchKernelGGL((bn_relu_pool_bwd_kernel<bf16_t, int>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)y,
                         (const bf16_t*)dpooled, (const bf16_t*)da_old, scale, shift, mean, (bf16_t*)dz, partial, n, h, w, c, ewnt);
    else
      hipLaunchKernelGGL((bn_relu_pool_bwd_kernel<bf16_t, long long>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)y,
                         (const bf16_t*)dpooled, (const bf16_t*)da_old, scale, shift, mean, (bf16_t*)dz, partial, n, h, w, c, ewnt);
  } else {
    hipLaunchKernelGGL((bn_relu_pool_bwd_kernel<float, long long>), dim3(blocks), dim3(256), 0, s, (const float*)y,
                       (const float*)dpooled, (const float*)da_old, scale, shift, mean, (float*)dz, partial, n, h, w, c, ewnt);
  }
  *n_parts = blocks;
  return unet_check_launch("bn_relu_pool_bwd_kernel");
}

extern "C" int32_t unet_upsample_bilinear2x_fwd(int32_t dtype, const void* x, int32_t n, int32_t h, int32_t w,
                                                int32_t c, void* y, void* stream) {
  UNET_REQUIRE(x && y, UNET_ERR_BAD_ARG, "unet_upsample_bilinear2x_fwd: null pointer");
  UNET_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && c % 8 == 0, UNET_ERR_UNSUPPORTED, "unet_upsample_bilinear2x_fwd: dims");
  EW_DISPATCH("bilinear2x_fwd_kernel", bilinear2x_fwd_kernel, (long long)n * 4 * h * w * c, (const T*)x, (T*)y, n, h, w, c);
}

extern "C" int32_t unet_upsample_bilinear2x_bwd(int32_t dtype, const void* dy, int32_t n, int32_t h, int32_t w,
                                                int32_t c, void* dx, void* stream) {
  UNET_REQUIRE(dy && dx, UNET_ERR_BAD_ARG, "unet_upsample_bilinear2x_bwd: null pointer");
  UNET_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && c % 8 == 0, UNET_ERR_UNSUPPORTED, "unet_upsample_bilinear2x_bwd: dims");
  EW_DISPATCH("bilinear2x_bwd_kernel", bilinear2x_bwd_kernel, (long long)n * h * w * c, (const T*)dy, (T*)dx, n, h, w, c);
}

// ------------------------------------------------------------------------------- Dropout2d / anomaly score
namespace {

// y[n][p][c] = x[n][p][c] * scale[n][c]   (nn.Dropout2d on the bottleneck of SegmentationUNet, src/model.py:129,146:
// scale = bernoulli(1-p) / (1-p) per (image, channel); the same kernel is its backward)
template <typename T>
__global__ void channel_scale_kernel(const T* __restrict__ x, const float* __restrict__ scale, T* __restrict__ y,
                                     long long hw, int C, long long total) {
  constexpr int PIECE = ET<T>::PIECE;
  const int G = C / PIECE;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(i % G);
    const long long n = i / G / hw;
    float v[PIECE];
    Vec<T>::load(x + i * PIECE, v);
#pragma unroll
    for (int j = 0; j < PIECE; ++j) v[j] *= scale[n * C + g * PIECE + j];
    Vec<T>::store(y + i * PIECE, v);
  }
}

// score[n][p] = mean_c (recon - image)^2 (or |.|) over fp32 NCHW planes (compute_anomaly_score, src/utils.py:205-215);
// image_score[n] = mean_p score[n][p] through ordered block partials (the image-level score of src/test.py)
__global__ __launch_bounds__(256) void anomaly_score_kernel(const float* __restrict__ recon, const float* __restrict__ image,
                                                           int C, long long hw, int l1, float* __restrict__ score,
                                                           float* __restrict__ part) {
  __shared__ float red[4];
  const int n = blockIdx.y;
  float acc = 0.f;
  for (long long q = blockIdx.x * 256LL + threadIdx.x; q < hw; q += (long long)gridDim.x * 256) {
    float s = 0.f;
    for (int c = 0; c < C; ++c) {
      const float d = recon[((long long)n * C + c) * hw + q] - image[((long long)n * C + c) * hw + q];
      s += l1 ? fabsf(d) : d * d;
    }
    s /= (float)C;
    score[(long long)n * hw + q] = s;
    acc += s;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[(size_t)n * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void anomaly_score_finalize_kernel(const float* __restrict__ part, int N, int nb, long long hw,
                                              float* __restrict__ out) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  double s = 0.0;
  for (int b = 0; b < nb; ++b) s += (double)part[(size_t)n * nb + b];
  out[n] = (float)(s / (double)hw);
}

}  // namespace

extern "C" int32_t unet_channel_scale(int32_t dtype, const void* x, const float* scale, int32_t n, int64_t hw, int32_t c,
                                      void* y, void* stream) {
  UNET_REQUIRE(x && scale && y, UNET_ERR_BAD_ARG, "unet_channel_scale: null pointer");
  UNET_REQUIRE(n > 0 && hw > 0 && c > 0 && c % 8 == 0, UNET_ERR_UNSUPPORTED, "unet_channel_scale: c=%d", c);
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(UNET_K_POOL, 0.0, s);
  if (dtype == UNET_BF16) {
    const long long total = (long long)n * hw * c / 8;
    hipLaunchKernelGGL(channel_scale_kernel<bf16_t>, dim3(ew_blocks(total)), dim3(256), 0, s, (const bf16_t*)x, scale,
                       (bf16_t*)y, (long long)hw, c, total);
  } else {
    const long long total = (long long)n * hw * c / 4;
    hipLaunchKernelGGL(channel_scale_kernel<float>, dim3(ew_blocks(total)), dim3(256), 0, s, (const float*)x, scale,
                       (float*)y, (long long)hw, c, total);
  }
  return unet_check_launch("channel_scale_kernel");
}

extern "C" size_t unet_anomaly_score_workspace(int32_t n, int64_t hw) {
  (void)hw;
  return (size_t)n * 64 * sizeof(float);
}

extern "C" int32_t unet_anomaly_score(const float* recon, const float* image, int32_t n, int32_t c, int64_t hw,
                                      int32_t l1, float* score, float* image_score, void* workspace,
                                      size_t workspace_bytes, void* stream) {
  UNET_REQUIRE(recon && image && score && image_score && workspace, UNET_ERR_BAD_ARG, "unet_anomaly_score: null pointer");
  UNET_REQUIRE(n > 0 && c > 0 && hw > 0, UNET_ERR_BAD_ARG, "unet_anomaly_score: bad dims");
  UNET_REQUIRE(workspace_bytes >= unet_anomaly_score_workspace(n, hw), UNET_ERR_WORKSPACE, "unet_anomaly_score: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  long long nb = cdiv64(hw, 256 * 4);
  if (nb > 64) nb = 64;
  ProfScope prof(UNET_K_LOSS, 0.0, s);
  hipLaunchKernelGGL(anomaly_score_kernel, dim3((unsigned)nb, n), dim3(256), 0, s, recon, image, c, (long long)hw, l1, score,
                     (float*)workspace);
  int32_t rc = unet_check_launch("anomaly_score_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(anomaly_score_finalize_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, (const float*)workspace, n, (int)nb,
                     (long long)hw, image_score);
  return unet_check_launch("anomaly_score_finalize_kernel");
}

// All parameter tensors of a model in ONE launch: descs[t] = {p, g, m, v, n}; chunks[b] = {tensor, first element}: block b
// updates up to ADAM_CHUNK elements of its tensor (vector path on the 16-byte-aligned body, scalar tail).
namespace {
constexpr int ADAM_CHUNK = 4096;
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float lr_bc1, float b1, float b2, float omb1,
                                         float omb2, float eps, float wd, float gscale, float bc2_sqrt, int decoupled) {
  float gr = g * gscale;
  if (!decoupled) gr = fmaf(wd, p, gr);               // Adam: the L2 term joins the gradient (torch.optim.Adam)
  m = b1 * m + omb1 * gr;
  v = b2 * v + omb2 * gr * gr;
  const float denom = sqrtf(v) / bc2_sqrt + eps;
  p -= lr_bc1 * (m / denom);
}
__global__ __launch_bounds__(256) void adam_multi_kernel(const unet_adam_desc* __restrict__ descs,
                                                         const unet_adam_chunk* __restrict__ chunks, float lr, float b1,
                                                         float b2, float omb1, float omb2, float eps, float wd,
                                                         float gscale, float bc1, float bc2_sqrt, int decoupled) {
  const unet_adam_chunk ck = chunks[blockIdx.x];
  const unet_adam_desc d = descs[ck.tensor];
  const long long begin = ck.first, end = begin + ADAM_CHUNK < d.n ? begin + ADAM_CHUNK : d.n;
  const float lr_bc1 = lr / bc1;
  const float decay = decoupled ? 1.f - lr * wd : 1.f;             // AdamW: p *= 1 - lr*wd before the update
  const long long vend = begin + ((end - begin) / 4) * 4;
  for (long long i = begin + threadIdx.x * 4LL; i < vend; i += 1024) {
    f32x4 pp = *reinterpret_cast<f32x4*>(d.p + i), gg = *reinterpret_cast<const f32x4*>(d.g + i);
    f32x4 mm = *reinterpret_cast<f32x4*>(d.m + i), vv = *reinterpret_cast<f32x4*>(d.v + i);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float pj = pp[j] * decay, mj = mm[j], vj = vv[j];
      adam_one(pj, gg[j], mj, vj, lr_bc1, b1, b2, omb1, omb2, eps, wd, gscale, bc2_sqrt, decoupled);
      pp[j] = pj; mm[j] = mj; vv[j] = vj;
    }
    *reinterpret_cast<f32x4*>(d.p + i) = pp;
    *reinterpret_cast<f32x4*>(d.m + i) = mm;
    *reinterpret_cast<f32x4*>(d.v + i) = vv;
  }
  for (long long i = vend + threadIdx.x; i < end; i += 256) {
    float pp = d.p[i] * decay, mm = d.m[i], vv = d.v[i];
    adam_one(pp, d.g[i], mm, vv, lr_bc1, b1, b2, omb1, omb2, eps, wd, gscale, bc2_sqrt, decoupled);
    d.p[i] = pp; d.m[i] = mm; d.v[i] = vv;
  }
}
}  // namespace

extern "C" int32_t unet_adam_chunk_elems(void) { return ADAM_CHUNK; }

extern "C" int32_t unet_adam_multi(const unet_adam_desc* descs, const unet_adam_chunk* chunks, int32_t n_chunks, float lr,
                                   double beta1, double beta2, float eps, float weight_decay, float grad_scale, int32_t step,
                                   int32_t decoupled, void* stream) {
  UNET_REQUIRE(descs && chunks && n_chunks > 0 && step >= 1, UNET_ERR_BAD_ARG, "unet_adam_multi: bad argument");
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(UNET_K_OTHER, 0.0, s, "adam_multi_kernel");
  hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)n_chunks), dim3(256), 0, s, descs, chunks, lr, (float)beta1, (float)beta2,
                     (float)(1.0 - beta1), (float)(1.0 - beta2), eps, weight_decay, grad_scale, (float)bc1, (float)sqrt(bc2), decoupled);
  return unet_check_launch("adam_multi_kernel");
}

extern "C" int32_t unet_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                                  float lr, double beta1, double beta2, float eps, float weight_decay,
                                  float grad_scale, int32_t step, void* stream) {
  UNET_REQUIRE(param && grad && exp_avg && exp_avg_sq, UNET_ERR_BAD_ARG, "unet_adam_step: null pointer");
  UNET_REQUIRE(n > 0 && n % 4 == 0 && step >= 1, UNET_ERR_BAD_ARG, "unet_adam_step: n=%lld (must be a multiple of 4), step=%d",
               (long long)n, step);
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(UNET_K_OTHER, 0.0, s);
  hipLaunchKernelGGL(adam_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, s, param, grad, exp_avg, exp_avg_sq,
                     (long long)(n / 4), lr, (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), eps, weight_decay, grad_scale, (float)bc1,
                     (float)sqrt(bc2));
  return unet_check_launch("adam_kernel");
}

// ---- ToTensor + Normalize (+ horizontal flip) of uint8 HWC image batches (src/dataset.py:134-146,
// src/kolektorsdd_dataset.py:133-150): out[n][c][y][x] = (u8[n][y][x'][c] / 255 - mean[c]) / std[c], x' = W-1-x when
// flip[n].  The same fp32 operations in the same order as torchvision's ToTensor().div(255) and Normalize.sub_().div_():
// bit-identical to the host transform.
namespace {
__global__ __launch_bounds__(256) void preprocess_u8_kernel(const unsigned char* __restrict__ src,
                                                            const unsigned char* __restrict__ flip,
                                                            float* __restrict__ dst, int N, int H, int W,
                                                            float m0, float m1, float m2, float s0, float s1, float s2) {
  const long long total = (long long)N * H * W;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int x = (int)(i % W);
    const long long r = i / W;
    const int y = (int)(r % H), n = (int)(r / H);
    const int xs = (flip && flip[n]) ? W - 1 - x : x;
    const unsigned char* p = src + (((long long)n * H + y) * W + xs) * 3;
    const long long plane = (long long)H * W;
    float* o = dst + (long long)n * 3 * plane + (long long)y * W + x;
    o[0] = ((float)p[0] / 255.f - m0) / s0;
    o[plane] = ((float)p[1] / 255.f - m1) / s1;
    o[2 * plane] = ((float)p[2] / 255.f - m2) / s2;
  }
}
}  // namespace

extern "C" int32_t unet_preprocess_u8(const uint8_t* images_hwc, const uint8_t* flip, float* out_nchw, int32_t n, int32_t h,
                                      int32_t w, const float* mean3, const float* std3, void* stream) {
  UNET_REQUIRE(images_hwc && out_nchw && mean3 && std3, UNET_ERR_BAD_ARG, "unet_preprocess_u8: null pointer");
  UNET_REQUIRE(n > 0 && h > 0 && w > 0, UNET_ERR_BAD_ARG, "unet_preprocess_u8: bad dims");
  const long long total = (long long)n * h * w;
  const int blocks = (int)std::min<long long>(cdiv64(total, 256), 256 * 16);
  hipLaunchKernelGGL(preprocess_u8_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, images_hwc, flip, out_nchw, n, h, w,
                     mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
  return unet_check_launch("preprocess_u8_kernel");
}
