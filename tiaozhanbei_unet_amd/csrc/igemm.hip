// Implicit-GEMM convolution on MFMA for gfx950 (CDNA4).
//
// One kernel template serves four operators of the hot path:
//   conv3x3 forward      (nn.Conv2d 3x3 p1, /root/reference/src/model.py:14,17)      TAPS=9
//   conv3x3 data grad    (same kernel, flipped/transposed packed weights)              TAPS=9
//   convT2x2 forward     (nn.ConvTranspose2d k2 s2, src/model.py:51): 4 one-tap GEMMs
//                        whose epilogue scatters to (2i+k, 2j+l) (pixel shuffle)       TAPS=1, omul=2
//   convT2x2 data grad   one-tap GEMM whose K dimension gathers the 4 sub-positions    TAPS=1, gtaps=4
//
// GEMM view: D[co][pixel] = sum_k W[co][k] * X[pixel][k], k = (tap, ci).
//   MFMA "A" operand = weights (rows = output channel), "B" operand = pixels, so every lane
//   ends up owning ONE pixel and 4-channel runs of it -> 8/16-byte NHWC stores, no LDS transpose.
// Tiling: block = 8x16 output pixels x BN output channels, 4 waves (64 lanes each),
//   wave tile 64 co x (64|32) px out of 32x32 MFMA tiles; fp32 accumulate.
// Staging: per Cin-chunk the (8+2)x(16+2) halo'd input patch is staged ONCE in LDS and reused
//   by all 9 taps (tap = constant LDS offset); per tap a BN x chunk weight slab is staged into
//   a 2-deep LDS ring, its global loads issued one tap ahead (register staged, T14-style).
//   LDS rows are padded by 16 B so ds_read_b128 fragments are bank-conflict free.
// The skip concat (src/model.py:65) and centre pad (src/model.py:57-61) are two source views:
//   a chunk reads from src[0] or src[1]; out-of-view pixels read as zero.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

struct DView { const char* p; int C, H, W, oy, ox; };
struct DViewW { char* p; int C, H, W, oy, ox; };

struct IgemmParams {
  DView src[2];
  DViewW dst[2];
  int N, H, W;      // GEMM pixel grid (the logical frame)
  int Ctot;         // input channels per (gather) tap = src[0].C + src[1].C
  int Cout;         // GEMM rows
  int wK;           // weight row length in elements
  const char* w;
  const float* bias;
  int dst_split;
  int accumulate;
  int relu;         // != 0: max(., 0) after the bias (inference with BatchNorm folded into weights + bias)
  int imul, gtaps;  // input position = frame*imul + gather tap (convT dgrad: 2, 4)
  int omul, nZ;     // output position = frame*omul + z tap     (convT fwd:   2, 4)
  int zdiv;         // > 0: GEMM row = z*zdiv + co (convT fwd as one GEMM with 4*Cout rows)
  float* stats;     // != NULL: per-block BatchNorm partials [part][2][Cout] written by the epilogue
  int tilesX, tilesY, nCo;
  // data gradient fused with the ReLU mask + BatchNorm-backward sums of the layer that PRODUCED this convolution's
  // input (kernels instantiated with BNBWD): bn_y = that layer's raw conv output [N][H][W][Cout] (same geometry as
  // dst[0]), bn_scale / bn_shift / bn_mean = its forward coefficients.  The epilogue stores dz = dx * [fma(y, scale,
  // shift) > 0] and the partial sums (sum dz, sum dz * (y - mean)) go where the forward statistics would (stats).
  const char* bn_y;
  const float* bn_scale;
  const float* bn_shift;
  const float* bn_mean;
  int ws_stagger;   // conv3_ws16_kernel: the two waves of a SIMD issue their patch DMAs at opposite ends of a tile (UNET_WS_STG=0: off, 1: without the deferred stores)
  int co_il;        // conv3_pdma: channel tiles interleaved per pixel tile in the work order (1, 2 or 4; see pdma_item)
  int pdma_stagger; // conv3_pdma (lock-step): DMA issues of a SIMD's two waves at opposite ends of a tap
  int pdma_dense;   // conv3_pdma: every destination view covers the frame at offset 0 (scalar output addressing)
  int pdma_dense_src; // conv3_pdma: every source view covers the frame at offset 0, one channel stride (scalar patch addressing)
};

constexpr int TH = 8, TW = 16, NPIX = TH * TW;

template <typename T, int TAPS, int BN, int KG>
struct Cfg {
  static constexpr int R = (TAPS == 9) ? 3 : 1;
  static constexpr int HH = TH + R - 1, HW = TW + R - 1;
  static constexpr int CHB = KG * 32;              // chunk bytes per pixel / weight row
  static constexpr int PSTR = CHB + 16;            // padded LDS row stride
  static constexpr int PPP = CHB / 16;             // 16-byte pieces per row
  static constexpr int A_BUFS = (TAPS == 1) ? 2 : 1;
  static constexpr int A_BYTES = HH * HW * PSTR;
  static constexpr int B_BYTES = BN * PSTR;
  static constexpr int LDS = A_BUFS * A_BYTES + 2 * B_BYTES;
  static constexpr int CK = KG * ET<T>::KGC;       // channels per chunk
  static constexpr int WCO = BN / 64, WPX = 4 / WCO, PXT = NPIX / (32 * WPX);
  static constexpr int NAP = (HH * HW * PPP + 255) / 256;
  static constexpr int NBP = (BN * PPP + 255) / 256;
  static constexpr bool BN_PIECES_EXACT = (BN * PPP) % 256 == 0;
};

template <typename T, int TAPS, int BN, int KG>
__global__ __launch_bounds__(256, 2) void igemm_kernel(const IgemmParams P) {
  using C = Cfg<T, TAPS, BN, KG>;
  using E = ET<T>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sA = smem;
  char* const sB = smem + C::A_BUFS * C::A_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wco = wave % C::WCO, wpx = wave / C::WCO;
  const int l31 = lane & 31, hh = lane >> 5;

  // ---- XCD-aware block decode: consecutive logical ids (same pixel tile, all co tiles) share an XCD
  int logical;
  {
    const int total = gridDim.x, b = blockIdx.x;
    const int xcd = b & 7, slot = b >> 3, q = total >> 3, r = total & 7;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int cot = logical % P.nCo;
  int t = logical / P.nCo;
  const int z = t % P.nZ;  t /= P.nZ;
  const int txi = t % P.tilesX;  t /= P.tilesX;
  const int tyi = t % P.tilesY;
  const int n = t / P.tilesY;
  const int ty0 = tyi * TH, tx0 = txi * TW;
  const int co0 = cot * BN;

  f32x16 acc[2][C::PXT];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < C::PXT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // ---- per-lane LDS fragment bases
  int aoff[2];   // weight rows (MFMA A operand)
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) aoff[ct] = (wco * 64 + ct * 32 + l31) * C::PSTR + hh * 16;
  int boff[C::PXT];  // pixel rows (MFMA B operand)
#pragma unroll
  for (int pt = 0; pt < C::PXT; ++pt) {
    const int m = wpx * (32 * C::PXT) + pt * 32 + l31;
    boff[pt] = ((m >> 4) * C::HW + (m & 15)) * C::PSTR + hh * 16;
  }

  const int cpt = P.Ctot / C::CK;          // chunks per gather tap
  const int nchunks = P.gtaps * cpt;

  u32x4 breg[C::NBP];
  auto load_b = [&](int tap, int chunk) {
#pragma unroll
    for (int i = 0; i < C::NBP; ++i) {
      const int id = tid + i * 256;
      if (C::BN_PIECES_EXACT || id < BN * C::PPP) {
        const int row = id / C::PPP, part = id % C::PPP;
        const size_t e = ((size_t)((z * TAPS + tap) * P.Cout + co0 + row)) * P.wK + (size_t)chunk * C::CK;
        breg[i] = *reinterpret_cast<const u32x4*>(P.w + e * E::ES + part * 16);
      }
    }
  };
  auto store_b = [&](int buf) {
#pragma unroll
    for (int i = 0; i < C::NBP; ++i) {
      const int id = tid + i * 256;
      if (C::BN_PIECES_EXACT || id < BN * C::PPP) {
        const int row = id / C::PPP, part = id % C::PPP;
        *reinterpret_cast<u32x4*>(sB + buf * C::B_BYTES + row * C::PSTR + part * 16) = breg[i];
      }
    }
  };

  auto stage_a = [&](int chunk, int abuf) {
    const int g = chunk / cpt, cc = chunk - g * cpt;
    int ch = cc * C::CK;
    const DView S = (ch < P.src[0].C) ? P.src[0] : P.src[1];
    if (ch >= P.src[0].C) ch -= P.src[0].C;
    const int gk = g >> 1, gl = g & 1;
    constexpr int PADP = (TAPS == 9) ? 1 : 0;
    u32x4 areg[C::NAP];
#pragma unroll
    for (int i = 0; i < C::NAP; ++i) {
      const int id = tid + i * 256;
      areg[i] = u32x4{0u, 0u, 0u, 0u};
      if (id < C::HH * C::HW * C::PPP) {
        const int pix = id / C::PPP, part = id % C::PPP;
        const int hy = pix / C::HW, hx = pix - hy * C::HW;
        const int y = (ty0 + hy - PADP) * P.imul + gk - S.oy;
        const int x = (tx0 + hx - PADP) * P.imul + gl - S.ox;
        if (y >= 0 && y < S.H && x >= 0 && x < S.W) {
          const size_t e = ((size_t)(n * S.H + y) * S.W + x) * S.C + ch;
          areg[i] = *reinterpret_cast<const u32x4*>(S.p + e * E::ES + part * 16);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < C::NAP; ++i) {
      const int id = tid + i * 256;
      if (id < C::HH * C::HW * C::PPP) {
        const int pix = id / C::PPP, part = id % C::PPP;
        *reinterpret_cast<u32x4*>(sA + abuf * C::A_BYTES + pix * C::PSTR + part * 16) = areg[i];
      }
    }
  };

  auto compute = [&](int tap, int abuf, int bbuf) {
    const int toff = ((tap / C::R) * C::HW + (tap % C::R)) * C::PSTR;
    const char* pa = sB + bbuf * C::B_BYTES;
    const char* pb = sA + abuf * C::A_BYTES + toff;
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) {
      if constexpr (sizeof(T) == 2) {
        bf16x8 fa[2], fb[C::PXT];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) fa[ct] = *reinterpret_cast<const bf16x8*>(pa + aoff[ct] + kg * 32);
#pragma unroll
        for (int pt = 0; pt < C::PXT; ++pt) fb[pt] = *reinterpret_cast<const bf16x8*>(pb + boff[pt] + kg * 32);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int pt = 0; pt < C::PXT; ++pt)
            acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ct], fb[pt], acc[ct][pt], 0, 0, 0);
      } else {
        f32x4 fa[2], fb[C::PXT];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) fa[ct] = *reinterpret_cast<const f32x4*>(pa + aoff[ct] + kg * 32);
#pragma unroll
        for (int pt = 0; pt < C::PXT; ++pt) fb[pt] = *reinterpret_cast<const f32x4*>(pb + boff[pt] + kg * 32);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int pt = 0; pt < C::PXT; ++pt)
              acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ct][j], fb[pt][j], acc[ct][pt], 0, 0, 0);
      }
    }
  };

  // ---- main loop
  load_b(0, 0);
  int bbuf = 0;
  for (int c = 0; c < nchunks; ++c) {
    const int abuf = (C::A_BUFS == 2) ? (c & 1) : 0;
    if (C::A_BUFS == 1) __syncthreads();   // all waves done with the previous chunk's patch
    stage_a(c, abuf);
#pragma unroll 1
    for (int tap = 0; tap < TAPS; ++tap) {
      store_b(bbuf);
      __syncthreads();
      if (tap + 1 < TAPS) load_b(tap + 1, c);
      else if (c + 1 < nchunks) load_b(0, c + 1);
      compute(tap, abuf, bbuf);
      bbuf ^= 1;
    }
  }

  // ---- epilogue: lane owns pixel (l31) and channels 8g+4h..+3 of each 32-row tile
#pragma unroll
  for (int pt = 0; pt < C::PXT; ++pt) {
    const int m = wpx * (32 * C::PXT) + pt * 32 + l31;
    const int fy = ty0 + (m >> 4), fx = tx0 + (m & 15);
    if (fy >= P.H || fx >= P.W) continue;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        int co = co0 + wco * 64 + ct * 32 + 8 * g + 4 * hh;
        int zz = z;
        if (P.zdiv > 0) { zz = co / P.zdiv; co -= zz * P.zdiv; }
        const int zk = zz >> 1, zl = zz & 1;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = acc[ct][pt][4 * g + j];
        if (P.bias) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += P.bias[co + j];
        }
        if (P.relu) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        const int accq = (co < P.dst_split) ? (P.accumulate & 1) : (P.accumulate & 2);   // per-view accumulate bit
        const DViewW D = (co < P.dst_split) ? P.dst[0] : P.dst[1];
        if (co >= P.dst_split) co -= P.dst_split;
        const int y = fy * P.omul + zk - D.oy, x = fx * P.omul + zl - D.ox;
        if (y < 0 || y >= D.H || x < 0 || x >= D.W) continue;
        T* o = reinterpret_cast<T*>(D.p) + ((size_t)(n * D.H + y) * D.W + x) * D.C + co;
        if constexpr (sizeof(T) == 2) {
          if (accq) {
            bf16x4 old = *reinterpret_cast<const bf16x4*>(o);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += (float)old[j];
          }
          bf16x4 r;
#pragma unroll
          for (int j = 0; j < 4; ++j) r[j] = (bf16_t)v[j];
          *reinterpret_cast<bf16x4*>(o) = r;
        } else {
          if (accq) {
            f32x4 old = *reinterpret_cast<const f32x4*>(o);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += old[j];
          }
          *reinterpret_cast<f32x4*>(o) = f32x4{v[0], v[1], v[2], v[3]};
        }
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------------
// conv3_kernel: the 3x3 specialisation (forward and data gradient).  Same tiling as igemm_kernel, but
//  * the 9 taps are fully unrolled (tap shift = immediate LDS offset, exact counted vmcnt waits),
//  * weight slabs are prefetched TWO steps ahead into two named register sets (an L2/HBM round trip is
//    longer than one 16-MFMA step),
//  * the next chunk's halo patch is prefetched into registers three taps before it is needed,
//  * all staging addresses are per-thread constants (+ a scalar base per step): no index math in the loop,
//  * halo rows are padded to a multiple of 256 B so the two pixel rows of a 32-lane MFMA operand land on
//    disjoint banks (ds_read_b128 conflict-free; the unpadded layout was 2-way).
template <typename T, int BN, int KG>
struct Cfg3 {
  static constexpr int HH = TH + 2, HW = TW + 2;
  static constexpr int CHB = KG * 32, PSTR = CHB + 16, PPP = CHB / 16;
  static constexpr int RS = (HW * PSTR + 255) / 256 * 256;   // halo row stride (bytes)
  static constexpr int A_BYTES = HH * RS;
  static constexpr int B_BYTES = BN * PSTR;
  static constexpr int LDS = A_BYTES + 2 * B_BYTES;
  static constexpr int CK = KG * ET<T>::KGC;
  static constexpr int WCO = BN / 64, WPX = 4 / WCO, PXT = NPIX / (32 * WPX);
  static constexpr int NAP = (HH * HW * PPP + 255) / 256;
  static constexpr int NBP = (BN * PPP + 255) / 256;
};

template <typename T, int BN, int KG>
__global__ __launch_bounds__(256, 2) void conv3_kernel(const IgemmParams P) {
  using C = Cfg3<T, BN, KG>;
  using E = ET<T>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sA = smem;
  char* const sB = smem + C::A_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wco = wave % C::WCO, wpx = wave / C::WCO;
  const int l31 = lane & 31, hh = lane >> 5;

  int logical;
  {
    const int total = gridDim.x, b = blockIdx.x;
    const int xcd = b & 7, slot = b >> 3, q = total >> 3, r = total & 7;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int cot = logical % P.nCo;
  int t = logical / P.nCo;
  const int txi = t % P.tilesX;  t /= P.tilesX;
  const int tyi = t % P.tilesY;
  const int n = t / P.tilesY;
  const int ty0 = tyi * TH, tx0 = txi * TW;
  const int co0 = cot * BN;

  f32x16 acc[2][C::PXT];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < C::PXT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  int aoff[2], boff[C::PXT];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) aoff[ct] = (wco * 64 + ct * 32 + l31) * C::PSTR + hh * 16;
#pragma unroll
  for (int pt = 0; pt < C::PXT; ++pt) {
    const int m = wpx * (32 * C::PXT) + pt * 32 + l31;
    boff[pt] = (m >> 4) * C::RS + (m & 15) * C::PSTR + hh * 16;
  }

  // ---- per-thread staging descriptors (constant for the whole block).  Loads are buffer loads: a
  // wave-uniform resource (SGPRs) + per-thread constant voffset + per-step scalar soffset, so the loop
  // carries no address VALU; out-of-image halo pixels use an out-of-range voffset and read as zero.
  constexpr unsigned OOB = 0xFFFFFFF0u;
  constexpr bool A_EXACT = (C::HH * C::HW * C::PPP) % 256 == 0;
  constexpr bool B_EXACT = (BN * C::PPP) % 256 == 0;
  int a_lds[C::NAP];
  unsigned a_g[2][C::NAP];
#pragma unroll
  for (int i = 0; i < C::NAP; ++i) {
    const int id = tid + i * 256;
    a_lds[i] = -1;
    a_g[0][i] = a_g[1][i] = OOB;
    if (id < C::HH * C::HW * C::PPP) {
      const int pix = id / C::PPP, part = id % C::PPP;
      const int hy = pix / C::HW, hx = pix - hy * C::HW;
      a_lds[i] = hy * C::RS + hx * C::PSTR + part * 16;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const DView S = P.src[k];
        const int y = ty0 + hy - 1 - S.oy, x = tx0 + hx - 1 - S.ox;
        if (S.C > 0 && y >= 0 && y < S.H && x >= 0 && x < S.W)
          a_g[k][i] = (unsigned)(((y * S.W + x) * S.C) * E::ES + part * 16);
      }
    }
  }
  int b_lds[C::NBP];
  unsigned b_g[C::NBP];
#pragma unroll
  for (int i = 0; i < C::NBP; ++i) {
    const int id = tid + i * 256;
    const int row = id / C::PPP, part = id % C::PPP;
    const bool ok = B_EXACT || id < BN * C::PPP;
    b_lds[i] = ok ? row * C::PSTR + part * 16 : -1;
    b_g[i] = ok ? (unsigned)(((co0 + row) * P.wK) * E::ES + part * 16) : OOB;
  }

  const int nchunks = P.Ctot / C::CK;
  const unsigned w_tap_stride = (unsigned)P.Cout * P.wK * E::ES;
  const __amdgpu_buffer_rsrc_t w_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)P.w, (short)0, (int)(9u * w_tap_stride), 0x00020000);
  __amdgpu_buffer_rsrc_t a_rsrc[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const DView S = P.src[k];
    const unsigned img = (unsigned)S.H * S.W * S.C * E::ES;
    a_rsrc[k] = __builtin_amdgcn_make_buffer_rsrc((void*)(S.p + (size_t)n * img), (short)0, (int)img, 0x00020000);
  }

  u32x4 breg[2][C::NBP];
  u32x4 areg[C::NAP];

  auto load_b = [&](u32x4 (&dst)[C::NBP], int chunk, int tap) {
    if (tap >= 9) { tap -= 9; chunk += 1; }
    chunk = chunk < nchunks ? chunk : nchunks - 1;        // past the end: harmless re-load, never consumed
    const unsigned soff = (unsigned)tap * w_tap_stride + (unsigned)chunk * (C::CK * E::ES);
#pragma unroll
    for (int i = 0; i < C::NBP; ++i)
      dst[i] = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, b_g[i], soff, 0);
  };
  auto store_b = [&](const u32x4 (&src)[C::NBP], int buf) {
#pragma unroll
    for (int i = 0; i < C::NBP; ++i)
      if (B_EXACT || b_lds[i] >= 0) *reinterpret_cast<u32x4*>(sB + buf * C::B_BYTES + b_lds[i]) = src[i];
  };
  auto load_a = [&](int chunk) {
    chunk = chunk < nchunks ? chunk : nchunks - 1;
    const int ch = chunk * C::CK;
    if (ch < P.src[0].C) {
      const unsigned soff = (unsigned)ch * E::ES;
#pragma unroll
      for (int i = 0; i < C::NAP; ++i) areg[i] = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc[0], a_g[0][i], soff, 0);
    } else {
      const unsigned soff = (unsigned)(ch - P.src[0].C) * E::ES;
#pragma unroll
      for (int i = 0; i < C::NAP; ++i) areg[i] = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc[1], a_g[1][i], soff, 0);
    }
  };
  auto store_a = [&]() {
#pragma unroll
    for (int i = 0; i < C::NAP; ++i)
      if (A_EXACT || i + 1 < C::NAP || a_lds[i] >= 0) *reinterpret_cast<u32x4*>(sA + a_lds[i]) = areg[i];
  };

  auto compute = [&](int toff, int bbuf) {
    const char* pa = sB + bbuf * C::B_BYTES;
    const char* pb = sA + toff;
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) {
      if constexpr (sizeof(T) == 2) {
        bf16x8 fa[2], fb[C::PXT];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) fa[ct] = *reinterpret_cast<const bf16x8*>(pa + aoff[ct] + kg * 32);
#pragma unroll
        for (int pt = 0; pt < C::PXT; ++pt) fb[pt] = *reinterpret_cast<const bf16x8*>(pb + boff[pt] + kg * 32);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int pt = 0; pt < C::PXT; ++pt)
            acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ct], fb[pt], acc[ct][pt], 0, 0, 0);
      } else {
        f32x4 fa[2], fb[C::PXT];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) fa[ct] = *reinterpret_cast<const f32x4*>(pa + aoff[ct] + kg * 32);
#pragma unroll
        for (int pt = 0; pt < C::PXT; ++pt) fb[pt] = *reinterpret_cast<const f32x4*>(pb + boff[pt] + kg * 32);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int pt = 0; pt < C::PXT; ++pt)
              acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ct][j], fb[pt][j], acc[ct][pt], 0, 0, 0);
      }
    }
  };

  // one chunk = 9 fully unrolled tap steps; PAR = parity of its first step (selects register set / LDS slot).
  // Step t:  barrier(t) | park slab t+1 in the other LDS slot (its last readers passed barrier(t)) | issue the
  // loads of slab t+3 into the registers just freed | 16 MFMAs on slab t.  The LDS writes of a slab are a
  // whole step old when the barrier that publishes them arrives, so a barrier only ever waits for skew.
  auto chunk_body = [&](int c, auto par_tag) {
    constexpr int PAR = decltype(par_tag)::value;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int set = (PAR + tap) & 1;              // slot / register set of THIS step's slab
      __syncthreads();
      if (tap + 1 < 9 || c + 1 < nchunks) store_b(breg[set ^ 1], set ^ 1);          // slab t+1
      if (tap + 3 < 9 || c + 1 < nchunks) load_b(breg[set ^ 1], c, tap + 3);        // slab t+3 (uniform branch)
      if (tap == 5 && c + 1 < nchunks) load_a(c + 1);
      compute((tap / 3) * C::RS + (tap % 3) * C::PSTR, set);
    }
    if (c + 1 < nchunks) {
      __syncthreads();        // every wave is done with this chunk's patch
      store_a();
    }
  };

  load_a(0);
  load_b(breg[0], 0, 0);
  load_b(breg[1], 0, 1);
  store_a();
  store_b(breg[0], 0);
  load_b(breg[0], 0, 2);
  int c = 0;
  for (; c + 1 < nchunks; c += 2) {
    chunk_body(c, std::integral_constant<int, 0>{});
    chunk_body(c + 1, std::integral_constant<int, 1>{});
  }
  if (c < nchunks) chunk_body(c, std::integral_constant<int, 0>{});

  // ---- epilogue (identical to igemm_kernel, omul = 1, no bias)
#pragma unroll
  for (int pt = 0; pt < C::PXT; ++pt) {
    const int m = wpx * (32 * C::PXT) + pt * 32 + l31;
    const int fy = ty0 + (m >> 4), fx = tx0 + (m & 15);
    if (fy >= P.H || fx >= P.W) continue;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        int co = co0 + wco * 64 + ct * 32 + 8 * g + 4 * hh;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = acc[ct][pt][4 * g + j];
        if (P.bias) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += P.bias[co + j];
        }
        if (P.relu) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        const int accq = (co < P.dst_split) ? (P.accumulate & 1) : (P.accumulate & 2);   // per-view accumulate bit
        const DViewW D = (co < P.dst_split) ? P.dst[0] : P.dst[1];
        if (co >= P.dst_split) co -= P.dst_split;
        const int y = fy - D.oy, x = fx - D.ox;
        if (y < 0 || y >= D.H || x < 0 || x >= D.W) continue;
        T* o = reinterpret_cast<T*>(D.p) + ((size_t)(n * D.H + y) * D.W + x) * D.C + co;
        if constexpr (sizeof(T) == 2) {
          if (accq) {
            bf16x4 old = *reinterpret_cast<const bf16x4*>(o);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += (float)old[j];
          }
          bf16x4 r;
#pragma unroll
          for (int j = 0; j < 4; ++j) r[j] = (bf16_t)v[j];
          *reinterpret_cast<bf16x4*>(o) = r;
        } else {
          if (accq) {
            f32x4 old = *reinterpret_cast<const f32x4*>(o);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += old[j];
          }
          *reinterpret_cast<f32x4*>(o) = f32x4{v[0], v[1], v[2], v[3]};
        }
      }
    }
  }
}

// ---- conv3m16_kernel: conv3_kernel on v_mfma_f32_16x16x32_bf16 (4x4 tiles of 16x16 per wave).  Same bytes
// and MFMA cycles; the chip sustains a higher clock on this shape (MI355X_MICROARCH, DVFS give-back item 7).
template <typename T, int BN, int KG>
struct Cfg3M {
  static constexpr int HH = TH + 2, HW = TW + 2;
  static constexpr int CHB = KG * 32, PSTR = CHB + 32, PPP = CHB / 16;   // +32 B: conflict-free 16x16x32 fragments
  static constexpr int RS = HW * PSTR;                       // a 16-pixel operand never straddles halo rows
  static constexpr int A_BYTES = HH * RS;
  static constexpr int B_BYTES = BN * PSTR;
  static constexpr int LDS = A_BYTES + 2 * B_BYTES;
  static constexpr int CK = KG * ET<T>::KGC;
  static constexpr int WCO = BN / 64, WPX = 4 / WCO, PXT = NPIX / (32 * WPX);
  static constexpr int NAP = (HH * HW * PPP + 255) / 256;
  static constexpr int NBP = (BN * PPP + 255) / 256;
};

template <typename T, int BN, int KG>
__global__ __launch_bounds__(256, 2) void conv3m16_kernel(const IgemmParams P) {
  using C = Cfg3M<T, BN, KG>;
  using E = ET<T>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sA = smem;
  char* const sB = smem + C::A_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wco = wave % C::WCO, wpx = wave / C::WCO;
  
  int logical;
  {
    const int total = gridDim.x, b = blockIdx.x;
    const int xcd = b & 7, slot = b >> 3, q = total >> 3, r = total & 7;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int cot = logical % P.nCo;
  int t = logical / P.nCo;
  const int txi = t % P.tilesX;  t /= P.tilesX;
  const int tyi = t % P.tilesY;
  const int n = t / P.tilesY;
  const int ty0 = tyi * TH, tx0 = txi * TW;
  const int co0 = cot * BN;

  constexpr int PT16 = 2 * C::PXT;              // 16-pixel operand tiles per wave (each = one tile row)
  const int l15 = lane & 15, kb = lane >> 4;
  f32x4 acc[4][PT16];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < PT16; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;

  int aoff[4], boff[PT16];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) aoff[ct] = (wco * 64 + ct * 16 + l15) * C::PSTR + kb * 16;
#pragma unroll
  for (int pt = 0; pt < PT16; ++pt) boff[pt] = (wpx * PT16 + pt) * C::RS + l15 * C::PSTR + kb * 16;

  // ---- per-thread staging descriptors (constant for the whole block).  Loads are buffer loads: a
  // wave-uniform resource (SGPRs) + per-thread constant voffset + per-step scalar soffset, so the loop
  // carries no address VALU; out-of-image halo pixels use an out-of-range voffset and read as zero.
  constexpr unsigned OOB = 0xFFFFFFF0u;
  constexpr bool A_EXACT = (C::HH * C::HW * C::PPP) % 256 == 0;
  constexpr bool B_EXACT = (BN * C::PPP) % 256 == 0;
  int a_lds[C::NAP];
  unsigned a_g[2][C::NAP];
#pragma unroll
  for (int i = 0; i < C::NAP; ++i) {
    const int id = tid + i * 256;
    a_lds[i] = -1;
    a_g[0][i] = a_g[1][i] = OOB;
    if (id < C::HH * C::HW * C::PPP) {
      const int pix = id / C::PPP, part = id % C::PPP;
      const int hy = pix / C::HW, hx = pix - hy * C::HW;
      a_lds[i] = hy * C::RS + hx * C::PSTR + part * 16;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const DView S = P.src[k];
        const int y = ty0 + hy - 1 - S.oy, x = tx0 + hx - 1 - S.ox;
        if (S.C > 0 && y >= 0 && y < S.H && x >= 0 && x < S.W)
          a_g[k][i] = (unsigned)(((y * S.W + x) * S.C) * E::ES + part * 16);
      }
    }
  }
  int b_lds[C::NBP];
  unsigned b_g[C::NBP];
#pragma unroll
  for (int i = 0; i < C::NBP; ++i) {
    const int id = tid + i * 256;
    const int row = id / C::PPP, part = id % C::PPP;
    const bool ok = B_EXACT || id < BN * C::PPP;
    b_lds[i] = ok ? row * C::PSTR + part * 16 : -1;
    b_g[i] = ok ? (unsigned)(((co0 + row) * P.wK) * E::ES + part * 16) : OOB;
  }

  const int nchunks = P.Ctot / C::CK;
  const unsigned w_tap_stride = (unsigned)P.Cout * P.wK * E::ES;
  const __amdgpu_buffer_rsrc_t w_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)P.w, (short)0, (int)(9u * w_tap_stride), 0x00020000);
  __amdgpu_buffer_rsrc_t a_rsrc[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const DView S = P.src[k];
    const unsigned img = (unsigned)S.H * S.W * S.C * E::ES;
    a_rsrc[k] = __builtin_amdgcn_make_buffer_rsrc((void*)(S.p + (size_t)n * img), (short)0, (int)img, 0x00020000);
  }

  u32x4 breg[2][C::NBP];
  u32x4 areg[C::NAP];

  auto load_b = [&](u32x4 (&dst)[C::NBP], int chunk, int tap) {
    if (tap >= 9) { tap -= 9; chunk += 1; }
    chunk = chunk < nchunks ? chunk : nchunks - 1;        // past the end: harmless re-load, never consumed
    const unsigned soff = (unsigned)tap * w_tap_stride + (unsigned)chunk * (C::CK * E::ES);
#pragma unroll
    for (int i = 0; i < C::NBP; ++i)
      dst[i] = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, b_g[i], soff, 0);
  };
  auto store_b = [&](const u32x4 (&src)[C::NBP], int buf) {
#pragma unroll
    for (int i = 0; i < C::NBP; ++i)
      if (B_EXACT || b_lds[i] >= 0) *reinterpret_cast<u32x4*>(sB + buf * C::B_BYTES + b_lds[i]) = src[i];
  };
  auto load_a = [&](int chunk) {
    chunk = chunk < nchunks ? chunk : nchunks - 1;
    const int ch = chunk * C::CK;
    if (ch < P.src[0].C) {
      const unsigned soff = (unsigned)ch * E::ES;
#pragma unroll
      for (int i = 0; i < C::NAP; ++i) areg[i] = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc[0], a_g[0][i], soff, 0);
    } else {
      const unsigned soff = (unsigned)(ch - P.src[0].C) * E::ES;
#pragma unroll
      for (int i = 0; i < C::NAP; ++i) areg[i] = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc[1], a_g[1][i], soff, 0);
    }
  };
  auto store_a = [&]() {
#pragma unroll
    for (int i = 0; i < C::NAP; ++i)
      if (A_EXACT || i + 1 < C::NAP || a_lds[i] >= 0) *reinterpret_cast<u32x4*>(sA + a_lds[i]) = areg[i];
  };

  auto compute = [&](int toff, int bbuf) {
    const char* pa = sB + bbuf * C::B_BYTES;
    const char* pb = sA + toff;
#pragma unroll
    for (int ks = 0; ks < KG / 2; ++ks) {       // k steps of 32 channels
      bf16x8 fa[4], fb[PT16];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) fa[ct] = *reinterpret_cast<const bf16x8*>(pa + aoff[ct] + ks * 64);
#pragma unroll
      for (int pt = 0; pt < PT16; ++pt) fb[pt] = *reinterpret_cast<const bf16x8*>(pb + boff[pt] + ks * 64);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int pt = 0; pt < PT16; ++pt)
          acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ct], fb[pt], acc[ct][pt], 0, 0, 0);
    }
  };

  // one chunk = 9 fully unrolled tap steps; PAR = parity of its first step (selects register set / LDS slot)
  auto chunk_body = [&](int c, auto par_tag) {
    constexpr int PAR = decltype(par_tag)::value;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int set = (PAR + tap) & 1;
      store_b(breg[set], set);
      __syncthreads();
      if (tap + 2 < 9 || c + 1 < nchunks) load_b(breg[set], c, tap + 2);   // uniform branch; no loads past the end
      if (tap == 6 && c + 1 < nchunks) load_a(c + 1);
      compute((tap / 3) * C::RS + (tap % 3) * C::PSTR, set);
    }
    if (c + 1 < nchunks) {
      __syncthreads();        // every wave is done with this chunk's patch
      store_a();
    }
  };

  load_a(0);
  load_b(breg[0], 0, 0);
  load_b(breg[1], 0, 1);
  store_a();
  int c = 0;
  for (; c + 1 < nchunks; c += 2) {
    chunk_body(c, std::integral_constant<int, 0>{});
    chunk_body(c + 1, std::integral_constant<int, 1>{});
  }
  if (c < nchunks) chunk_body(c, std::integral_constant<int, 0>{});

  // ---- epilogue: D of 16x16x32: col = lane&15 (pixel), rows (lane>>4)*4 + reg (4 consecutive channels)
  float bs[4][4], bq[4][4];            // BatchNorm partials of this lane: [ct][reg] over its pixels
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int j = 0; j < 4; ++j) { bs[ct][j] = 0.f; bq[ct][j] = 0.f; }
#pragma unroll
  for (int pt = 0; pt < PT16; ++pt) {
    const int fy = ty0 + wpx * PT16 + pt, fx = tx0 + l15;
    if (fy >= P.H || fx >= P.W) continue;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      int co = co0 + wco * 64 + ct * 16 + kb * 4;
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = acc[ct][pt][j];
      if (P.bias) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += P.bias[co + j];
      }
      if (P.relu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
      }
      const int accq = (co < P.dst_split) ? (P.accumulate & 1) : (P.accumulate & 2);   // per-view accumulate bit
        const DViewW D = (co < P.dst_split) ? P.dst[0] : P.dst[1];
      if (co >= P.dst_split) co -= P.dst_split;
      const int y = fy - D.oy, x = fx - D.ox;
      if (y < 0 || y >= D.H || x < 0 || x >= D.W) continue;
      T* o = reinterpret_cast<T*>(D.p) + ((size_t)(n * D.H + y) * D.W + x) * D.C + co;
      if (accq) {
        bf16x4 old = *reinterpret_cast<const bf16x4*>(o);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += (float)old[j];
      }
      bf16x4 r;
#pragma unroll
      for (int j = 0; j < 4; ++j) r[j] = (bf16_t)v[j];
      *reinterpret_cast<bf16x4*>(o) = r;
#pragma unroll
      for (int j = 0; j < 4; ++j) {              // statistics of the value as STORED (bf16-rounded)
        const float q = (float)r[j];
        bs[ct][j] += q;
        bq[ct][j] = fmaf(q, q, bq[ct][j]);
      }
    }
  }
  if (P.stats) {
    // wavefront reduction over the 16 pixel lanes of each channel group, then the two pixel-waves through LDS
#pragma unroll
    for (int m = 1; m < 16; m <<= 1)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          bs[ct][j] += __shfl_xor(bs[ct][j], m);
          bq[ct][j] += __shfl_xor(bq[ct][j], m);
        }
    __syncthreads();                               // all MFMA operand reads of the tile are done: reuse LDS
    float* red = reinterpret_cast<float*>(smem);   // [WPX][2][BN]
    if (l15 == 0) {
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int cl = wco * 64 + ct * 16 + kb * 4 + j;
          red[(wpx * 2 + 0) * BN + cl] = bs[ct][j];
          red[(wpx * 2 + 1) * BN + cl] = bq[ct][j];
        }
    }
    __syncthreads();
    const int part = (n * P.tilesY + tyi) * P.tilesX + txi;
    for (int i = tid; i < 2 * BN; i += 256) {
      const int q = i / BN, cl = i - q * BN;
      float t = 0.f;
#pragma unroll
      for (int wp = 0; wp < C::WPX; ++wp) t += red[(wp * 2 + q) * BN + cl];
      P.stats[((size_t)part * 2 + q) * P.Cout + co0 + cl] = t;
    }
  }
}

template <typename T, int BN, int KG>
int32_t launch3(const IgemmParams& Pin, int kclass, hipStream_t s, int* stat_parts) {
  using C = Cfg3<T, BN, KG>;
  IgemmParams P = Pin;
  auto kern = conv3_kernel<T, BN, KG>;
  unet_set_max_lds(reinterpret_cast<const void*>(kern), C::LDS);
  const long long blocks = (long long)P.N * P.tilesY * P.tilesX * P.nCo;
  UNET_REQUIRE(blocks > 0 && blocks < (1LL << 31), UNET_ERR_UNSUPPORTED, "conv3: grid of %lld blocks", blocks);
  const double flops = 2.0 * P.N * P.H * P.W * (double)P.Cout * P.Ctot * 9;
  if constexpr (sizeof(T) == 2 && BN == 128 && KG == 4) {
    // default: the 16x16x32 MFMA variant (up to 7 % faster in interleaved A/B runs: the chip holds a
    // higher clock on that shape); UNET_CONV_VAR=0 selects the 32x32x16 kernel (tuning hook)
    if (unet_tuning().conv_var != '0') {
      using CM = Cfg3M<T, BN, KG>;
      auto km = conv3m16_kernel<T, BN, KG>;
      unet_set_max_lds(reinterpret_cast<const void*>(km), CM::LDS);
      if (P.stats && stat_parts) *stat_parts = P.N * P.tilesY * P.tilesX;   // epilogue writes the BN partials
      ProfScope prof(kclass, flops, s, "conv3m16_kernel");
      hipLaunchKernelGGL(km, dim3((unsigned)blocks), dim3(256), CM::LDS, s, P);
      return unet_check_launch("conv3m16_kernel");
    }
  }
  P.stats = nullptr;
  ProfScope prof(kclass, flops, s, "conv3_kernel");
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), C::LDS, s, P);
  return unet_check_launch("conv3_kernel");
}

// ------------------------------------------------------------------------------------------------------
// conv3_pdma_kernel<BN>: the 3x3 convolution (forward and data gradient) of every 16-aligned bf16 layer with at least
// 128 input channels, BOTH operands staged by LDS-DMA, persistent blocks.
//   One 512-thread block per CU owns 16x16 output pixels x BN (128 or 64) output channels at a time (8 waves x
//   (BN/2 co x 64 px) on v_mfma_f32_16x16x32_bf16).  Compared with conv3m16_kernel (two 256-thread blocks per CU,
//   each staging its own weight slab through registers + ds_write_b128) the weight slab of a tap is fetched ONCE per
//   CU and written to LDS by the DMA engine: half the L2 traffic per MFMA and no ds_write_b128 (79 B/clk) competing
//   with the fragment reads for the LDS array, which is what held the old kernel at 46 % MFMA busy.
//   LDS: two halo'd 18x18 pixel patches of one 64-channel chunk (rows padded to 160 B: conflict-free 16x16x32
//   fragments) + a 3-slot ring of BN x 64 weight slabs (unpadded 128-B rows, 16-B pieces XOR-swizzled by
//   (row>>1)&7 through the DMA's per-lane SOURCE address).  Weight slabs are issued two taps ahead, the next
//   chunk's patch is issued one DMA per wave per tap during the current chunk; every wait is a counted vmcnt.
//   PERSISTENT: a block walks a list of (output-channel tile, pixel tile) work items; the DMA stream (patch of the
//   next chunk, weight slabs two taps ahead) simply continues into the next work item, so a block's un-overlapped
//   prologue is paid once per launch instead of once per tile, and the epilogue's stores drain behind the next
//   tile's MFMAs.  That is what the 128-input-channel layers (2 chunks = 18 steps per tile) needed.
// Work order: consecutive work items = consecutive pixel tiles of ONE channel tile, and XCD x owns a contiguous
// run of them, so the blocks of an XCD stream the same weight slabs and neighbouring halos through its L2.
// sum over the 16 lanes of a DPP row, result in every lane: 4 VALU adds with DPP operands (quad xor 1, quad xor 2,
// half-row mirror, row mirror) instead of 4 ds_bpermute + 4 adds; fixed order -> deterministic
// sum over the 16 lanes of a DPP row, for N >= 3 independent values at once: v_add_f32 with a DPP source operand, one
// instruction per value and step (quad xor 1, quad xor 2, half-row mirror, row mirror; fixed order -> deterministic).
// Through __builtin_amdgcn_update_dpp hipcc emitted v_mov_b32 (old = 0) + v_mov_b32_dpp + half a v_pk_add_f32 per step
// (its packed-add vectoriser defeats the DPP combine): 2.4x the instructions.  Step-major order + `asm volatile` (kept in
// source order) puts N - 1 >= 2 instructions between the VALU write of a value and the DPP read of it -- the wait states
// hipcc does not pad inside asm; one s_nop covers the producers of the inputs.  dst = dpp(src) + src: the same sums, bit
// for bit.
template <int N>
__device__ __forceinline__ void row16_sum_n(float (&v)[N]) {
  static_assert(N >= 3, "hazard distance");
  asm volatile("s_nop 1");
#pragma unroll
  for (int i = 0; i < N; ++i)
    asm volatile("v_add_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(v[i]) : "0"(v[i]));
#pragma unroll
  for (int i = 0; i < N; ++i)
    asm volatile("v_add_f32_dpp %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=v"(v[i]) : "0"(v[i]));
#pragma unroll
  for (int i = 0; i < N; ++i)
    asm volatile("v_add_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf" : "=v"(v[i]) : "0"(v[i]));
#pragma unroll
  for (int i = 0; i < N; ++i)
    asm volatile("v_add_f32_dpp %0, %1, %1 row_mirror row_mask:0xf bank_mask:0xf" : "=v"(v[i]) : "0"(v[i]));
}

// Work order of conv3_pdma: item wk -> (channel tile, pixel tile).  XCD x owns 32 consecutive items per round; with
// co_il = c those are 32 / c pixel tiles x c channel tiles (super-groups of c channel tiles are walked tile-major), so the
// c blocks that read the SAME input patches run on one L2 at the same time and the patch leaves the Infinity Cache / HBM
// once per super-group instead of once per channel tile (c = 1: channel-tile-major, every channel tile re-streams X).
__device__ __forceinline__ void pdma_item(int wk, int n_tiles, int c, int& cot, int& tile) {
  const int span = c * n_tiles;
  const int sg = wk / span, rem = wk - sg * span;
  tile = rem / c;
  cot = sg * c + (rem - tile * c);
}

template <int BN, bool PAIR = false, bool ROW3 = false>
struct CfgP {
  static constexpr int TH = 16, TW = 16, HH = 18, HW = 18;
  static constexpr int PSTR = 160, PPP = 10, RS = HW * PSTR;
  static constexpr int A_INSTR = (HH * HW * PPP + 63) / 64;         // 51 wave-instructions of 1 KiB
  static constexpr int A_BYTES = A_INSTR * 1024;
  static constexpr int NDA = (A_INSTR + 7) / 8;                      // 7 per wave
  // PAIR (BN = 64): a ring slot holds the slabs of TWO consecutive taps (a step = two taps between barriers)
  // ROW3 (BN = 64): a slot holds the three slabs of a tap ROW, and the ring has two slots (the next step's slabs are fetched
  // during the current step)
  static constexpr int W_BYTES = (ROW3 ? 3 : (PAIR ? 2 : 1)) * BN * 128, NDW = W_BYTES / 1024 / 8; // 2 (BN 128, PAIR), 3 (ROW3) or 1 (BN 64) per wave
  static constexpr int NSLOT = ROW3 ? 2 : 3;
  // the weight ring sits FIRST: slot * W_BYTES (<= 32 KiB) then folds into the 16-bit offset field of the fragment
  // ds_reads (behind the patches, at 102 KiB, every read cost a v_add and the tap a spilled-SGPR v_readlane)
  static constexpr int W_BASE = 0;
  static constexpr int A_BASE = NSLOT * W_BYTES;
  static constexpr int RED_BASE = A_BASE + 2 * A_BYTES;              // BatchNorm partials of the epilogue
  static constexpr int RED_BYTES = 4 * 2 * BN * 4;
  static constexpr int DUMMY = RED_BASE + RED_BYTES;
  static constexpr int LDS = DUMMY + 1024;
  static constexpr int CT = BN / 32;                                 // 16-channel MFMA tiles per wave (2 waves along channels)
  static constexpr int NST = CT / 2 * 4;                             // 16-byte output stores per lane per work item
};

// PP ("ping-pong"): the two waves of a SIMD (w, w + 4) run HALF A STEP apart.  Waves 0-3 own the tile's first BN/2
// output channels, waves 4-7 the second; a step is [LOAD: 16 fragment reads of the tap, this wave's LDS-DMA issues, the
// counted vmcnt, lgkmcnt(0)] s_barrier [COMPUTE: the tap's 32 MFMAs straight from registers] s_barrier, and waves 4-7
// start one barrier late -- while one wave of a SIMD feeds the matrix pipe its partner reads LDS and issues DMAs,
// instead of all eight bursting their DMAs and fragment reads together behind one barrier per tap (stamps: 35-45 % of a
// lock-step tap went to the DMA issue burst, profiles/r02_pdma_stamps.txt).  LDS hazards at distance one barrier: a
// slab / patch buffer is re-filled by DMAs issued in the slot after its last reads, which are retired (lgkmcnt(0))
// BEFORE the barrier that ends their LOAD.
#ifndef PDMA_DEFER128
#define PDMA_DEFER128 (!PP && !BNBWD)
#endif
// PAIR (round 4; BN = 64, exactly two 64-channel chunks = the 128 -> 64 layers at 256 x 256 and the 128 -> 64 data
// gradient): with 64-channel tiles a tap is only 16 MFMAs per wave, and the stamps (profiles/r04_pdma64_stamps.txt) put
// ~800 of its 1 300 cycles into what a tap costs regardless of its size -- the barrier skew, the DMA-issue burst, the
// fragment-read latency in front of the first MFMA.  A STEP is therefore two consecutive taps of the 18 of a work item
// (9 steps; step 4 straddles the chunks): one barrier, one counted wait and one DMA burst per 32 MFMAs, as in the
// 128-channel kernel; a ring slot holds both taps' weight slabs (16 KiB, the 128-channel ring), the second tap's
// fragments are fetched behind the first tap's MFMAs.  Same accumulation order, bit-identical outputs.
// ROW3 (BN = 64, any number of chunks): a step = the three taps of one tap ROW (48 MFMAs per wave between barriers, three
// steps per chunk); a two-slot weight ring, every wave issues its DMAs in front of its MFMAs (with this much work per step
// the placement of the burst no longer matters: UNET_PDMA_STG 0 / 1 are +-0 on the pair kernel).
template <int BN, bool BNBWD = false, bool PP = false, bool PAIR = false, bool ROW3 = false>
__device__ __forceinline__ void conv3_pdma_body(const IgemmParams& P) {
  static_assert(!PAIR || (BN == 64 && !PP), "pair steps: the lock-step 64-channel kernel");
  static_assert(!ROW3 || (BN == 64 && !PP && !PAIR), "row steps: the lock-step 64-channel kernel");
  using C = CfgP<BN, PAIR, ROW3>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void lds_void;
  constexpr unsigned OOB = 0xFFFFFFF0u;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave >> 2;                     // PP: 0 = the leading half, 1 = one barrier behind
  const int wco = PP ? grp : (wave & 1), wpx = PP ? (wave & 3) : (wave >> 1);
  const int l15 = lane & 15, kb = lane >> 4;

  const int G = gridDim.x;                       // launch_pdma makes it a multiple of 8
  const int tiles_img = P.tilesY * P.tilesX;
  const int n_tiles = P.N * tiles_img;
  const int total = n_tiles * P.nCo;
  // XCD x (= blockIdx % 8) owns the contiguous logical range [x*G/8, (x+1)*G/8)
  const int logical = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  if (logical >= total) return;

  int aoff[C::CT][2], boff[4];
#pragma unroll
  for (int ct = 0; ct < C::CT; ++ct) {
    const int row = wco * (BN / 2) + ct * 16 + l15;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) aoff[ct][ks] = row * 128 + (((ks * 4 + kb) ^ ((row >> 1) & 7)) << 4);
  }
#pragma unroll
  for (int pt = 0; pt < 4; ++pt) boff[pt] = (wpx * 4 + pt) * C::RS + l15 * C::PSTR + kb * 16;

  // lane geometry of this wave's patch DMA pieces (constant), per-work offsets (a_g) derived from it
  int a_code[C::NDA];                            // hy | hx << 8 | part << 16, -1 = pad piece
#pragma unroll
  for (int j = 0; j < C::NDA; ++j) {
    const int q = (j * 8 + wave) * 64 + lane;
    const int pix = q / C::PPP, part = q - pix * C::PPP;
    const int hy = pix / C::HW, hx = pix - hy * C::HW;
    // bits 24-27: the pixel lies in the patch's top / bottom row, left / right column (the halo of a frame-edge tile)
    const int edge = (hy == 0) | ((hy == C::HH - 1) << 1) | ((hx == 0) << 2) | ((hx == C::HW - 1) << 3);
    a_code[j] = (pix < C::HH * C::HW && part < 8) ? (hy | (hx << 8) | (part << 16) | (edge << 24)) : -1;
  }
  unsigned w_g[C::NDW];                          // (PAIR: both instructions of a step use w_g[0], rows 0-63 of a tap's slab)
#pragma unroll
  for (int j = 0; j < C::NDW; ++j) {
    const int q = (j * 8 + wave) * 64 + lane;
    const int row = q >> 3, pos = q & 7;
    w_g[j] = (unsigned)((row * P.wK) * 2 + ((pos ^ ((row >> 1) & 7)) << 4));
  }

  const int nchunks = P.Ctot / 64;
  const unsigned w_tap_stride = (unsigned)P.Cout * P.wK * 2;
  const __amdgpu_buffer_rsrc_t w_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)P.w, (short)0, (int)(9u * w_tap_stride), 0x00020000);
  const unsigned img0 = (unsigned)P.src[0].H * P.src[0].W * P.src[0].C * 2;
  const unsigned img1 = (unsigned)P.src[1].H * P.src[1].W * P.src[1].C * 2;

  // ---- DMA-side state: the work item whose patches / weights are being fetched
  unsigned a_g[2][C::NDA];
  __amdgpu_buffer_rsrc_t a_rsrc[2];
  unsigned d_wbase = 0;                          // byte offset of the work item's first weight row
  bool d_live = true;
  // Dense sources (P.pdma_dense_src: both views frame-sized at offset 0, one channel stride): a lane's patch offsets
  // relative to the patch origin never change -- kept in a_g[1][], which the general path uses for the second view -- and
  // the work item enters through the descriptors' base addresses; per item only the halo of a frame-edge tile is masked
  // (3 vector instructions per piece instead of ~17 x 2 views, in the last chunk's taps where issue slots are scarce).
  const bool dsrc = P.pdma_dense_src != 0;
  if (dsrc) {
#pragma unroll
    for (int j = 0; j < C::NDA; ++j) {
      const int code = a_code[j];
      const int hy = code & 255, hx = (code >> 8) & 255, part = (code >> 16) & 255;
      a_g[1][j] = code >= 0 ? (unsigned)(((hy * P.src[0].W + hx) * P.src[0].C) * 2 + part * 16) : OOB;
    }
  }
  auto setup_dma = [&](int wk) {
    int cot, tile;
    pdma_item(wk, n_tiles, P.co_il, cot, tile);
    const int n = tile / tiles_img, r = tile - n * tiles_img;
    const int ty0 = (r / P.tilesX) * C::TH, tx0 = (r % P.tilesX) * C::TW;
    d_wbase = (unsigned)(cot * BN) * P.wK * 2;
    if (dsrc) {
      const unsigned E = (unsigned)((ty0 == 0) | ((ty0 + C::TH == P.H) << 1) | ((tx0 == 0) << 2) | ((tx0 + C::TW == P.W) << 3)) << 24;
      // (the patch origin of a top / left tile lies in front of the image: only in-frame lanes carry an in-range offset)
      const long long tb = ((long long)(ty0 - 1) * P.src[0].W + (tx0 - 1)) * (P.src[0].C * 2);
      a_rsrc[0] = __builtin_amdgcn_make_buffer_rsrc((void*)(P.src[0].p + (long long)n * img0 + tb), (short)0, 0x7FFFFFF0, 0x00020000);
      a_rsrc[1] = __builtin_amdgcn_make_buffer_rsrc((void*)(P.src[1].p ? P.src[1].p + (long long)n * img1 + tb : P.src[0].p),
                                                    (short)0, P.src[1].p ? 0x7FFFFFF0 : 0, 0x00020000);
#pragma unroll
      for (int j = 0; j < C::NDA; ++j) a_g[0][j] = ((unsigned)a_code[j] & E) ? OOB : a_g[1][j];
      return;
    }
    a_rsrc[0] = __builtin_amdgcn_make_buffer_rsrc((void*)(P.src[0].p + (size_t)n * img0), (short)0, (int)img0, 0x00020000);
    a_rsrc[1] = __builtin_amdgcn_make_buffer_rsrc((void*)(P.src[1].p ? P.src[1].p + (size_t)n * img1 : P.src[0].p),
                                                  (short)0, P.src[1].p ? (int)img1 : 0, 0x00020000);
#pragma unroll
    for (int j = 0; j < C::NDA; ++j) {
      const int code = a_code[j];
      const int hy = code & 255, hx = (code >> 8) & 255, part = (code >> 16) & 255;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const DView S = P.src[k];
        const int y = ty0 + hy - 1 - S.oy, x = tx0 + hx - 1 - S.ox;
        a_g[k][j] = (code >= 0 && S.C > 0 && y >= 0 && y < S.H && x >= 0 && x < S.W)
                        ? (unsigned)(((y * S.W + x) * S.C) * 2 + part * 16) : OOB;
      }
    }
  };
  // wave-instruction j of the patch of `chunk` (of the DMA-side work item) into patch buffer `buf`
  auto dma_patch = [&](int chunk, int j, int buf, bool live) {
    const int idx = j * 8 + wave;
    live = live && idx < C::A_INSTR;
    char* dst = live ? smem + C::A_BASE + buf * C::A_BYTES + idx * 1024 : smem + C::DUMMY;
    const int ch = chunk * 64;
    if (ch < P.src[0].C) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc[0], (lds_void*)dst, 16, live ? a_g[0][j] : OOB,
                                               (unsigned)ch * 2, 0, 0);
    } else {
      // (the two candidates pass through an opaque copy: folded into a load through a selected POINTER they would take the
      //  whole a_g array out of registers -- scratch traffic inside the hand-counted vmcnt stream)
      unsigned o0 = a_g[0][j], o1 = a_g[1][j];
      asm volatile("" : "+v"(o0), "+v"(o1));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc[1], (lds_void*)dst, 16, live ? (dsrc ? o0 : o1) : OOB,
                                               (unsigned)(ch - P.src[0].C) * 2, 0, 0);
    }
  };
  auto dma_w = [&](unsigned wbase, int chunk, int tap, int slot, bool live) {
    const unsigned soff = live ? wbase + (unsigned)tap * w_tap_stride + (unsigned)chunk * 128 : 0u;
#pragma unroll
    for (int j = 0; j < C::NDW; ++j) {
      char* dst = live ? smem + C::W_BASE + slot * C::W_BYTES + (j * 8 + wave) * 1024 : smem + C::DUMMY;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)dst, 16, live ? w_g[j] : OOB, soff, 0, 0);
    }
  };

  // PAIR: the slabs of linear tap-steps (sA, sA + 1) of the work item whose first weight row is `wbase` into ring slot `slot`
  // (rows 0-63: tap A, rows 64-127: tap B; the same per-lane row / piece offsets, two scalar bases)
  auto dma_w2 = [&](unsigned wbase, int sA, int slot, bool live) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int sj = sA + j, cj = sj / 9, tj = sj - cj * 9;
      const unsigned soff = live ? wbase + (unsigned)tj * w_tap_stride + (unsigned)cj * 128 : 0u;
      char* dst = live ? smem + C::W_BASE + slot * C::W_BYTES + j * (BN * 128) + wave * 1024 : smem + C::DUMMY;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)dst, 16, live ? w_g[0] : OOB, soff, 0, 0);
    }
  };

  f32x4 acc[C::CT][4];
  // One tap = two 32-channel half-steps (ks) of CT x 4 MFMAs.  The fragment reads are software-pipelined BY HAND and
  // pinned with sched_barriers: left alone, hipcc funnels the weight fragments through one register quad and waits
  // for each ds_read right before its MFMAs (eight exposed LDS latencies per tap; SQ_WAIT_ANY 44 %).  Here every
  // fragment is requested at least four MFMAs before its first use; at most 11 fragments are live.
  auto compute = [&](int pbuf, int toff, int slot) {
    const char* pa = smem + C::W_BASE + slot * C::W_BYTES;
    const char* pb = smem + pbuf + toff;
    auto ra = [&](int ks, int ct) { return *reinterpret_cast<const bf16x8*>(pa + aoff[ct][ks]); };
    auto rb = [&](int ks, int pt) { return *reinterpret_cast<const bf16x8*>(pb + boff[pt] + ks * 64); };
    auto mm = [&](int ct, const bf16x8& fa, const bf16x8 (&fb)[4]) {
#pragma unroll
      for (int pt = 0; pt < 4; ++pt)
        acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb[pt], acc[ct][pt], 0, 0, 0);
    };
    bf16x8 fb0[4], fb1[4];
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) fb0[pt] = rb(0, pt);
    if constexpr (C::CT == 4) {
      bf16x8 a0 = ra(0, 0), a1 = ra(0, 1);
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 a2 = ra(0, 2), a3 = ra(0, 3);
      mm(0, a0, fb0);
      __builtin_amdgcn_sched_barrier(0);
      fb1[0] = rb(1, 0); fb1[1] = rb(1, 1);
      mm(1, a1, fb0);
      __builtin_amdgcn_sched_barrier(0);
      fb1[2] = rb(1, 2); fb1[3] = rb(1, 3);
      mm(2, a2, fb0);
      __builtin_amdgcn_sched_barrier(0);
      a0 = ra(1, 0); a1 = ra(1, 1);
      mm(3, a3, fb0);
      __builtin_amdgcn_sched_barrier(0);
      a2 = ra(1, 2);
      mm(0, a0, fb1);
      __builtin_amdgcn_sched_barrier(0);
      a3 = ra(1, 3);
      mm(1, a1, fb1);
      __builtin_amdgcn_sched_barrier(0);
      mm(2, a2, fb1);
      mm(3, a3, fb1);
    } else {
      // 8 MFMAs per half-step cannot cover an LDS round trip: the whole second half-step is fetched behind the first
      bf16x8 a0 = ra(0, 0), a1 = ra(0, 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) fb1[pt] = rb(1, pt);
      const bf16x8 a2 = ra(1, 0), a3 = ra(1, 1);
      mm(0, a0, fb0);
      mm(1, a1, fb0);
      __builtin_amdgcn_sched_barrier(0);
      mm(0, a2, fb1);
      mm(1, a3, fb1);
    }
  };

  // PAIR: two taps back to back (A then B: the accumulation order of two single taps); tap B's fragments are requested
  // behind tap A's MFMAs, so only the first half-step of a step waits for LDS
  auto compute2 = [&](int pbufA, int toffA, int pbufB, int toffB, int slot) {
    const char* pa = smem + C::W_BASE + slot * C::W_BYTES;
    const char* pbA = smem + pbufA + toffA;
    const char* pbB = smem + pbufB + toffB;
    auto ra = [&](int tb, int ks, int ct) { return *reinterpret_cast<const bf16x8*>(pa + tb * (BN * 128) + aoff[ct][ks]); };
    auto rb = [&](const char* pb, int ks, int pt) { return *reinterpret_cast<const bf16x8*>(pb + boff[pt] + ks * 64); };
    auto mm = [&](int ct, const bf16x8& fa_, const bf16x8 (&fb_)[4]) {
#pragma unroll
      for (int pt = 0; pt < 4; ++pt)
        acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_, fb_[pt], acc[ct][pt], 0, 0, 0);
    };
    bf16x8 f0[4], f1[4], g0[4], g1[4];
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) f0[pt] = rb(pbA, 0, pt);
    const bf16x8 a0 = ra(0, 0, 0), a1 = ra(0, 0, 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) f1[pt] = rb(pbA, 1, pt);
    const bf16x8 a2 = ra(0, 1, 0), a3 = ra(0, 1, 1);
    mm(0, a0, f0);
    mm(1, a1, f0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) g0[pt] = rb(pbB, 0, pt);
    const bf16x8 c0 = ra(1, 0, 0), c1 = ra(1, 0, 1);
    mm(0, a2, f1);
    mm(1, a3, f1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) g1[pt] = rb(pbB, 1, pt);
    const bf16x8 c2 = ra(1, 1, 0), c3 = ra(1, 1, 1);
    mm(0, c0, g0);
    mm(1, c1, g0);
    __builtin_amdgcn_sched_barrier(0);
    mm(0, c2, g1);
    mm(1, c3, g1);
  };

  // ROW3: the slabs of taps 3r .. 3r+2 of `chunk` into ring slot `slot`; the three taps back to back, fragments two
  // half-steps ahead of their MFMAs
  auto dma_w3 = [&](unsigned wbase, int chunk, int r, int slot, bool live) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const unsigned soff = live ? wbase + (unsigned)(3 * r + j) * w_tap_stride + (unsigned)chunk * 128 : 0u;
      char* dst = live ? smem + C::W_BASE + slot * C::W_BYTES + j * (BN * 128) + wave * 1024 : smem + C::DUMMY;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)dst, 16, live ? w_g[0] : OOB, soff, 0, 0);
    }
  };
  auto compute3 = [&](int pbuf, int r, int slot) {
    const char* pa = smem + C::W_BASE + slot * C::W_BYTES;
    const char* pb = smem + pbuf + r * C::RS;
    bf16x8 fbq[3][4], faq[3][2];
    auto load = [&](int h, int q) {              // half-step h = 2 * tap + ks into fragment set q
      const int tj = h >> 1, ks = h & 1;
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) fbq[q][pt] = *reinterpret_cast<const bf16x8*>(pb + tj * C::PSTR + boff[pt] + ks * 64);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) faq[q][ct] = *reinterpret_cast<const bf16x8*>(pa + tj * (BN * 128) + aoff[ct][ks]);
    };
    load(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    load(1, 1);
#pragma unroll
    for (int h = 0; h < 6; ++h) {
      const int q = h % 3;
      if (h + 2 < 6) load(h + 2, (h + 2) % 3);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int pt = 0; pt < 4; ++pt)
          acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(faq[q][ct], fbq[q][pt], acc[ct][pt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // PP: the same tap as two halves -- every fragment of the tap into registers, then nothing but MFMAs
  bf16x8 fa[2][C::CT], fb[2][4];
  auto load_frags = [&](int pbuf, int toff, int slot) {
    const char* pa = smem + C::W_BASE + slot * C::W_BYTES;
    const char* pb = smem + pbuf + toff;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) fb[ks][pt] = *reinterpret_cast<const bf16x8*>(pb + boff[pt] + ks * 64);
#pragma unroll
      for (int ct = 0; ct < C::CT; ++ct) fa[ks][ct] = *reinterpret_cast<const bf16x8*>(pa + aoff[ct][ks]);
    }
  };
  auto mma_frags = [&]() {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int ct = 0; ct < C::CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < 4; ++pt)
          acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ks][ct], fb[ks][pt], acc[ct][pt], 0, 0, 0);
  };

  // prologue: patch of chunk 0 and the first two weight slabs of the first work item
  setup_dma(logical);
#pragma unroll
  for (int j = 0; j < C::NDA; ++j) dma_patch(0, j, 0, true);
  if constexpr (ROW3) {
    dma_w3(d_wbase, 0, 0, 0, true);
  } else if constexpr (PAIR) {
    dma_w2(d_wbase, 0, 0, true);
    dma_w2(d_wbase, 2, 1, true);
  } else {
    dma_w(d_wbase, 0, 0, 0, true);
    dma_w(d_wbase, 0, 1, 1, true);
  }
  if constexpr (PP) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NDW) : "memory");      // patch 0 + W(0) landed; W(1) in flight
    __builtin_amdgcn_s_barrier();
    if (grp) __builtin_amdgcn_s_barrier();                              // the stagger: waves 4-7 one barrier behind
  }

#ifdef PDMA_STAMPS
  // diagnostic build: per-wave cycle sums of (vmcnt wait, barrier, DMA issue, fragment reads + MFMAs) over all taps
  unsigned long long st_sum[4] = {0, 0, 0, 0}, st_prev = 0, st_taps = 0, st_epi = 0, st_b2 = 0, st_rd = 0;
  const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  int pbuf_i = 0;                                 // patch buffer of the chunk being computed
  int wslot = 0;                                  // ROW3: ring slot of the step being computed
  bool after_epilogue = false;
  const bool late_dma = !PP && P.pdma_stagger && __builtin_amdgcn_readfirstlane(wave) < 4;
  // BatchNorm partial sums of this lane's outputs (4 channels x CT tiles, 2 statistics).  Block mode (P.zdiv: every
  // block visits every channel tile): a layer of thousands of tiles has 256 partials to finalise -- the block keeps a
  // running total per (statistic, channel) over its work items of one channel tile.  BN = 64 (DEFER): the per-lane sums
  // themselves run on across those items and are reduced over lanes and waves ONCE, at the last of them (the 64 DPP adds
  // + LDS exchange + barrier leave the per-item epilogue: +3 %).  At BN = 128 that is 32 more live registers: the lock-step
  // forward kernel has them since the output addressing went scalar (215 -> 247 VGPRs, +0..4 % per layer,
  // profiles/r03_pdma_dense_epilogue.txt); the ping-pong and BatchNorm-backward instantiations (251 / 236) would spill, so
  // there every item reduces and a thread carries the total.  Fixed order either way: deterministic.
  constexpr bool DEFER = BN == 64 || PDMA_DEFER128;
  // Output addressing of dense destinations (P.pdma_dense: every destination view covers the frame at offset 0; frames are
  // whole 16x16 tiles here anyway): a lane's offset inside a (tile, 32-channel pair) never changes -- lp[view]: pixel row 0
  // of its four; rows 1-3 through the scalar offset operand, which the range check ignores -- and the work item enters
  // through the descriptor's base address.  Scalar arithmetic per item instead of ~25 vector instructions per store in
  // an epilogue that all eight waves run together (stamps: 10-20 % of a 128/256-channel layer's launch).
  const bool dense = BNBWD || P.pdma_dense != 0;
  unsigned lp[2];
#pragma unroll
  for (int q = 0; q < 2; ++q)
    lp[q] = (unsigned)((((wpx * 4) * P.dst[q].W + l15) * P.dst[q].C + (kb & 1) * 16 + (kb >> 1) * 8) * 2);
  float stat_tot = 0.f;
  float bs[C::CT][4], bq[C::CT][4];
#pragma unroll
  for (int ct = 0; ct < C::CT; ++ct)
#pragma unroll
    for (int j = 0; j < 4; ++j) { bs[ct][j] = 0.f; bq[ct][j] = 0.f; }
  for (int wk = logical; wk < total; wk += G) {
    int cot, tile;
    pdma_item(wk, n_tiles, P.co_il, cot, tile);
    const int n = tile / tiles_img, r = tile - n * tiles_img;
    const int tyi = r / P.tilesX, txi = r - tyi * P.tilesX;
    const int ty0 = tyi * C::TH, tx0 = txi * C::TW;
    const int co0 = cot * BN;
    const unsigned c_wbase = d_wbase;             // this work item's weights (DMA side moves on in the last chunk)
    const bool has_next = wk + G < total;
#pragma unroll
    for (int a = 0; a < C::CT; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[a][b][q] = 0.f;

    if constexpr (ROW3) {
      // three steps (tap rows) per chunk.  W(step) was issued during the previous step, behind that step's patch pieces: the
      // youngest operations in flight [+ the output stores of the previous item's epilogue] -> wait for everything older
      for (int c = 0; c < nchunks; ++c) {
        const bool last = c + 1 == nchunks;
        if (last) {                                // from here on the DMA stream belongs to the next work item
          d_live = has_next;
          if (has_next) setup_dma(wk + G);
        }
        const int pbuf = C::A_BASE + pbuf_i * C::A_BYTES;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          if (r == 0 && c == 0 && after_epilogue) {
            if (P.stats) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NST + 1) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NST) : "memory");
          } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
          __builtin_amdgcn_s_barrier();
          const int j0 = r == 0 ? 0 : (r == 1 ? 3 : 5), nj = r == 0 ? 3 : 2;       // patch pieces 0-2 | 3,4 | 5,6 of the next chunk
#pragma unroll
          for (int q = 0; q < 3; ++q)
            if (q < nj) dma_patch(last ? 0 : c + 1, j0 + q, pbuf_i ^ 1, last ? d_live : true);
          if (r < 2) dma_w3(c_wbase, c, r + 1, wslot ^ 1, true);
          else if (!last) dma_w3(c_wbase, c + 1, 0, wslot ^ 1, true);
          else dma_w3(d_wbase, 0, 0, wslot ^ 1, d_live);
          compute3(pbuf, r, wslot);
          wslot ^= 1;
        }
        pbuf_i ^= 1;
      }
    } else if constexpr (PAIR) {
      // 9 steps of two taps; chunk 0 lives in patch buffer 0, chunk 1 in buffer 1 (nchunks == 2: the launcher's condition).
      // Patch pieces: steps 0-3 bring THIS item's chunk 1 (2, 2, 2, 1 pieces per wave), steps 5-8 the NEXT item's chunk 0
      // (buffer 0 is read for the last time by step 4); the weights of step d + 2 follow the pieces of step d.
#pragma unroll
      for (int d = 0; d < 9; ++d) {
        constexpr int NPIECE[9] = {2, 2, 2, 1, 0, 2, 2, 2, 1};
        if (d == 5) {                              // from here on the DMA stream belongs to the next work item
          d_live = has_next;
          if (has_next) setup_dma(wk + G);
        }
        // W(d) was issued two steps ago; younger: the previous step's patch pieces + NDW weight DMAs [+ the output stores of
        // the previous item's epilogue].  Step 4 also needs chunk 1's LAST patch piece, issued in step 3 in front of W(5)
        if (d == 0) {
          if (after_epilogue) {
            if (P.stats) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NDW + C::NST + 1) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NDW + C::NST) : "memory");
          } else {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NDW) : "memory");
          }
        } else if (d == 4) {
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NDW) : "memory");
        } else {
          const int np = NPIECE[d == 0 ? 0 : d - 1];
          if (np == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NDW + 2) : "memory");
          else if (np == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NDW + 1) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NDW) : "memory");
        }
        __builtin_amdgcn_s_barrier();
        auto issue_dma = [&]() {
          const int first = d < 4 ? 2 * d : 2 * (d - 5);           // (steps 0-3 / 5-8: pieces 0,1 | 2,3 | 4,5 | 6)
#pragma unroll
          for (int q = 0; q < NPIECE[d]; ++q) {
            if (d < 4) dma_patch(1, first + q, 1, true);
            else dma_patch(0, first + q, 0, d_live);
          }
          if (d + 2 < 9) dma_w2(c_wbase, 2 * (d + 2), (d + 2) % 3, true);
          else dma_w2(d_wbase, 2 * (d + 2 - 9), (d + 2) % 3, d_live);
        };
        if (!late_dma) issue_dma();
        const int sA = 2 * d, sB = 2 * d + 1;
        const int cA = sA / 9, tA = sA % 9, cB = sB / 9, tB = sB % 9;
        compute2(C::A_BASE + cA * C::A_BYTES, (tA / 3) * C::RS + (tA % 3) * C::PSTR,
                 C::A_BASE + cB * C::A_BYTES, (tB / 3) * C::RS + (tB % 3) * C::PSTR, d % 3);
        if (late_dma) issue_dma();
      }
    } else
    for (int c = 0; c < nchunks; ++c) {
      const bool last = c + 1 == nchunks;
      if (last) {                                  // from here on the DMA stream belongs to the next work item
        d_live = has_next;
        if (has_next) setup_dma(wk + G);
      }
      const int pbuf = C::A_BASE + pbuf_i * C::A_BYTES;
      if constexpr (PP) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          // ---- LOAD
#ifdef PDMA_STAMPS
          const unsigned long long st_a = __builtin_amdgcn_s_memtime();
          if (st_prev) st_b2 += st_a - st_prev;
#endif
          load_frags(pbuf, (tap / 3) * C::RS + (tap % 3) * C::PSTR, tap % 3);
          __builtin_amdgcn_sched_barrier(0);
#ifdef PDMA_STAMPS
          st_rd += __builtin_amdgcn_s_memtime() - st_a;
#endif
          if (tap < C::NDA) dma_patch(last ? 0 : c + 1, tap, pbuf_i ^ 1, last ? d_live : true);
          if (tap + 2 < 9) dma_w(c_wbase, c, tap + 2, (tap + 2) % 3, true);
          else if (!last) dma_w(c_wbase, c + 1, tap + 2 - 9, (tap + 2) % 3, true);
          else dma_w(d_wbase, 0, tap + 2 - 9, (tap + 2) % 3, d_live);
          // everything older than THIS phase's DMAs has landed: the next step's weight slab (issued one step ago) and,
          // by then, every patch piece of the next chunk [the previous work item's output stores sit in between]
#ifdef PDMA_STAMPS
          const unsigned long long st_b = __builtin_amdgcn_s_memtime();
          st_sum[0] += st_b - st_a;
#endif
          if (tap == 0 && c == 0 && after_epilogue) {                  // (tap 0 always carries a patch piece)
            if (P.stats) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 + C::NDW + C::NST + 1) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 + C::NDW + C::NST) : "memory");
          } else if (tap < C::NDA) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 + C::NDW) : "memory");
          } else {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NDW) : "memory");
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef PDMA_STAMPS
          const unsigned long long st_c = __builtin_amdgcn_s_memtime();
          st_sum[1] += st_c - st_b;
#endif
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
#ifdef PDMA_STAMPS
          const unsigned long long st_d = __builtin_amdgcn_s_memtime();
          st_sum[2] += st_d - st_c;
#endif
          // ---- COMPUTE
          __builtin_amdgcn_s_setprio(1);
          mma_frags();
          __builtin_amdgcn_s_setprio(0);
          __builtin_amdgcn_sched_barrier(0);
#ifdef PDMA_STAMPS
          st_prev = __builtin_amdgcn_s_memtime();
          st_sum[3] += st_prev - st_d;
          st_taps += 1;
#endif
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
#ifdef PDMA_STAMPS
        const unsigned long long st_a = __builtin_amdgcn_s_memtime();
        if (st_prev) st_sum[3] += st_a - st_prev;
#endif
        // W(step) was issued two steps ago; younger: the previous step's [patch DMA] + NDW weight DMAs
        // [+ the NST (+1) output stores of the previous work item's epilogue]
        if (tap == 0 && c == 0 && after_epilogue) {
          if (P.stats) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NDW + C::NST + 1) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NDW + C::NST) : "memory");
        } else if (tap == 0 || tap == 8) {
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NDW) : "memory");          // previous tap 8 / 7: no patch DMA
        } else {
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NDW + 1) : "memory");
        }
#ifdef PDMA_STAMPS
        const unsigned long long st_b = __builtin_amdgcn_s_memtime();
        st_sum[0] += st_b - st_a;
#endif
        __builtin_amdgcn_s_barrier();
#ifdef PDMA_STAMPS
        const unsigned long long st_c = __builtin_amdgcn_s_memtime();
        st_sum[1] += st_c - st_b;
#endif
        // This tap's DMA issues (a patch piece of the next chunk, the weight slab two taps ahead).  The two waves of a SIMD
        // (w, w + 4) issue at opposite ends of the tap -- waves 4-7 here, waves 0-3 behind their MFMAs -- so that one of
        // them has MFMAs to issue while the other sits in its burst (UNET_PDMA_STG=0: all eight behind the barrier, as in
        // round 2).  The per-wave ORDER of vector-memory operations is unchanged, so every counted vmcnt above still holds;
        // a slot is refilled after the barrier that follows its last reads either way.
        auto issue_dma = [&]() {
          if (tap < C::NDA) dma_patch(last ? 0 : c + 1, tap, pbuf_i ^ 1, last ? d_live : true);
          if (tap + 2 < 9) dma_w(c_wbase, c, tap + 2, (tap + 2) % 3, true);
          else if (!last) dma_w(c_wbase, c + 1, tap + 2 - 9, (tap + 2) % 3, true);
          else dma_w(d_wbase, 0, tap + 2 - 9, (tap + 2) % 3, d_live);
        };
        if (!late_dma) issue_dma();
#ifdef PDMA_STAMPS
        st_prev = __builtin_amdgcn_s_memtime();
        st_sum[2] += st_prev - st_c;
        st_taps += 1;
#endif
        compute(pbuf, (tap / 3) * C::RS + (tap % 3) * C::PSTR, tap % 3);
        if (late_dma) issue_dma();
      }
      }
      pbuf_i ^= 1;
    }

#ifdef PDMA_STAMPS
    { const unsigned long long t = __builtin_amdgcn_s_memtime(); if (!PP) st_sum[3] += t - st_prev; st_prev = 0; st_epi -= t; }
#endif
    // ---- epilogue: D of 16x16x32: col = lane&15 (pixel), rows (lane>>4)*4 + reg (4 consecutive channels).
    // Buffer stores (out-of-range offset = dropped) so every lane issues exactly NST of them.
    __amdgpu_buffer_rsrc_t drs[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const DViewW D = P.dst[q];
      const unsigned dimg = (unsigned)D.H * D.W * D.C * 2u;
      drs[q] = __builtin_amdgcn_make_buffer_rsrc((void*)(D.p ? D.p + (size_t)n * dimg : P.dst[0].p), (short)0,
                                                 D.p ? (int)dimg : 0, 0x00020000);
    }
    if constexpr (!DEFER) {
#pragma unroll
      for (int ct = 0; ct < C::CT; ++ct)
#pragma unroll
        for (int j = 0; j < 4; ++j) { bs[ct][j] = 0.f; bq[ct][j] = 0.f; }
    }
    if constexpr (BNBWD) {
      // dgrad + ReLU mask + BatchNorm-backward sums of the producing layer.  dst[0] is dense and frame-sized, so the
      // store offset of a (pixel, tile pair) is also the offset of its 8 y values; y comes in with the same 16-byte
      // loads as the gradient fan-in's old values and is un-swapped to the accumulator layout.  ALL loads of the work
      // item are issued before the first use: one exposed memory round trip per item.
      const DViewW D = P.dst[0];
      const unsigned dimg = (unsigned)D.H * D.W * D.C * 2u;
      const __amdgpu_buffer_rsrc_t yrs =
          __builtin_amdgcn_make_buffer_rsrc((void*)(P.bn_y + (size_t)n * dimg), (short)0, (int)dimg, 0x00020000);
      u32x4 yraw[4][C::CT / 2];
      f32x4 csc[C::CT / 2][2], csh[C::CT / 2][2], cmu[C::CT / 2][2];
      (void)yrs;
      const unsigned rowb = (unsigned)(D.W * D.C * 2);
      __amdgpu_buffer_rsrc_t yrs_c[C::CT / 2], drs_c[C::CT / 2];
#pragma unroll
      for (int cp = 0; cp < C::CT / 2; ++cp) {
        const int cw = co0 + wco * (BN / 2) + cp * 32;
        const unsigned off = (unsigned)(((ty0 * D.W + tx0) * D.C + cw) * 2);
        yrs_c[cp] = __builtin_amdgcn_make_buffer_rsrc((void*)(P.bn_y + (size_t)n * dimg + off), (short)0, (int)(dimg - off), 0x00020000);
        drs_c[cp] = __builtin_amdgcn_make_buffer_rsrc((void*)(D.p + (size_t)n * dimg + off), (short)0, (int)(dimg - off), 0x00020000);
      }
#pragma unroll
      for (int pt = 0; pt < 4; ++pt)
#pragma unroll
        for (int cp = 0; cp < C::CT / 2; ++cp) yraw[pt][cp] = __builtin_amdgcn_raw_buffer_load_b128(yrs_c[cp], lp[0], pt * rowb, 0);
#pragma unroll
      for (int cp = 0; cp < C::CT / 2; ++cp) {
        const int cw = co0 + wco * (BN / 2) + cp * 32 + kb * 4;   // native layout: tile 2cp rows kb*4.., +16: tile 2cp+1
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          csc[cp][t] = *reinterpret_cast<const f32x4*>(P.bn_scale + cw + 16 * t);
          csh[cp][t] = *reinterpret_cast<const f32x4*>(P.bn_shift + cw + 16 * t);
          cmu[cp][t] = *reinterpret_cast<const f32x4*>(P.bn_mean + cw + 16 * t);
        }
      }
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) {
#pragma unroll
        for (int cp = 0; cp < C::CT / 2; ++cp) {
          const u32x4 o = yraw[pt][cp];
          const auto o0 = __builtin_amdgcn_permlane16_swap(o[0], o[2], false, false);
          const auto o1 = __builtin_amdgcn_permlane16_swap(o[1], o[3], false, false);
          const bf16x4 ya = __builtin_bit_cast(bf16x4, u32x2{o0[0], o1[0]});
          const bf16x4 yb = __builtin_bit_cast(bf16x4, u32x2{o0[1], o1[1]});
          bf16x4 ra, rb;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float fa = (float)ya[j], fb = (float)yb[j];
            const bool ona = fmaf(fa, csc[cp][0][j], csh[cp][0][j]) > 0.f;
            const bool onb = fmaf(fb, csc[cp][1][j], csh[cp][1][j]) > 0.f;
            ra[j] = (bf16_t)(ona ? acc[2 * cp][pt][j] : 0.f);
            rb[j] = (bf16_t)(onb ? acc[2 * cp + 1][pt][j] : 0.f);
            const float qa = (float)ra[j], qb = (float)rb[j];       // dz as stored (an OOB pixel loads y = 0 and is
            bs[2 * cp][j] += qa;                                    //  dropped by its store: frames are 16-aligned here,
            bq[2 * cp][j] = fmaf(qa, fa - cmu[cp][0][j], bq[2 * cp][j]);   // so that never happens)
            bs[2 * cp + 1][j] += qb;
            bq[2 * cp + 1][j] = fmaf(qb, fb - cmu[cp][1][j], bq[2 * cp + 1][j]);
          }
          const u32x2 ua = __builtin_bit_cast(u32x2, ra), ub = __builtin_bit_cast(u32x2, rb);
          const auto s0 = __builtin_amdgcn_permlane16_swap(ua[0], ub[0], false, false);
          const auto s1 = __builtin_amdgcn_permlane16_swap(ua[1], ub[1], false, false);
          __builtin_amdgcn_raw_buffer_store_b128(u32x4{s0[0], s1[0], s0[1], s1[1]}, drs_c[cp], lp[0], pt * rowb, 0);
        }
      }
    } else {
    // v_permlane16_swap trades the (kb odd) rows of tile ct for the (kb even) rows of tile ct+1: afterwards lane kb
    // holds 8 CONSECUTIVE channels -- tile ct + (kb & 1), channels 8*(kb >> 1) .. +7 -- and writes 16 bytes (half
    // the store instructions, 64 contiguous bytes per pixel and tile pair).  Old values for the gradient fan-in
    // come in with the same 16-byte loads and are un-swapped (the exchange is an involution) before the fp32 add.
    // (pt, cp): the wave's pixel row and 32-channel tile pair; rs / vo / so: descriptor, lane offset, scalar offset of its store
    auto finish = [&](int pt, int cp, int cw, bool second, int accq, __amdgpu_buffer_rsrc_t rs, unsigned vo, unsigned so, bool ok) {
      float va[4], vb[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { va[j] = acc[2 * cp][pt][j]; vb[j] = acc[2 * cp + 1][pt][j]; }
      if (P.bias) {                                // inference: BatchNorm shift (+ ReLU) of the folded layer
        const float* bp = P.bias + cw + kb * 4;   // native accumulator layout: tile 2cp (+16: tile 2cp+1), rows kb*4..+3
#pragma unroll
        for (int j = 0; j < 4; ++j) { va[j] += bp[j]; vb[j] += bp[16 + j]; }
      }
      if (P.relu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { va[j] = fmaxf(va[j], 0.f); vb[j] = fmaxf(vb[j], 0.f); }
      }
      if (accq) {
        const u32x4 o = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, 0);
        const auto o0 = __builtin_amdgcn_permlane16_swap(o[0], o[2], false, false);
        const auto o1 = __builtin_amdgcn_permlane16_swap(o[1], o[3], false, false);
        const bf16x4 oa = __builtin_bit_cast(bf16x4, u32x2{o0[0], o1[0]});
        const bf16x4 ob = __builtin_bit_cast(bf16x4, u32x2{o0[1], o1[1]});
#pragma unroll
        for (int j = 0; j < 4; ++j) { va[j] += (float)oa[j]; vb[j] += (float)ob[j]; }
      }
      bf16x4 ra, rb;
#pragma unroll
      for (int j = 0; j < 4; ++j) { ra[j] = (bf16_t)va[j]; rb[j] = (bf16_t)vb[j]; }
      const u32x2 ua = __builtin_bit_cast(u32x2, ra), ub = __builtin_bit_cast(u32x2, rb);
      const auto s0 = __builtin_amdgcn_permlane16_swap(ua[0], ub[0], false, false);
      const auto s1 = __builtin_amdgcn_permlane16_swap(ua[1], ub[1], false, false);
      __builtin_amdgcn_raw_buffer_store_b128(u32x4{s0[0], s1[0], s0[1], s1[1]}, rs, vo, so, 0);
      if (ok && P.stats) {                         // (a launch without statistics skips the 24 vector instructions per store)
#pragma unroll
        for (int j = 0; j < 4; ++j) {              // statistics of the values as STORED (bf16-rounded)
          const float qa = (float)ra[j], qb = (float)rb[j];
          bs[2 * cp][j] += qa;
          bq[2 * cp][j] = fmaf(qa, qa, bq[2 * cp][j]);
          bs[2 * cp + 1][j] += qb;
          bq[2 * cp + 1][j] = fmaf(qb, qb, bq[2 * cp + 1][j]);
        }
      }
    };
    if (dense) {
#pragma unroll
      for (int cp = 0; cp < C::CT / 2; ++cp) {
        const int cw = co0 + wco * (BN / 2) + cp * 32;               // first channel of the tile pair
        const bool second = cw >= P.dst_split;                       // uniform per (wave, pair): dst_split % 64 == 0
        const int accq = second ? (P.accumulate & 2) : (P.accumulate & 1);
        const DViewW D = second ? P.dst[1] : P.dst[0];
        const unsigned dimg = (unsigned)D.H * D.W * D.C * 2u;
        const unsigned off = (unsigned)(((ty0 * D.W + tx0) * D.C + cw - (second ? P.dst_split : 0)) * 2);
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc((void*)(D.p + (size_t)n * dimg + off), (short)0, (int)(dimg - off), 0x00020000);
        const unsigned rowb = (unsigned)(D.W * D.C * 2);
        const unsigned lpq = second ? lp[1] : lp[0];
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) finish(pt, cp, cw, second, accq, rs, lpq, pt * rowb, true);
      }
    } else {
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) {
        const int fy = ty0 + wpx * 4 + pt, fx = tx0 + l15;
        const bool pix_ok = fy < P.H && fx < P.W;
#pragma unroll
        for (int cp = 0; cp < C::CT / 2; ++cp) {
          const int cw = co0 + wco * (BN / 2) + cp * 32;               // first channel of the tile pair
          const bool second = cw >= P.dst_split;                       // uniform per (wave, pair): dst_split % 64 == 0
          const int accq = second ? (P.accumulate & 2) : (P.accumulate & 1);
          const DViewW D = second ? P.dst[1] : P.dst[0];
          const int co = cw - (second ? P.dst_split : 0) + (kb & 1) * 16 + (kb >> 1) * 8;
          const int y = fy - D.oy, x = fx - D.ox;
          const bool ok = pix_ok && y >= 0 && y < D.H && x >= 0 && x < D.W;
          const unsigned vo = ok ? (unsigned)(((y * D.W + x) * D.C + co) * 2) : OOB;
          if (second) finish(pt, cp, cw, true, accq, drs[1], vo, 0u, ok);
          else finish(pt, cp, cw, false, accq, drs[0], vo, 0u, ok);
        }
      }
    }
    }
    if (P.stats) {
      // exactly one (possibly dropped) statistics store per work item: static vmcnt counts
      int ncot = cot, ntile_ = 0;
      if (has_next) pdma_item(wk + G, n_tiles, P.co_il, ncot, ntile_);
      (void)ntile_;
      const bool flush = !P.zdiv || !has_next || ncot != cot;
      // block mode: the co_il blocks that share a super-group's pixel tiles own different channel tiles -> ONE partial row
      const int part = P.zdiv ? logical / P.co_il : (n * P.tilesY + tyi) * P.tilesX + txi;
      float tsum = 0.f;
      unsigned so = OOB;
      if (!DEFER || flush) {
        {
          float rv[2 * C::CT * 4];
#pragma unroll
          for (int ct = 0; ct < C::CT; ++ct)
#pragma unroll
            for (int j = 0; j < 4; ++j) { rv[ct * 4 + j] = bs[ct][j]; rv[C::CT * 4 + ct * 4 + j] = bq[ct][j]; }
          row16_sum_n(rv);
#pragma unroll
          for (int ct = 0; ct < C::CT; ++ct)
#pragma unroll
            for (int j = 0; j < 4; ++j) { bs[ct][j] = rv[ct * 4 + j]; bq[ct][j] = rv[C::CT * 4 + ct * 4 + j]; }
        }
        float* red = reinterpret_cast<float*>(smem + C::RED_BASE);     // [4 pixel-waves][2][BN]
        if (l15 == 0) {
#pragma unroll
          for (int ct = 0; ct < C::CT; ++ct)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int cl = wco * (BN / 2) + ct * 16 + kb * 4 + j;
              red[(wpx * 2 + 0) * BN + cl] = bs[ct][j];
              red[(wpx * 2 + 1) * BN + cl] = bq[ct][j];
            }
        }
#pragma unroll
        for (int ct = 0; ct < C::CT; ++ct)
#pragma unroll
          for (int j = 0; j < 4; ++j) { bs[ct][j] = 0.f; bq[ct][j] = 0.f; }
        // LDS-only exchange: raw barrier (a __syncthreads() would also wait for the output stores)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // PP: that barrier is this half's own exchange (the other half is a barrier apart); a half owns BN/2 channels
        // outright, so its BN threads (statistic, channel) total the four pixel-waves of THEIR half
        const int st_t = PP ? (tid & 255) : tid;
        if (st_t < (PP ? BN : 2 * BN)) {
          const int q = PP ? st_t / (BN / 2) : st_t / BN;
          const int cl = PP ? grp * (BN / 2) + st_t % (BN / 2) : st_t - q * BN;
#pragma unroll
          for (int wp = 0; wp < 4; ++wp) tsum += red[(wp * 2 + q) * BN + cl];   // fixed order: deterministic
          if (!DEFER && P.zdiv) {
            stat_tot += tsum;
            tsum = stat_tot;
            if (flush) stat_tot = 0.f;
          }
          if (flush) so = (unsigned)((((size_t)part * 2 + q) * P.Cout + co0 + cl) * 4);
        }
      }
      const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)P.stats, (short)0, (int)std::min<long long>((long long)n_tiles * 2 * P.Cout * 4, 0x7FFFFFFFLL), 0x00020000);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, tsum), srs, so, 0, 0);
    }
    after_epilogue = true;
#ifdef PDMA_STAMPS
    st_epi += __builtin_amdgcn_s_memtime();
#endif
  }
#ifdef PDMA_STAMPS
  if (P.bn_mean && lane == 0) {
    unsigned long long* o = (unsigned long long*)P.bn_mean + ((size_t)blockIdx.x * 8 + wave) * 8;
    o[0] = st_sum[0]; o[1] = st_sum[1]; o[2] = st_sum[2]; o[3] = st_sum[3]; o[4] = st_taps; o[5] = PP ? st_rd : st_epi;
    o[6] = st_b2;                                    // PP: wait at the barrier that ends COMPUTE
    o[7] = ((__builtin_amdgcn_s_memtime() - st_t0) << 20) / (__builtin_amdgcn_s_memrealtime() - st_r0 + 1);   // clock / 100 MHz, x 2^20
  }
#endif
  if constexpr (PP) { if (!grp) __builtin_amdgcn_s_barrier(); }        // pairs with the stagger barrier of waves 4-7
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // drain the dummy DMAs before the wave ends
}

__global__ __launch_bounds__(512, 1) void conv3_pdma128_kernel(const IgemmParams P) { conv3_pdma_body<128>(P); }
__global__ __launch_bounds__(512, 1) void conv3_pdma64_kernel(const IgemmParams P) { conv3_pdma_body<64>(P); }
__global__ __launch_bounds__(512, 1) void conv3_pdma128_bnbwd_kernel(const IgemmParams P) { conv3_pdma_body<128, true>(P); }
__global__ __launch_bounds__(512, 1) void conv3_pdma64_bnbwd_kernel(const IgemmParams P) { conv3_pdma_body<64, true>(P); }
__global__ __launch_bounds__(512, 1) void conv3_pp128_kernel(const IgemmParams P) { conv3_pdma_body<128, false, true>(P); }
__global__ __launch_bounds__(512, 1) void conv3_pp64_kernel(const IgemmParams P) { conv3_pdma_body<64, false, true>(P); }
__global__ __launch_bounds__(512, 1) void conv3_pp128_bnbwd_kernel(const IgemmParams P) { conv3_pdma_body<128, true, true>(P); }
__global__ __launch_bounds__(512, 1) void conv3_pp64_bnbwd_kernel(const IgemmParams P) { conv3_pdma_body<64, true, true>(P); }
__global__ __launch_bounds__(512, 1) void conv3_pdma64x2_kernel(const IgemmParams P) { conv3_pdma_body<64, false, false, true>(P); }
__global__ __launch_bounds__(512, 1) void conv3_pdma64x2_bnbwd_kernel(const IgemmParams P) { conv3_pdma_body<64, true, false, true>(P); }
__global__ __launch_bounds__(512, 1) void conv3_pdma64x3_kernel(const IgemmParams P) { conv3_pdma_body<64, false, false, false, true>(P); }
__global__ __launch_bounds__(512, 1) void conv3_pdma64x3_bnbwd_kernel(const IgemmParams P) { conv3_pdma_body<64, true, false, false, true>(P); }

#ifdef PDMA_STAMPS
void* g_pdma_debug = nullptr;      // (also read by wgrad.hip)
#endif

template <int BN>
int32_t launch_pdma(const IgemmParams& Pin, int kclass, hipStream_t s, int* stat_parts) {
  using C = CfgP<BN>;
  IgemmParams P = Pin;
  P.nCo = P.Cout / BN;
  P.tilesX = cdiv(P.W, C::TW);
  P.tilesY = cdiv(P.H, C::TH);
  const bool bnbwd = P.bn_y != nullptr;
  // the ping-pong schedule wins where a work item is long (>= 8 chunks: +2 % at 512, +6 % at 1024 input channels) and
  // loses where the epilogue -- run once per half, each exposed -- is a large part of an item (-10 % at 128 channels);
  // UNET_PDMA_PP=0 / 1 force lock-step / ping-pong
  P.pdma_dense = unet_tuning().pdma_stg != '2';             // (UNET_PDMA_STG=2: the per-lane output geometry, for A/B)
  P.pdma_dense_src = unet_tuning().pdma_stg != '2' && unet_tuning().pdma_stg != '4';      // (4: per-lane patch geometry only)
  for (int k = 0; k < 2; ++k)
    if (P.src[k].p && P.src[k].C > 0 &&
        (P.src[k].oy || P.src[k].ox || P.src[k].H != P.H || P.src[k].W != P.W || P.src[k].C != P.src[0].C))
      P.pdma_dense_src = 0;
  for (int q = 0; q < 2; ++q)
    if (P.dst[q].p && (P.dst[q].oy || P.dst[q].ox || P.dst[q].H != P.H || P.dst[q].W != P.W)) P.pdma_dense = 0;
  P.pdma_stagger = unet_tuning().pdma_stg != '0';           // default on: +3..8 % on the lock-step layers (profiles/r03_pdma_stagger.txt)
  const char ppv = unet_tuning().pdma_pp;
  const bool pp = ppv == '1' || (ppv != '0' && BN == 128 && P.Ctot >= 512);
  // two taps per step for 64-channel tiles over exactly two chunks (UNET_PDMA_PAIR=0: one tap per step, for A/B)
  const bool row3 = BN == 64 && !pp && unet_tuning().pdma_pair == '3';          // UNET_PDMA_PAIR=3: a tap row per step (A/B)
  const bool pair = BN == 64 && !pp && !row3 && P.Ctot == 128 && unet_tuning().pdma_pair != '0';
  auto kern = row3 ? (bnbwd ? conv3_pdma64x3_bnbwd_kernel : conv3_pdma64x3_kernel)
              : pair ? (bnbwd ? conv3_pdma64x2_bnbwd_kernel : conv3_pdma64x2_kernel)
              : pp ? (bnbwd ? (BN == 128 ? conv3_pp128_bnbwd_kernel : conv3_pp64_bnbwd_kernel)
                            : (BN == 128 ? conv3_pp128_kernel : conv3_pp64_kernel))
                   : (bnbwd ? (BN == 128 ? conv3_pdma128_bnbwd_kernel : conv3_pdma64_bnbwd_kernel)
                            : (BN == 128 ? conv3_pdma128_kernel : conv3_pdma64_kernel));
  const int lds_bytes = row3 ? CfgP<64, false, true>::LDS : (pair ? CfgP<64, true>::LDS : C::LDS);
  unet_set_max_lds(reinterpret_cast<const void*>(kern), lds_bytes);
  const long long work = (long long)P.N * P.tilesY * P.tilesX * P.nCo;
  UNET_REQUIRE(work > 0 && work < (1LL << 30), UNET_ERR_UNSUPPORTED, "conv3_pdma: %lld work items", work);
  const long long stat_bytes = (long long)P.N * P.tilesY * P.tilesX * 2 * P.Cout * 4;
  if (stat_bytes >= 0x7FFFFFFFLL) {
    UNET_REQUIRE(!bnbwd, UNET_ERR_UNSUPPORTED, "conv3_pdma: partial-sum buffer of %lld bytes", stat_bytes);
    P.stats = nullptr;
  }
  const int blocks = (int)std::min<long long>(unet_cu_budget(), cdiv64(work, 8) * 8);   // one per (non-reserved) CU, a multiple of 8 (XCDs)
  const double flops = 2.0 * P.N * P.H * P.W * (double)P.Cout * P.Ctot * 9;
  const long long n_tiles = (long long)P.N * P.tilesY * P.tilesX;
  // UNET_CONV_XCD: channel tiles interleaved per pixel tile (default: up to 4; 1 = channel-tile-major as in round 2)
  {
    const char xv = unet_tuning().conv_xcd;
    const int want = xv == '1' ? 1 : (xv == '2' ? 2 : 4);
    P.co_il = 1;
    while (P.co_il * 2 <= want && P.nCo % (P.co_il * 2) == 0 && blocks % (P.co_il * 2 * 8) == 0) P.co_il *= 2;
  }
  // block-mode statistics: a block stays on one channel tile for whole super-groups and the co_il blocks of a row cover them all
  P.zdiv = (P.stats && (n_tiles * P.co_il) % blocks == 0) ? 1 : 0;
  if (P.stats && stat_parts) *stat_parts = P.zdiv ? blocks / P.co_il : (int)n_tiles;
#ifdef PDMA_STAMPS
  if (!bnbwd) P.bn_mean = (const float*)g_pdma_debug;
#endif
  // (one bracket name per body: the lock-step and ping-pong instantiations of conv3_pdma_body<BN> are one kernel family)
  // algorithmic bytes: input + packed weights + output, each once (+ y of the fused BatchNorm-backward form, + the old
  // values of an accumulating epilogue), bf16
  const double px = (double)P.N * P.H * P.W;
  const double alg_bytes = 2.0 * (px * (P.Ctot + P.Cout * (1.0 + (bnbwd ? 1 : 0) + (P.accumulate ? 1 : 0))) + 9.0 * P.Ctot * P.Cout);
  ProfScope prof(kclass, flops, s, bnbwd ? (BN == 128 ? "conv3_pdma128_bnbwd_kernel" : "conv3_pdma64_bnbwd_kernel")
                                          : (BN == 128 ? "conv3_pdma128_kernel" : "conv3_pdma64_kernel"), alg_bytes);
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), lds_bytes, s, P);
  return unet_check_launch("conv3_pdma_kernel");
}

// ------------------------------------------------------------------------------------------------------
// conv3_ws_kernel: weight-stationary 3x3 convolution for the wide-spatial / narrow-channel layers
// (64 input channels: inc.*, up4.conv.3, their data gradients, the 128-row dgrad of up4.conv.0).
// These layers are HBM-bound (AI ~ 288 FLOP/B at bf16), their whole filter bank is tiny (9*64*Cout bf16),
// and the per-tap weight staging + barrier of the generic kernel dominated their run time.  Here every
// wave keeps ITS 32 output channels x 576 K of weights in REGISTERS (36 MFMA A-fragments = 144 VGPRs,
// loaded once per block straight from global memory) and the block streams pixel tiles: the halo'd patch
// of tile t+1 is prefetched (buffer loads -> registers) while tile t computes and lands in the other LDS
// buffer; one barrier per tile, no weight traffic through LDS at all, 72 MFMAs per wave between barriers.
struct CfgWS {
  static constexpr int WTH = 16, WTW = 16;                      // 256-pixel tiles: halo overhead 1.27x
  static constexpr int HH = WTH + 2, HW = WTW + 2;
  static constexpr int PSTR = 128 + 16;
  static constexpr int RS = (HW * PSTR + 255) / 256 * 256;
  static constexpr int PIECES = HH * (RS / 16);                 // 16-byte pieces of the padded image (pads included)
  static constexpr int NWAVE = 8;                               // 512 threads: 2 (channels) x 4 (pixels)
  static constexpr int NINSTR = (PIECES + 63) / 64;             // 1 KiB LDS-DMA instructions per patch
  static constexpr int NDMA = (NINSTR + NWAVE - 1) / NWAVE;     // per wave per tile (surplus ones hit a dummy KiB)
  static constexpr int A_BYTES = NINSTR * 1024;
  static constexpr int NBUF = 3;                                // patch ring: 2 tiles in flight behind the one computing
  static constexpr int RED_BASE = NBUF * A_BYTES + 1024;        // (+ dummy target of the surplus (all-OOB) DMAs)
  static constexpr int RED_BYTES = 2 * 2 * 8 * 64 * 4;          // [2 tiles][2 statistics][8 half-wave slots][64 channels]
  static constexpr int CT_BASE = RED_BASE + RED_BYTES;          // BatchNorm coefficients of the fused backward mask
  static constexpr int LDS = CT_BASE + 3 * 64 * 4;
  static constexpr int PXT = 2;                                 // 64 pixels per wave
  static constexpr int ROWS = 64;                               // output channels per block
};

// STATS: 0 none; 1 = BatchNorm batch statistics of the stored outputs (sum, sum of squares: the forward of
// conv -> BatchNorm, no separate pass over y); 2 = data gradient with the ReLU mask of the producing layer and its
// BatchNorm-backward sums (sum dz, sum dz * (y - mean); see IgemmParams::bn_y).  A wave cannot afford per-lane running
// sums next to its 144 weight registers, so every tile's 2 x 16 per-lane values are reduced over the 16 lanes of a DPP
// row at once (4 VALU adds each), the four row leaders leave them in an LDS slot, and 128 threads keep the block's
// running total of their (statistic, channel) -- one ordered partial per block: deterministic.
// ST ("stagger"): a tile is two phases with a barrier after each -- M: the DMA issue of the tile two ahead + the 72
// MFMAs (+ the counted wait for the NEXT tile's patch), E: the tile's epilogue (pack, statistics, stores) + the next
// tile's output geometry -- and waves 4-7 run one barrier behind waves 0-3, so on every SIMD one wave's MFMA phase
// covers its partner's VALU/store phase (in lock-step all eight did their epilogues together, then fought over the
// matrix pipe: 40 % MFMA busy).  Waves 0-3 own output channels 0-31, waves 4-7 channels 32-63 (a half's statistics
// stay inside the half).
template <bool ACC, int STATS = 0, bool ST = false>
__global__ __launch_bounds__(512, 1) void conv3_ws_kernel(const IgemmParams P, int tiles_per_block) {
  using C = CfgWS;
  static_assert(!(ACC && STATS), "the gradient fan-in form carries no statistics");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave >> 2;
  const int wco = ST ? grp : (wave & 1), wpx = ST ? (wave & 3) : (wave >> 1);
  const int l31 = lane & 31, hh = lane >> 5;
  const int nCg = P.Cout / C::ROWS;
  // channel groups of one tile range sit on the SAME XCD (b % 8) in adjacent dispatch slots, so the
  // second group finds the patches in that XCD's L2
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int cg = slot % nCg, tr = (slot / nCg) * 8 + xcd;
  const int co_lane = cg * C::ROWS + wco * 32 + l31;
  const int tiles_img = P.tilesX * P.tilesY;
  const int total_tiles = P.N * tiles_img;
  const int t_begin = tr * tiles_per_block;
  const int t_end = min(t_begin + tiles_per_block, total_tiles);
  if (t_begin >= t_end) {
    if (STATS && tid < 128)                       // an empty tile range still owns a partial: zeros
      P.stats[((size_t)tr * 2 + (tid >> 6)) * P.Cout + cg * C::ROWS + (tid & 63)] = 0.f;
    return;
  }
  float* const red = reinterpret_cast<float*>(smem + C::RED_BASE);
  float* const ctab = reinterpret_cast<float*>(smem + C::CT_BASE);      // [scale | shift | mean][64]
  if (STATS == 2 && tid < 192) {
    const float* srcp = tid < 64 ? P.bn_scale : (tid < 128 ? P.bn_shift : P.bn_mean);
    ctab[tid] = srcp[cg * C::ROWS + (tid & 63)];
  }
  float stat_tot = 0.f;

  // ---- this wave's weights -> registers: A fragment (tap, kg) = W[co_lane][tap][16*kg + 8*hh .. +7]
  bf16x8 wreg[36];
  {
    const bf16_t* wp = reinterpret_cast<const bf16_t*>(P.w);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int kg = 0; kg < 4; ++kg)
        wreg[tap * 4 + kg] = *reinterpret_cast<const bf16x8*>(wp + ((size_t)(tap * P.Cout + co_lane)) * P.wK + kg * 16 + hh * 8);
    // retire the weight loads HERE (vmcnt(0)), through the builtin so hipcc's wait bookkeeping sees it:
    // otherwise it places counted waits for them inside the tile loop, which would drain the DMA ring.
    __builtin_amdgcn_s_waitcnt(0x0F70);
  }

  int boff[C::PXT];
#pragma unroll
  for (int pt = 0; pt < C::PXT; ++pt) {
    const int m = wpx * (32 * C::PXT) + pt * 32 + l31;
    boff[pt] = (m >> 4) * C::RS + (m & 15) * C::PSTR + hh * 16;
  }
  constexpr unsigned OOB = 0xFFFFFFF0u;
  const DView S = P.src[0];
  // LDS-DMA staging (buffer_load_dwordx4 ... lds): the padded LDS image is filled LINEARLY, 1 KiB per wave
  // instruction; pad pieces and out-of-image halo pixels use an out-of-range voffset and land as zeros.
  // No staging registers: whole patches are in flight while this tile computes.
  // per lane and instruction, constant over the tiles, two to a register: hy | hx << 5 | piece << 10 | 1 << 14 (0 = pad
  // piece).  A tile then costs a dozen VALU instructions per DMA (the tile's own origin is scalar) where the lane
  // geometry used to be divided out and multiplied up per tile (~300 per tile, fighting the partner wave's epilogue for
  // the SIMD's vector issue: the stamps showed them taking as long as the tile's 72 MFMAs).
  unsigned a_pk[(C::NDMA + 1) / 2];
#pragma unroll
  for (int j = 0; j < C::NDMA; ++j) {
    const int q = (j * C::NWAVE + wave) * 64 + lane;              // instruction index j*NWAVE + wave
    const int hy = q / (C::RS / 16), rem = q - hy * (C::RS / 16);
    const int hx = rem / 9, part = rem - hx * 9;
    const unsigned code = (hy < C::HH && hx < C::HW && part < 8) ? (unsigned)(hy | (hx << 5) | (part << 10) | (1 << 14)) : 0u;
    if (j & 1) a_pk[j >> 1] |= code << 16;
    else a_pk[j >> 1] = code;
  }
  const unsigned img_bytes = (unsigned)S.H * S.W * S.C * 2u;
  typedef __attribute__((address_space(3))) void lds_void;

  // tile coordinates advance by increments (no per-tile divisions): one iterator per consumer
  struct TileIt { int n, ty, tx; };
  auto tile_at = [&](int tile) {
    TileIt it;
    it.n = tile / tiles_img;
    const int r = tile - it.n * tiles_img;
    it.ty = r / P.tilesX;
    it.tx = r - it.ty * P.tilesX;
    return it;
  };
  auto tile_next = [&](TileIt& it) {
    if (++it.tx == P.tilesX) {
      it.tx = 0;
      if (++it.ty == P.tilesY) { it.ty = 0; ++it.n; }
    }
  };
  TileIt dma_it = tile_at(t_begin), geo_it = dma_it;

  auto dma_a = [&](int buf, bool live) {         // the patch of the tile at dma_it (then advance); dead = to the dummy KiB
    const int ym1 = dma_it.ty * C::WTH - 1, xm1 = dma_it.tx * C::WTW - 1;
    const unsigned base = (unsigned)((ym1 * S.W + xm1) * S.C * 2);      // (may wrap below zero: only valid sums are used)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(S.p + (size_t)(live ? dma_it.n : 0) * img_bytes), (short)0, (int)img_bytes, 0x00020000);
#pragma unroll
    for (int j = 0; j < C::NDMA; ++j) {
      unsigned code = (a_pk[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu;
      asm volatile("" : "+v"(code));               // decode per tile: hoisted out of the loop it costs 21 live registers
      const int hy = code & 31, hx = (code >> 5) & 31, part = (code >> 10) & 15;
      const unsigned y = (unsigned)(ym1 + hy), x = (unsigned)(xm1 + hx);
      const bool ok = live && (code >> 14) && y < (unsigned)S.H && x < (unsigned)S.W;
      const unsigned vo = ok ? base + (unsigned)((hy * S.W + hx) * S.C * 2 + part * 16) : OOB;
      const int idx = j * C::NWAVE + wave;                          // wave-uniform
      char* dst = (live && idx < C::NINSTR) ? smem + buf * C::A_BYTES + idx * 1024 : smem + C::NBUF * C::A_BYTES;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, vo, 0, 0, 0);
    }
    tile_next(dma_it);
  };

  // ring of NBUF patches: tile k computes from slot k % NBUF while the DMAs of tiles k+1, k+2 are in flight.
  // Waits are COUNTED.  vmcnt counts loads, DMAs and stores in issue order; per tile every wave issues
  // exactly NDMA DMAs and NST stores (out-of-range ones are buffer ops with an OOB offset: issued, counted,
  // dropped by the range check), so the ops younger than tile j's DMAs are known exactly:
  //   stores(j-2) + DMA(j+1) + stores(j-1)  ->  vmcnt(2*NST + NDMA) retires tile j's patch while the next
  //   patch and 32 stores stay in flight.  Raw s_barrier (a __syncthreads() here would emit vmcnt(0)).
  constexpr int NVIEW = STATS ? 1 : 2;           // the statistics forms write ONE dense destination
  constexpr int NST = 2 * C::PXT * NVIEW;        // stores per wave per tile: 2 sixteen-channel groups x PXT x dst views
  // STATS == 2 adds NY loads of y per tile, issued BEFORE the tile's DMAs (so that waiting for them in the epilogue
  // leaves those DMAs in flight); by the next tile's wait they are long complete but still count as issued-after
  constexpr int NY = STATS == 2 ? 2 * C::PXT : 0;
  static_assert(2 * NST + C::NDMA + NY <= 63, "vmcnt range");
  // ---- output geometry of a tile (buffer stores: an OOB offset = dropped, so the op count is static)
  // A lane of the 32x32 accumulator owns rows 8g+4hh..+3 of pixel l31; v_permlane32_swap trades the g-odd run
  // of the lower half-wave for the g-even run of the upper one, so every lane ends up with 8 CONSECUTIVE
  // channels (rows 16gp + 8hh ..+7) and writes 16 bytes: half as many store instructions, 32-byte segments.
  __amdgpu_buffer_rsrc_t drs[2];
  unsigned ovo[C::PXT][2][NVIEW];
  int n_img = 0;
  auto geometry = [&]() {                        // of the tile at geo_it (then advance)
    const int n = geo_it.n;
    const int ty0 = geo_it.ty * C::WTH, tx0 = geo_it.tx * C::WTW;
    tile_next(geo_it);
    n_img = n;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const DViewW D = P.dst[q];
      const unsigned dimg = (unsigned)D.H * D.W * D.C * 2u;
      drs[q] = __builtin_amdgcn_make_buffer_rsrc((void*)(D.p ? D.p + (size_t)n * dimg : P.dst[0].p), (short)0,
                                                 D.p ? (int)dimg : 0, 0x00020000);
    }
#pragma unroll
    for (int pt = 0; pt < C::PXT; ++pt) {
      const int m = wpx * (32 * C::PXT) + pt * 32 + l31;
      const int fy = ty0 + (m >> 4), fx = tx0 + (m & 15);
      const bool pix_ok = fy < P.H && fx < P.W;
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        const int co = cg * C::ROWS + wco * 32 + 16 * gp + 8 * hh;
#pragma unroll
        for (int q = 0; q < NVIEW; ++q) {             // one store per destination view; the other one is OOB
          const DViewW D = P.dst[q];
          const int cq = q == 0 ? co : co - P.dst_split;
          const bool mine = (q == 0) == (co < P.dst_split);
          const int y = fy - D.oy, x = fx - D.ox;
          const bool ok = mine && pix_ok && D.p && y >= 0 && y < D.H && x >= 0 && x < D.W;
          ovo[pt][gp][q] = ok ? (unsigned)(((y * D.W + x) * D.C + cq) * 2) : OOB;
        }
      }
    }
  };
  // statistics: the slots of tile kk (this half's four waves, or all eight in lock-step) -> the thread's running total
  const int st_u = ST ? (tid & 255) : tid;
  const bool st_on = STATS && st_u < (ST ? 64 : 128);
  const int st_q = ST ? (st_u >> 5) : (st_u >> 6), st_c = ST ? grp * 32 + (st_u & 31) : (st_u & 63);
  auto take_slots = [&](int kk) {
    if (st_on) {
      const float* rp = red + (kk & 1) * 1024 + st_q * 512 + st_c;
#pragma unroll
      for (int sl = 0; sl < 8; ++sl) stat_tot += rp[sl * 64];
    }
  };

#pragma unroll
  for (int d = 0; d < C::NBUF - 1; ++d)
    if (ST || STATS == 2 || t_begin + d < t_end) dma_a(d, t_begin + d < t_end);
  if constexpr (ST) {
    geometry();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NDMA) : "memory");          // the first patch; the second one flies
    __builtin_amdgcn_s_barrier();
    if (grp) __builtin_amdgcn_s_barrier();                                  // the stagger
  }
#ifdef PDMA_STAMPS
  unsigned long long ws_st[5] = {0, 0, 0, 0, 0}, ws_prev = __builtin_amdgcn_s_memtime(), ws_dma_sum = 0;
  const unsigned long long ws_t0 = ws_prev, ws_r0 = __builtin_amdgcn_s_memrealtime();
#define WS_STAMP(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ws_st[i] += t_ - ws_prev; ws_prev = t_; }
#else
#define WS_STAMP(i)
#endif
  for (int tile = t_begin; tile < t_end; ++tile) {
    const int k = tile - t_begin;
    const int cur = k % C::NBUF;
    const bool next_in_flight = tile + 1 < t_end;
    if constexpr (!ST) {
    if (k >= 2) {
      if (next_in_flight) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NST + C::NDMA + NY) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NST) : "memory");
    } else if (k == 1) {
      if (next_in_flight) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST + C::NDMA + NY) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
    } else {
      if (next_in_flight) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NDMA) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (k >= 1) take_slots(k - 1);                 // the previous tile's slots -> this thread's running total
    geometry();
    }
    const int n = n_img;
    (void)n;
    // STATS == 2: this tile's y values (same offsets as the stores: dst[0] is dense and frame-sized) are requested
    // FIRST, then the DMAs of the tile two ahead
    u32x4 yv[STATS == 2 ? C::PXT : 1][2];
    if constexpr (STATS == 2) {
      const unsigned dimg = (unsigned)P.dst[0].H * P.dst[0].W * P.dst[0].C * 2u;
      const __amdgpu_buffer_rsrc_t yrs =
          __builtin_amdgcn_make_buffer_rsrc((void*)(P.bn_y + (size_t)n * dimg), (short)0, (int)dimg, 0x00020000);
#pragma unroll
      for (int pt = 0; pt < C::PXT; ++pt)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp)     // inline asm + the hand-counted wait below: hipcc does not count LDS-DMA
                                           // instructions, its own wait for a builtin load here would drain the DMA ring
          asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(yv[pt][gp]) : "v"(ovo[pt][gp][0]), "s"(yrs) : "memory");
    }
    // (the statistics-2 form always issues its NDMA instructions -- dead ones to the dummy KiB -- so that one wait form
    //  covers every tile)
    if (ST || STATS == 2 || tile + C::NBUF - 1 < t_end) dma_a((k + C::NBUF - 1) % C::NBUF, tile + C::NBUF - 1 < t_end);
    // gradient fan-in (ACC): the old values are fetched NOW, behind the tile's 72 MFMAs (one load per output
    // run, from whichever view owns it and has its accumulate bit set; everything else reads as 0)
    u32x4 oldv[ACC ? C::PXT : 1][2];
    if constexpr (ACC) {
#pragma unroll
      for (int pt = 0; pt < C::PXT; ++pt)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          const bool second = ovo[pt][gp][0] == OOB;
          const bool want = (P.accumulate >> (second ? 1 : 0)) & 1;
          const unsigned vo = want ? (second ? ovo[pt][gp][NVIEW - 1] : ovo[pt][gp][0]) : OOB;
          oldv[pt][gp] = second ? __builtin_amdgcn_raw_buffer_load_b128(drs[1], vo, 0, 0)
                                : __builtin_amdgcn_raw_buffer_load_b128(drs[0], vo, 0, 0);
        }
    }

    f32x16 acc[C::PXT];
#pragma unroll
    for (int pt = 0; pt < C::PXT; ++pt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[pt][r] = 0.f;
    const char* pb = smem + cur * C::A_BYTES;
    // 36 (tap, 16-channel group) steps of PXT MFMAs; the pixel fragments of step i+2 are requested before the MFMAs
    // of step i and pinned there (left alone, hipcc requests them one MFMA ahead: the LDS round trip showed)
#ifndef WS_DEPTH
#define WS_DEPTH 2
#endif
    constexpr int DEPTH = STATS == 2 ? 1 : WS_DEPTH;      // (the masked-gradient form needs the registers for its y values)
    auto frag = [&](int i, int pt) {
      const int tap = i >> 2, kg = i & 3;
      return *reinterpret_cast<const bf16x8*>(pb + boff[pt] + (tap / 3) * C::RS + (tap % 3) * C::PSTR + kg * 32);
    };
    bf16x8 ring[DEPTH + 1][C::PXT];
#ifdef PDMA_STAMPS
    unsigned long long ws_dma = 0;
    if (ST) ws_dma = __builtin_amdgcn_s_memtime() - ws_prev;
#endif
#pragma unroll
    for (int i = 0; i < DEPTH; ++i)
#pragma unroll
      for (int pt = 0; pt < C::PXT; ++pt) ring[i][pt] = frag(i, pt);
#pragma unroll
    for (int i = 0; i < 36; ++i) {
      if (i + DEPTH < 36) {
#pragma unroll
        for (int pt = 0; pt < C::PXT; ++pt) ring[(i + DEPTH) % (DEPTH + 1)][pt] = frag(i + DEPTH, pt);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int pt = 0; pt < C::PXT; ++pt)
        acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[i], ring[i % (DEPTH + 1)][pt], acc[pt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }

    if constexpr (STATS == 2) {
      static_assert(C::PXT == 2, "the wait statement names 4 destinations");
      // the y loads are older than this tile's NDMA instructions: exactly those stay in flight
      asm volatile("s_waitcnt vmcnt(%4)" : "+v"(yv[0][0]), "+v"(yv[0][1]), "+v"(yv[1][0]), "+v"(yv[1][1]) : "n"(C::NDMA));
    }
    if constexpr (ST) {
#ifdef PDMA_STAMPS
      ws_dma_sum += ws_dma;
#endif
      // end of the M phase: the NEXT tile's patch (issued one tile ago) has landed; younger than it: the previous
      // tile's stores, this tile's y loads and the patch just issued
      WS_STAMP(0)
      if (k >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST + NY + C::NDMA) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NY + C::NDMA) : "memory");
      WS_STAMP(1)
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      WS_STAMP(2)
    }
    // ---- epilogue for this tile: exactly NST buffer stores per wave (OOB offset = dropped).  Per 16-channel group gp
    // the two 4-row runs of a lane (t = 0: rows 16gp+4hh.., t = 1: +8) are finished one after the other so that only
    // one run's coefficients and sums are live (the statistics forms sit at the 256-register limit).
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
      u32x2 pk[C::PXT][2];                         // packed bf16x4 results: [pixel tile][run]
      u32x2 inp[C::PXT][2];                        // ACC: old values / STATS 2: y, in the accumulator's lane layout
      if constexpr (ACC || STATS == 2) {
#pragma unroll
        for (int pt = 0; pt < C::PXT; ++pt) {
          const u32x4 o = ACC ? oldv[pt][gp] : yv[pt][gp];
          const auto o0 = __builtin_amdgcn_permlane32_swap(o[0], o[2], false, false);
          const auto o1 = __builtin_amdgcn_permlane32_swap(o[1], o[3], false, false);
          inp[pt][0] = u32x2{o0[0], o1[0]};
          inp[pt][1] = u32x2{o0[1], o1[1]};
        }
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
        f32x4 csc, csh, cmu;
        if constexpr (STATS == 2) {
          const int cb = wco * 32 + 16 * gp + 4 * hh + 8 * t;
          csc = *reinterpret_cast<const f32x4*>(ctab + cb);
          csh = *reinterpret_cast<const f32x4*>(ctab + 64 + cb);
          cmu = *reinterpret_cast<const f32x4*>(ctab + 128 + cb);
        }
#pragma unroll
        for (int pt = 0; pt < C::PXT; ++pt) {
          float f[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) f[j] = acc[pt][8 * gp + 4 * t + j];
          bf16x4 x;
          if constexpr (ACC) {
            const bf16x4 o = __builtin_bit_cast(bf16x4, inp[pt][t]);
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = (bf16_t)(f[j] + (float)o[j]);      // add in fp32, round once
          } else if constexpr (STATS == 2) {
            const bf16x4 yq = __builtin_bit_cast(bf16x4, inp[pt][t]);
            const bool ok = ovo[pt][gp][0] != OOB;                                // a tile pixel outside the frame: no sums
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float yy = (float)yq[j];
              x[j] = (bf16_t)((ok && fmaf(yy, csc[j], csh[j]) > 0.f) ? f[j] : 0.f);
              const float q = (float)x[j];                                        // dz as stored
              s0[j] += q;
              s1[j] = fmaf(q, yy - cmu[j], s1[j]);
            }
          } else {
            if (P.bias) {                            // inference: BatchNorm shift (+ ReLU) of the folded layer
              const float* bp = P.bias + cg * C::ROWS + wco * 32 + 16 * gp + 4 * hh + 8 * t;
#pragma unroll
              for (int j = 0; j < 4; ++j) f[j] += bp[j];
            }
            if (P.relu) {
#pragma unroll
              for (int j = 0; j < 4; ++j) f[j] = fmaxf(f[j], 0.f);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = (bf16_t)f[j];
            if constexpr (STATS == 1) {
              const bool ok = ovo[pt][gp][0] != OOB;
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const float q = ok ? (float)x[j] : 0.f;                           // the value as stored
                s0[j] += q;
                s1[j] = fmaf(q, q, s1[j]);
              }
            }
          }
          pk[pt][t] = __builtin_bit_cast(u32x2, x);
        }
        if constexpr (STATS != 0) {
          float* rw = red + (k & 1) * 1024 + (wpx * 2 + ((lane >> 4) & 1)) * 64 + wco * 32 + 16 * gp + 4 * hh + 8 * t;
          float rv[8];
#pragma unroll
          for (int j = 0; j < 4; ++j) { rv[j] = s0[j]; rv[4 + j] = s1[j]; }
          row16_sum_n(rv);
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if ((lane & 15) == 0) { rw[j] = rv[j]; rw[512 + j] = rv[4 + j]; }
        }
      }
#pragma unroll
      for (int pt = 0; pt < C::PXT; ++pt) {
        const auto s0w = __builtin_amdgcn_permlane32_swap(pk[pt][0][0], pk[pt][1][0], false, false);
        const auto s1w = __builtin_amdgcn_permlane32_swap(pk[pt][0][1], pk[pt][1][1], false, false);
        const u32x4 bits = u32x4{s0w[0], s1w[0], s0w[1], s1w[1]};
#pragma unroll
        for (int q = 0; q < NVIEW; ++q) __builtin_amdgcn_raw_buffer_store_b128(bits, drs[q], ovo[pt][gp][q], 0, 0);
      }
    }
    if constexpr (ST) {
      // rest of the E phase: the previous tile's slots (written a barrier pair ago), the next tile's geometry
      if (k >= 1) take_slots(k - 1);
      if (next_in_flight) geometry();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // this tile's slot writes, before the half's barrier
      WS_STAMP(3)
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      WS_STAMP(4)
    }
  }
#ifdef PDMA_STAMPS
  if (ST && STATS != 2 && P.bn_mean && lane == 0) {
    unsigned long long* o = (unsigned long long*)P.bn_mean + ((size_t)(blockIdx.x & 255) * 8 + wave) * 8;
    for (int i = 0; i < 5; ++i) o[i] = ws_st[i];
    o[5] = (unsigned long long)(t_end - t_begin);
    o[7] = ws_dma_sum;
    o[6] = ((__builtin_amdgcn_s_memtime() - ws_t0) << 20) / (__builtin_amdgcn_s_memrealtime() - ws_r0 + 1);
  }
#endif
  if constexpr (ST) { if (!grp) __builtin_amdgcn_s_barrier(); }    // pairs with the stagger barrier of waves 4-7
  if constexpr (STATS != 0) {
    // the last tile's slots, then ONE partial per block
    if constexpr (!ST) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    take_slots(t_end - 1 - t_begin);
    if (st_on) P.stats[((size_t)tr * 2 + st_q) * P.Cout + cg * C::ROWS + st_c] = stat_tot;
  }
  if constexpr (ST || STATS == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the dummy DMAs before the wave ends
}

// ------------------------------------------------------------------------------------------------------
// conv3_ws16_kernel: the weight-stationary kernel on v_mfma_f32_16x16x32_bf16 (round 2).  Same work split as
// conv3_ws_kernel (8 waves = 2 channel halves x 4 pixel rows-of-4, a wave keeps 32 output channels x 576 K of weights in
// 144 VGPRs, 16x16-pixel tiles stream through a 3-slot LDS-DMA ring, counted vmcnt, one barrier per tile), but
//  * 8 accumulator tiles (2 channel x 4 pixel) of 16x16 per wave instead of 2 of 32x32: eight independent MFMA chains
//    (the 32x32x16 form had two, each MFMA waiting for the one two back: 64-cycle latency at a 32-cycle issue rate) and
//    the shape the chip clocks higher on (MI355X_MICROARCH "DVFS give-back" item 7): a timing-only swap of the
//    instruction measured 208 -> 154 us on 64 -> 64 @256x256;
//  * the LDS patch is UNPADDED (18 x 18 pixels x 128 B, 41 instead of 50 one-KiB DMA instructions per tile), 16-byte
//    pieces XOR-swizzled by (pixel index & 7) through the DMA's per-lane source address: a 16x16x32 pixel fragment
//    (lanes = 16 consecutive pixels x 4 piece columns) is conflict-free for every start pixel; the swizzle term of a
//    read depends on (2 * (pixel row + tap row) + tap column) & 7 only, so 8 per-lane base addresses + immediates cover
//    all 72 reads of a K-step pair;
//  * the epilogue is conv3_pdma's (v_permlane16_swap -> 16-byte stores; the DPP row = the 16 pixels of a tile row).
struct CfgWS16 {
  static constexpr int HH = 18, HW = 18, NPIXP = HH * HW;
  static constexpr int PIECES = NPIXP * 8;
  static constexpr int NWAVE = 8;
  static constexpr int NINSTR = (PIECES + 63) / 64;             // 41
  static constexpr int NDMA = (NINSTR + NWAVE - 1) / NWAVE;     // 6 per wave per tile (surplus ones hit a dummy KiB)
  static constexpr int A_BYTES = NINSTR * 1024;
  static constexpr int NBUF = 3;
  static constexpr int DUMMY = NBUF * A_BYTES;
  static constexpr int RED_BASE = DUMMY + 1024;
  static constexpr int RED_BYTES = 2 * 2 * 4 * 64 * 4;          // [2 tiles][2 statistics][4 pixel-wave slots][64 channels]
  static constexpr int CT_BASE = RED_BASE + RED_BYTES;
  static constexpr int LDS = CT_BASE + 3 * 64 * 4;
  static constexpr int ROWS = 64;
};

template <bool ACC, int STATS = 0>
__global__ __launch_bounds__(512, 1) void conv3_ws16_kernel(const IgemmParams P, int tiles_per_block) {
  using C = CfgWS16;
  static_assert(!(ACC && STATS), "the gradient fan-in form carries no statistics");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void lds_void;
  constexpr unsigned OOB = 0xFFFFFFF0u;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wco = wave & 1, wpx = wave >> 1;
  const int l15 = lane & 15, kb = lane >> 4;
  const int nCg = P.Cout / C::ROWS;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int cg = slot % nCg, tr = (slot / nCg) * 8 + xcd;
  const int tiles_img = P.tilesX * P.tilesY;
  const int total_tiles = P.N * tiles_img;
  const int t_begin = tr * tiles_per_block;
  const int t_end = min(t_begin + tiles_per_block, total_tiles);
  if (t_begin >= t_end) {
    if (STATS && tid < 128)                       // an empty tile range still owns a partial: zeros
      P.stats[((size_t)tr * 2 + (tid >> 6)) * P.Cout + cg * C::ROWS + (tid & 63)] = 0.f;
    return;
  }
  float* const red = reinterpret_cast<float*>(smem + C::RED_BASE);
  float* const ctab = reinterpret_cast<float*>(smem + C::CT_BASE);      // [scale | shift | mean][64]
  if (STATS == 2 && tid < 192) {
    const float* srcp = tid < 64 ? P.bn_scale : (tid < 128 ? P.bn_shift : P.bn_mean);
    ctab[tid] = srcp[cg * C::ROWS + (tid & 63)];
  }
  float stat_tot = 0.f;
  const int ch0 = cg * C::ROWS + wco * 32;          // first output channel of this wave
  // (the gradient fan-in form keeps every wave's DMAs in front: its old-value loads are builtin loads, whose
  //  compiler-placed wait would drain DMAs issued behind them; the fused BatchNorm-backward form too: its y loads would
  //  need a vmcnt(0) in front of the late burst and the extra code path costs it 6 more spills -- measured 605 -> 828 us/step)
  const bool late = !ACC && STATS != 2 && P.ws_stagger && __builtin_amdgcn_readfirstlane(wave) < 4;
  // ... and the other half (waves 4-7) keeps a tile's packed results in registers across the barrier and stores them at
  // the top of the NEXT tile, behind its DMA burst: every vector-memory instruction of a wave is then issued while its
  // SIMD partner runs MFMAs (a store or DMA that waits for a queue slot stalls the wave that issues it, and at the old
  // tile end both waves of a SIMD stalled together).  UNET_WS_STG=1: the DMA placement without the deferred stores.
  const bool defer = !ACC && STATS != 2 && P.ws_stagger >= 2 && __builtin_amdgcn_readfirstlane(wave) >= 4;

  // ---- this wave's weights -> registers: A fragment (tile ct, tap, ks) = W[ch0 + 16ct + l15][tap][32ks + 8kb .. +7]
  bf16x8 wreg[2][18];
  {
    const bf16_t* wp = reinterpret_cast<const bf16_t*>(P.w);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          wreg[ct][tap * 2 + ks] = *reinterpret_cast<const bf16x8*>(
              wp + ((size_t)(tap * P.Cout + ch0 + ct * 16 + l15)) * P.wK + ks * 32 + kb * 8);
    __builtin_amdgcn_s_waitcnt(0x0F70);          // retire the weight loads here (see conv3_ws_kernel)
  }

  // ---- pixel-fragment addresses: pixel p = (4 wpx + pt + r) * 18 + l15 + c of the patch, piece (4ks + kb) ^ (p & 7).
  // (p & 7) = (b + l15) & 7 with b = (2 (pt + r) + c) & 7 a compile-time constant of the read (72 wpx = 0 mod 8), so
  // vb[b] holds the lane part for K-step 0; K-step 1 flips bit 6; everything else is an immediate.  vb[] also carries
  // the byte offset of the ring slot being read and is stepped in place from tile to tile.
  unsigned vb[8];
#pragma unroll
  for (int b = 0; b < 8; ++b) vb[b] = (unsigned)((wpx * 4 * C::HW + l15) * 128 + ((kb ^ ((b + l15) & 7)) << 4));
  const DView S = P.src[0];
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  // DMA lane offsets relative to the patch origin (tile-invariant): lane q of instruction j fetches piece pos ^ (pp & 7)
  // of patch pixel pp = q >> 3 (beyond the patch: dropped).  The tile enters through the descriptor's base address
  // (scalar arithmetic), so an interior tile costs no vector instruction per DMA; a tile on the frame's edge checks its
  // halo pixels per lane.
  unsigned a_rel[C::NDMA];
#pragma unroll
  for (int j = 0; j < C::NDMA; ++j) {
    const int q = (j * C::NWAVE + wave) * 64 + lane;
    const int pp = q >> 3, pos = q & 7;
    const int hy = pp / C::HW, hx = pp - hy * C::HW;
    a_rel[j] = pp < C::NPIXP ? (unsigned)((hy * S.W + hx) * S.C * 2 + ((pos ^ (pp & 7)) << 4)) : OOB;
  }
  const unsigned img_bytes = (unsigned)S.H * S.W * S.C * 2u;

  struct TileIt { int n, ty, tx; };
  auto tile_at = [&](int tile) {
    TileIt it;
    it.n = tile / tiles_img;
    const int r = tile - it.n * tiles_img;
    it.ty = r / P.tilesX;
    it.tx = r - it.ty * P.tilesX;
    return it;
  };
  auto tile_next = [&](TileIt& it) {
    if (++it.tx == P.tilesX) {
      it.tx = 0;
      if (++it.ty == P.tilesY) { it.ty = 0; ++it.n; }
    }
  };
  TileIt dma_it = tile_at(t_begin), geo_it = dma_it;

  auto dma_a = [&](int buf, bool live) {         // the patch of the tile at dma_it (then advance); dead = to the dummy KiB
    const int ym1 = dma_it.ty * 16 - 1, xm1 = dma_it.tx * 16 - 1;
    // base = the patch origin (it may lie in front of the image: only lanes of pixels inside the frame carry an offset
    // below num_records; a valid lane's offset stays below 18 rows of the frame)
    const long long org = ((long long)ym1 * S.W + xm1) * (S.C * 2);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(S.p + (long long)(live ? dma_it.n : 0) * img_bytes + org), (short)0, 0x7FFFFFF0, 0x00020000);
    const bool inner = live && ym1 >= 0 && xm1 >= 0 && ym1 + C::HH <= S.H && xm1 + C::HW <= S.W;
    if (inner) {
#pragma unroll
      for (int j = 0; j < C::NDMA; ++j) {
        const int idx = j * C::NWAVE + wave_s;
        char* dst = idx < C::NINSTR ? smem + buf * C::A_BYTES + idx * 1024 : smem + C::DUMMY;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, a_rel[j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int j = 0; j < C::NDMA; ++j) {
        int q = (j * C::NWAVE + wave_s) * 64 + lane;
        asm volatile("" : "+v"(q));                 // (per tile: hoisted out of the loop it costs live registers)
        const int pp = q >> 3;
        const int hy = pp / C::HW, hx = pp - hy * C::HW;
        const unsigned y = (unsigned)(ym1 + hy), x = (unsigned)(xm1 + hx);
        const bool ok = live && y < (unsigned)S.H && x < (unsigned)S.W;
        const unsigned vo = ok ? a_rel[j] : OOB;
        const int idx = j * C::NWAVE + wave_s;
        char* dst = (live && idx < C::NINSTR) ? smem + buf * C::A_BYTES + idx * 1024 : smem + C::DUMMY;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, vo, 0, 0, 0);
      }
    }
    tile_next(dma_it);
  };

  constexpr int NVIEW = STATS ? 1 : 2;           // the statistics forms write ONE dense destination
  constexpr int NST = 4 * NVIEW;                 // stores per wave per tile: 4 pixel rows x dst views
  constexpr int NY = STATS == 2 ? 4 : 0;         // y loads per tile (inline asm, hand-counted)
  static_assert(2 * NST + C::NDMA + NY <= 63, "vmcnt range");
  constexpr bool ALWAYS = STATS == 2;            // that form always issues its NDMA instructions: one wait form

  // Dense frames only (the launcher sends everything else to conv3_ws_kernel): whole 16x16 tiles, every destination view
  // covers the frame at offset 0.  A lane's offset inside a tile never changes (ovb: pixel row 0 of its four, per view;
  // rows 1-3 through the scalar offset operand, which the range check ignores: a lane that does not store carries an
  // out-of-range offset) and the tile enters through the descriptors' base addresses: scalar arithmetic only, where the
  // general lane geometry was ~200 vector instructions per tile.
  __amdgpu_buffer_rsrc_t drs[2];
  unsigned ovb[NVIEW];
  u32x4 pend[4];                                 // deferred stores: the packed results of the previous tile
#pragma unroll
  for (int pt = 0; pt < 4; ++pt) pend[pt] = u32x4{0u, 0u, 0u, 0u};
  {
    const int co = ch0 + (kb & 1) * 16 + (kb >> 1) * 8;       // after the swap a lane holds 8 consecutive channels: tile (kb & 1), channels 8 (kb >> 1) .. + 7
#pragma unroll
    for (int q = 0; q < NVIEW; ++q) {
      const DViewW D = P.dst[q];
      const int cq = q == 0 ? co : co - P.dst_split;
      const bool mine = (q == 0) == (co < P.dst_split);
      ovb[q] = mine ? (unsigned)(((wpx * 4 * D.W + l15) * D.C + cq) * 2) : OOB;     // (an absent second view arrives as a copy of the first: no lane is its)
    }
  }
  // (tile 0's deferred group: NST stores against empty descriptors -- dropped, same vmcnt arithmetic)
#pragma unroll
  for (int q = 0; q < 2; ++q) drs[q] = __builtin_amdgcn_make_buffer_rsrc((void*)P.dst[0].p, (short)0, 0, 0x00020000);
  const unsigned rowb[2] = {(unsigned)(P.dst[0].W * P.dst[0].C * 2), (unsigned)(P.dst[1].W * P.dst[1].C * 2)};   // bytes per pixel row
  int n_img = 0;
  unsigned soff0 = 0;                            // byte offset of the tile in view 0 (the y loads add it too)
  auto geometry = [&]() {                        // of the tile at geo_it (then advance)
    const int n = geo_it.n;
    const int ty0 = geo_it.ty * 16, tx0 = geo_it.tx * 16;
    tile_next(geo_it);
    n_img = n;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const DViewW D = P.dst[q];
      const unsigned dimg = (unsigned)D.H * D.W * D.C * 2u;
      const unsigned so = (unsigned)((ty0 * D.W + tx0) * D.C * 2);
      if (q == 0) soff0 = so;
      drs[q] = __builtin_amdgcn_make_buffer_rsrc((void*)(D.p + (size_t)n * dimg + so), (short)0, (int)(dimg - so), 0x00020000);
    }
  };
  auto take_slots = [&](int kk) {                // the four pixel-wave slots of tile kk -> this thread's running total
    if (STATS && tid < 128) {
      const float* rp = red + (kk & 1) * 512 + (tid >> 6) * 256 + (tid & 63);
#pragma unroll
      for (int sl = 0; sl < 4; ++sl) stat_tot += rp[sl * 64];
    }
  };

#pragma unroll
  for (int d = 0; d < C::NBUF - 1; ++d)
    if (ALWAYS || t_begin + d < t_end) dma_a(d, t_begin + d < t_end);
  int cur = 0;
#ifdef PDMA_STAMPS
  unsigned long long w6_st[6] = {0, 0, 0, 0, 0, 0}, w6_prev = __builtin_amdgcn_s_memtime();
  const unsigned long long w6_t0 = w6_prev, w6_r0 = __builtin_amdgcn_s_memrealtime();
#define W6_STAMP(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); w6_st[i] += t_ - w6_prev; w6_prev = t_; __builtin_amdgcn_sched_barrier(0); }
#else
#define W6_STAMP(i)
#endif
  for (int tile = t_begin; tile < t_end; ++tile) {
    const int k = tile - t_begin;
    const bool next_in_flight = ALWAYS || tile + 1 < t_end;
    // vmcnt counts loads, DMAs and stores in issue order: younger than tile k's patch are stores(k-2), y(k-1),
    // DMA(k+1), stores(k-1)  (see conv3_ws_kernel)
    if (k >= 2) {
      if (next_in_flight) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NST + C::NDMA + NY) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NST) : "memory");
    } else if (k == 1) {
      if (next_in_flight) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST + C::NDMA + NY) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
    } else {
      if (next_in_flight) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NDMA) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    W6_STAMP(0)
    __builtin_amdgcn_s_barrier();
    W6_STAMP(1)
    if (k >= 1) take_slots(k - 1);
    constexpr bool RING2 = !ACC && STATS != 2;
    if constexpr (RING2) {
      // per-wave order of vector-memory operations in a tile: DMA(k + 2), then ONE group of NST stores -- tile k's own
      // at its end (waves 0-3 and the lock-step form) or tile k - 1's here (waves 4-7): the counted waits above hold for both
      if (!late) {
        if (tile + C::NBUF - 1 < t_end) dma_a((k + C::NBUF - 1) % C::NBUF, true);
      }
      if (defer) {
#pragma unroll
        for (int pt = 0; pt < 4; ++pt)
#pragma unroll
          for (int q = 0; q < NVIEW; ++q) __builtin_amdgcn_raw_buffer_store_b128(pend[pt], drs[q], ovb[q], pt * rowb[q], 0);
      }
    }
    geometry();
    const int n = n_img;
    (void)n;
    u32x4 yv[STATS == 2 ? 4 : 1];
    if constexpr (STATS == 2) {
      const unsigned dimg = (unsigned)P.dst[0].H * P.dst[0].W * P.dst[0].C * 2u;
      const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(P.bn_y + (size_t)n * dimg + soff0), (short)0, (int)(dimg - soff0), 0x00020000);
#pragma unroll
      for (int pt = 0; pt < 4; ++pt)             // inline asm + hand-counted wait (hipcc does not count LDS-DMAs)
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(yv[pt]) : "v"(ovb[0]), "s"(yrs), "s"(pt * rowb[0]) : "memory");
    }
    // The patch DMAs of tile k + 2.  A wave inside its burst of six one-KiB issues feeds no MFMAs, and with all eight
    // waves bursting behind the barrier the matrix pipe idles for the whole burst (r02 stamps: the burst costs as much
    // as the tile's 72 MFMAs).  The two waves of a SIMD (w, w + 4) therefore issue at opposite ends of the tile: waves
    // 4-7 here, waves 0-3 -- the older ones, which win the SIMD's issue arbitration and so should compute first -- behind
    // their MFMAs, still in front of the tile's stores (the vmcnt bookkeeping above counts the same operations either way).
    if constexpr (!RING2) {
      if (ALWAYS || tile + C::NBUF - 1 < t_end) dma_a((k + C::NBUF - 1) % C::NBUF, tile + C::NBUF - 1 < t_end);
    }
    u32x4 oldv[ACC ? 4 : 1];
    if constexpr (ACC) {                          // gradient fan-in: the old values, behind the tile's MFMAs
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) {
        const bool second = ovb[0] == OOB;
        const bool want = (P.accumulate >> (second ? 1 : 0)) & 1;
        const unsigned vo = want ? (second ? ovb[NVIEW - 1] : ovb[0]) : OOB;
        oldv[pt] = second ? __builtin_amdgcn_raw_buffer_load_b128(drs[1], vo, pt * rowb[1], 0)
                          : __builtin_amdgcn_raw_buffer_load_b128(drs[0], vo, pt * rowb[0], 0);
      }
    }

    W6_STAMP(2)
    f32x4 acc[2][4];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) acc[ct][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // 18 K-steps (tap, 32-channel half) of 2 x 4 MFMAs; the four pixel fragments of step i + 1 are requested before the
    // MFMAs of step i and pinned there
    auto frag = [&](int i, int pt) {
      const int tap = i >> 1, ks = i & 1, r = tap / 3, c = tap % 3;
      const int b = (2 * (pt + r) + c) & 7;
      const unsigned a = (ks ? vb[b] ^ 64u : vb[b]);
      return *reinterpret_cast<const bf16x8*>(smem + a + ((pt + r) * C::HW + c) * 128);
    };
    // (the forms that hold y / old values across the loop have 16 registers fewer: ONE fragment set, each fragment
    //  re-requested for the next step right behind the two MFMAs that read it -- six MFMAs of cover)
    bf16x8 ring[RING2 ? 2 : 1][4];
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) ring[0][pt] = frag(0, pt);
#ifdef WS16_NO_MFMA           // diagnostic build: the tile's DMAs / stores / barriers without its MFMAs and fragment reads
    if (false)
#endif
#pragma unroll
    for (int i = 0; i < 18; ++i) {
      if constexpr (RING2) {
        if (i + 1 < 18) {
#pragma unroll
          for (int pt = 0; pt < 4; ++pt) ring[(i + 1) & 1][pt] = frag(i + 1, pt);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int pt = 0; pt < 4; ++pt)
            acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[ct][i], ring[i & 1][pt], acc[ct][pt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      } else {
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
          __builtin_amdgcn_sched_barrier(0);
          acc[0][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[0][i], ring[0][pt], acc[0][pt], 0, 0, 0);
          acc[1][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[1][i], ring[0][pt], acc[1][pt], 0, 0, 0);
          if (i + 1 < 18) ring[0][pt] = frag(i + 1, pt);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    W6_STAMP(3)
    // the ring slot of the next tile
    {
      const int nxt = (cur + 1) % C::NBUF;
      const unsigned delta = (unsigned)((nxt - cur) * C::A_BYTES);
#pragma unroll
      for (int b = 0; b < 8; ++b) vb[b] += delta;
      cur = nxt;
    }

    if constexpr (STATS == 2) {                   // the y loads are older than this tile's NDMA instructions ...
      if (late) asm volatile("s_waitcnt vmcnt(0)" : "+v"(yv[0]), "+v"(yv[1]), "+v"(yv[2]), "+v"(yv[3]));   // ... not issued yet
      else asm volatile("s_waitcnt vmcnt(%4)" : "+v"(yv[0]), "+v"(yv[1]), "+v"(yv[2]), "+v"(yv[3]) : "n"(C::NDMA));
    }
    if (late) {
      if (ALWAYS || tile + C::NBUF - 1 < t_end) dma_a((k + C::NBUF - 1) % C::NBUF, tile + C::NBUF - 1 < t_end);
    }

    W6_STAMP(4)
    // ---- epilogue: D of 16x16x32: column = lane & 15 (pixel), rows 4 kb + j (channel of the 16-tile).  Exactly NST
    // buffer stores per wave (an OOB offset = dropped).
    float sa[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f}, qa[4] = {0.f, 0.f, 0.f, 0.f}, qb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
      float va[4], vv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { va[j] = acc[0][pt][j]; vv[j] = acc[1][pt][j]; }
      bf16x4 ra, rb;
      if constexpr (ACC) {
        const u32x4 o = oldv[pt];                  // stored layout -> the accumulator's (the exchange is an involution)
        const auto o0 = __builtin_amdgcn_permlane16_swap(o[0], o[2], false, false);
        const auto o1 = __builtin_amdgcn_permlane16_swap(o[1], o[3], false, false);
        const bf16x4 oa = __builtin_bit_cast(bf16x4, u32x2{o0[0], o1[0]});
        const bf16x4 ob = __builtin_bit_cast(bf16x4, u32x2{o0[1], o1[1]});
#pragma unroll
        for (int j = 0; j < 4; ++j) { ra[j] = (bf16_t)(va[j] + (float)oa[j]); rb[j] = (bf16_t)(vv[j] + (float)ob[j]); }
      } else if constexpr (STATS == 2) {
        const u32x4 o = yv[pt];
        const auto o0 = __builtin_amdgcn_permlane16_swap(o[0], o[2], false, false);
        const auto o1 = __builtin_amdgcn_permlane16_swap(o[1], o[3], false, false);
        const bf16x4 ya = __builtin_bit_cast(bf16x4, u32x2{o0[0], o1[0]});
        const bf16x4 yb = __builtin_bit_cast(bf16x4, u32x2{o0[1], o1[1]});
        // (dense frames: every tile pixel is a frame pixel.)  The 3 x 4 coefficients of a channel half are re-read from
        // LDS per pixel row and half -- 12 live registers instead of 24 in a kernel that must not spill: scratch traffic
        // inside the loop would join the hand-counted vmcnt stream
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          asm volatile("" ::: "memory");
          const int cb = wco * 32 + 16 * t + 4 * kb;
          const f32x4 sc = *reinterpret_cast<const f32x4*>(ctab + cb);
          const f32x4 sh = *reinterpret_cast<const f32x4*>(ctab + 64 + cb);
          const f32x4 mu = *reinterpret_cast<const f32x4*>(ctab + 128 + cb);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float fy = (float)(t ? yb[j] : ya[j]);
            const bf16_t r = (bf16_t)(fmaf(fy, sc[j], sh[j]) > 0.f ? (t ? vv[j] : va[j]) : 0.f);
            const float q0 = (float)r;                                         // dz as stored
            if (t) { rb[j] = r; sb[j] += q0; qb[j] = fmaf(q0, fy - mu[j], qb[j]); }
            else { ra[j] = r; sa[j] += q0; qa[j] = fmaf(q0, fy - mu[j], qa[j]); }
          }
        }
      } else {
        if (P.bias) {                              // inference: BatchNorm shift (+ ReLU) of the folded layer
          const float* bp = P.bias + ch0 + kb * 4;
#pragma unroll
          for (int j = 0; j < 4; ++j) { va[j] += bp[j]; vv[j] += bp[16 + j]; }
        }
        if (P.relu) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { va[j] = fmaxf(va[j], 0.f); vv[j] = fmaxf(vv[j], 0.f); }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { ra[j] = (bf16_t)va[j]; rb[j] = (bf16_t)vv[j]; }
        if constexpr (STATS == 1) {
          constexpr bool ok = true;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float q0 = ok ? (float)ra[j] : 0.f, q1 = ok ? (float)rb[j] : 0.f;   // the values as stored
            sa[j] += q0; qa[j] = fmaf(q0, q0, qa[j]);
            sb[j] += q1; qb[j] = fmaf(q1, q1, qb[j]);
          }
        }
      }
      const u32x2 ua = __builtin_bit_cast(u32x2, ra), ub = __builtin_bit_cast(u32x2, rb);
      const auto s0 = __builtin_amdgcn_permlane16_swap(ua[0], ub[0], false, false);
      const auto s1 = __builtin_amdgcn_permlane16_swap(ua[1], ub[1], false, false);
      const u32x4 bits = u32x4{s0[0], s1[0], s0[1], s1[1]};
      pend[pt] = bits;             // (unconditional: dead across the MFMA loop for the register allocator)
      if (!defer) {
#pragma unroll
        for (int q = 0; q < NVIEW; ++q) {
#ifdef WS16_NO_STORE          // diagnostic build: every store dropped (out-of-range offset), counts unchanged
          __builtin_amdgcn_raw_buffer_store_b128(bits, drs[q], OOB, 0, 0);
#else
          __builtin_amdgcn_raw_buffer_store_b128(bits, drs[q], ovb[q], pt * rowb[q], 0);
#endif
        }
      }
    }
    if constexpr (STATS != 0) {
      // the DPP row is the 16 pixels of a tile row: one reduction leaves a (statistic, channel) total of this wave's
      // 64 pixels in every lane; the four row leaders (l15 == 0, one per kb) file them in this pixel-wave's slot
      float rv[16];
#pragma unroll
      for (int j = 0; j < 4; ++j) { rv[j] = sa[j]; rv[4 + j] = sb[j]; rv[8 + j] = qa[j]; rv[12 + j] = qb[j]; }
      row16_sum_n(rv);
      if (l15 == 0) {
        float* rw = red + (k & 1) * 512 + wpx * 64 + wco * 32 + 4 * kb;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          rw[j] = rv[j]; rw[16 + j] = rv[4 + j];
          rw[256 + j] = rv[8 + j]; rw[256 + 16 + j] = rv[12 + j];
        }
      }
    }
    W6_STAMP(5)
  }
#ifdef PDMA_STAMPS
  if (STATS != 2 && P.bn_mean && lane == 0) {
    unsigned long long* o = (unsigned long long*)P.bn_mean + ((size_t)(blockIdx.x & 255) * 8 + wave) * 8;
    for (int i = 0; i < 6; ++i) o[i] = w6_st[i];
    o[6] = ((__builtin_amdgcn_s_memtime() - w6_t0) << 20) / (__builtin_amdgcn_s_memrealtime() - w6_r0 + 1);
    o[7] = (unsigned long long)(t_end - t_begin);
  }
#endif
  if (defer) {                                   // the last tile's results
#pragma unroll
    for (int pt = 0; pt < 4; ++pt)
#pragma unroll
      for (int q = 0; q < NVIEW; ++q) __builtin_amdgcn_raw_buffer_store_b128(pend[pt], drs[q], ovb[q], pt * rowb[q], 0);
  }
  if constexpr (STATS != 0) {
    // the last tile's slots, then ONE partial per block.  (The thread index is re-derived here: values computed from the
    // launch-time one before the loop would be spilled across it, and scratch traffic joins the hand-counted vmcnt stream.)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int tid2 = wave_s * 64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    if (tid2 < 128) {
      const float* rp = red + ((t_end - 1 - t_begin) & 1) * 512 + (tid2 >> 6) * 256 + (tid2 & 63);
#pragma unroll
      for (int sl = 0; sl < 4; ++sl) stat_tot += rp[sl * 64];
      P.stats[((size_t)tr * 2 + (tid2 >> 6)) * P.Cout + cg * C::ROWS + (tid2 & 63)] = stat_tot;
    }
  }
  if constexpr (ALWAYS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the dummy DMAs before the wave ends
}

int32_t launch_ws(IgemmParams P, int kclass, hipStream_t s, int* stat_parts) {
  using C = CfgWS;
  // (UNET_WS_STATS=0: statistics by the streaming pass)
  const bool stats_ok = unet_tuning().ws_stats != '0' && !P.accumulate && P.dst_split == P.Cout &&
                        P.dst[0].oy == 0 && P.dst[0].ox == 0 && P.dst[0].H == P.H && P.dst[0].W == P.W;
  const int mode = P.bn_y ? 2 : ((P.stats && stats_ok) ? 1 : 0);
  UNET_REQUIRE(mode != 2 || stats_ok, UNET_ERR_UNSUPPORTED, "conv3_ws: fused BatchNorm backward needs one dense destination");
  // the staggered schedule pays where the epilogue is long (the statistics forms: +3.5 % on 64 -> 64 @256x256) and costs
  // 20 % where it is short (plain forward / data gradient: the DMA burst of a half then lands inside the other half's
  // MFMA phase); UNET_WS_ST=0 / 1 force lock-step / staggered, 3 = staggered for the BatchNorm-backward form too
  const char stv = unet_tuning().ws_st;
  // default: the 16x16x32 kernel (conv3_ws16_kernel); UNET_WS_MFMA=3 selects the 32x32x16 one (and its staggered forms)
  // (every form of it is spill-free -- tools/check_dpp_hazards.py asserts that; 18.87 -> 18.71 ms per step when it came in)
  P.ws_stagger = unet_tuning().ws_stg == '0' ? 0 : (unet_tuning().ws_stg == '1' ? 1 : 2);
  const bool old32 = unet_tuning().ws_mfma == '3';
  bool dense16 = P.H % 16 == 0 && P.W % 16 == 0;
  for (int q = 0; q < 2; ++q)
    if (P.dst[q].p && (P.dst[q].oy || P.dst[q].ox || P.dst[q].H != P.H || P.dst[q].W != P.W)) dense16 = false;
  if (!old32 && dense16) {          // (ragged frames / offset views: conv3_ws_kernel's per-lane geometry)
    if (!P.dst[1].p) P.dst[1] = P.dst[0];         // no lane stores to it (dst_split == Cout); saves the kernel a select per tile
    auto k16 = P.accumulate ? conv3_ws16_kernel<true, 0>
                            : (mode == 2 ? conv3_ws16_kernel<false, 2> : (mode == 1 ? conv3_ws16_kernel<false, 1> : conv3_ws16_kernel<false, 0>));
    P.tilesX = cdiv(P.W, 16);
    P.tilesY = cdiv(P.H, 16);
    const long long tiles16 = (long long)P.N * P.tilesY * P.tilesX;
    const int nCg16 = P.Cout / CfgWS16::ROWS;
    int tpb16 = (int)cdiv64(tiles16 * nCg16, unet_cu_budget());
    if (tpb16 < 2) tpb16 = 2;
    const long long ranges = cdiv64(cdiv64(tiles16, tpb16), 8) * 8;
    if (mode == 0) P.stats = nullptr;
    if (stat_parts) *stat_parts = mode ? (int)ranges : 0;
    unet_set_max_lds(reinterpret_cast<const void*>(k16), CfgWS16::LDS);
    const double px16 = (double)P.N * P.H * P.W;
    ProfScope prof(kclass, 2.0 * P.N * P.H * P.W * (double)P.Cout * P.Ctot * 9, s,
                   mode == 2 ? "conv3_ws_bnbwd_kernel" : "conv3_ws_kernel",
                   2.0 * (px16 * (P.Ctot + P.Cout * (1.0 + (mode == 2 ? 1 : 0) + (P.accumulate ? 1 : 0))) + 9.0 * P.Ctot * P.Cout));
#ifdef PDMA_STAMPS
    if (mode != 2) P.bn_mean = (const float*)g_pdma_debug;
#endif
    hipLaunchKernelGGL(k16, dim3((unsigned)(ranges * nCg16)), dim3(512), CfgWS16::LDS, s, P, tpb16);
    return unet_check_launch("conv3_ws16_kernel");
  }
  const bool st = stv == '1' || (stv != '0' && (mode == 1 || (mode == 2 && stv == '3')));
  auto kern = st ? (P.accumulate ? conv3_ws_kernel<true, 0, true>
                                 : (mode == 2 ? conv3_ws_kernel<false, 2, true>
                                              : (mode == 1 ? conv3_ws_kernel<false, 1, true> : conv3_ws_kernel<false, 0, true>)))
                 : (P.accumulate ? conv3_ws_kernel<true, 0>
                                 : (mode == 2 ? conv3_ws_kernel<false, 2>
                                              : (mode == 1 ? conv3_ws_kernel<false, 1> : conv3_ws_kernel<false, 0>)));
  P.tilesX = cdiv(P.W, C::WTW);
  P.tilesY = cdiv(P.H, C::WTH);
  const long long tiles = (long long)P.N * P.tilesY * P.tilesX;
  const int nCg = P.Cout / C::ROWS;
  int tpb = (int)cdiv64(tiles * nCg, unet_cu_budget());        // one resident block per CU, one round
  if (tpb < 2) tpb = 2;
  const long long ranges8 = cdiv64(cdiv64(tiles, tpb), 8) * 8;      // tile ranges, padded to a multiple of 8 (XCDs)
  const long long blocks = ranges8 * nCg;
  const double flops = 2.0 * P.N * P.H * P.W * (double)P.Cout * P.Ctot * 9;
  if (mode == 0) P.stats = nullptr;               // (statistics, if wanted, by the caller's streaming pass)
  if (stat_parts) *stat_parts = mode ? (int)ranges8 : 0;          // one ordered partial per tile range
#ifdef PDMA_STAMPS
  if (mode != 2) P.bn_mean = (const float*)g_pdma_debug;
#endif
  ProfScope prof(kclass, flops, s, mode == 2 ? "conv3_ws_bnbwd_kernel" : "conv3_ws_kernel");
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), C::LDS, s, P, tpb);
  return unet_check_launch("conv3_ws_kernel");
}


// ------------------------------------------------------------------------------------------------------
// convt_ws_kernel<CIN>: weight-stationary streaming kernel for the wide transposed convolutions (up3: 256->128,
// up4: 128->64; AI ~ 85-170 FLOP/B => HBM-bound).  ConvTranspose2d(k2,s2) is ONE GEMM [pixels x CIN] x
// [CIN x 4*Cout] whose rows scatter to the 2x2 sub-positions.  Eight waves each keep 32 of 256 GEMM rows x CIN
// of weights in registers (CIN/16 MFMA A-fragments); 128-pixel input tiles (no halo) stream through an LDS ring
// filled by LDS-DMA; per tile and wave CIN/16 x 4 MFMAs, 16 buffer stores (8 B, bias added), one barrier.
template <int CIN>
struct CfgTW {
  static constexpr int TP = 128;                               // pixels per tile
  static constexpr int ROWP = CIN / 8 + 1;                      // 16-byte pieces per LDS row (one pad piece)
  static constexpr int RSTR = ROWP * 16;                        // 272 / 528 B: conflict-free ds_read_b128
  static constexpr int PIECES = TP * ROWP;
  static constexpr int NWAVE = 8;
  static constexpr int NINSTR = (PIECES + 63) / 64;
  static constexpr int NDMA = (NINSTR + NWAVE - 1) / NWAVE;
  static constexpr int A_BYTES = NINSTR * 1024;
  static constexpr int NBUF = (CIN <= 128) ? 3 : 2;
  static constexpr int LDS = NBUF * A_BYTES + 1024;
  static constexpr int PXT = TP / 32;                           // 4 MFMA pixel tiles per wave
  static constexpr int KGN = CIN / 16;
  static constexpr int NST = 2 * PXT;                           // 16-byte stores per wave per tile
};

struct ConvTParams {
  const char* x; char* y; const char* w; const float* bias;
  int N, H, W, Cout;      // input spatial dims; output is [N][2H][2W][Cout]
  int tiles, tiles_per_block;
  // data gradient fused with the ReLU mask and the BatchNorm-backward sums of the layer that produced the transposed
  // convolution's input (convt_dgrad_ws_kernel<COUT, true>): that layer's raw conv output, its coefficients, the
  // partial sums [blocks][2][CIN]
  const char* bn_y; const float* bn_scale; const float* bn_shift; const float* bn_mean; float* stats;
};

template <int CIN>
__global__ __launch_bounds__(512, 1) void convt_ws_kernel(const ConvTParams P) {
  using C = CfgTW<CIN>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int rows_total = 4 * P.Cout;
  const int nCg = rows_total / 256;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int cg = slot % nCg, tr = (slot / nCg) * 8 + xcd;
  const int row_lane = cg * 256 + wave * 32 + l31;              // GEMM row = z*Cout + co
  const int t_begin = tr * P.tiles_per_block;
  const int t_end = min(t_begin + P.tiles_per_block, P.tiles);
  if (t_begin >= t_end) return;

  bf16x8 wreg[C::KGN];
  {
    const bf16_t* wp = reinterpret_cast<const bf16_t*>(P.w);
#pragma unroll
    for (int kg = 0; kg < C::KGN; ++kg)
      wreg[kg] = *reinterpret_cast<const bf16x8*>(wp + (size_t)row_lane * CIN + kg * 16 + hh * 8);
    __builtin_amdgcn_s_waitcnt(0x0F70);                          // retire here, not inside the tile loop
  }

  constexpr unsigned OOB = 0xFFFFFFF0u;
  const long long total_px = (long long)P.N * P.H * P.W;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)P.x, (short)0, (int)std::min<long long>(total_px * CIN * 2, 0x7FFFFFFFLL), 0x00020000);
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)P.y, (short)0, (int)std::min<long long>(total_px * 4 * P.Cout * 2, 0x7FFFFFFFLL), 0x00020000);
  typedef __attribute__((address_space(3))) void lds_void;

  // DMA descriptors: piece q = idx*64 + lane -> (pixel row, piece in row); pad piece / beyond the tile = OOB
  int d_row[C::NDMA], d_off[C::NDMA];
#pragma unroll
  for (int j = 0; j < C::NDMA; ++j) {
    const int q = (j * C::NWAVE + wave) * 64 + lane;
    const int row = q / C::ROWP, pc = q - row * C::ROWP;
    d_row[j] = (row < C::TP && pc < CIN / 8) ? row : -1;
    d_off[j] = pc * 16;
  }
  auto dma = [&](int tile, int buf) {
    const long long p0 = (long long)tile * C::TP;
#pragma unroll
    for (int j = 0; j < C::NDMA; ++j) {
      const int idx = j * C::NWAVE + wave;
      const long long px = p0 + d_row[j];
      const bool ok = d_row[j] >= 0 && px < total_px;
      const unsigned vo = ok ? (unsigned)(px * (CIN * 2) + d_off[j]) : OOB;
      char* dst = idx < C::NINSTR ? smem + buf * C::A_BYTES + idx * 1024 : smem + C::NBUF * C::A_BYTES;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void*)dst, 16, vo, 0, 0, 0);
    }
  };

  float bias4[4][4];                      // bias of this lane's 16 rows: [g][j] -> row 8g + 4hh + j
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = cg * 256 + wave * 32 + 8 * g + 4 * hh + j;
      bias4[g][j] = P.bias ? P.bias[r % P.Cout] : 0.f;
    }
  __builtin_amdgcn_s_waitcnt(0x0F70);

  static_assert(2 * C::NST + C::NDMA * (C::NBUF - 1) <= 63, "vmcnt range");
#pragma unroll
  for (int d = 0; d < C::NBUF - 1; ++d)
    if (t_begin + d < t_end) dma(t_begin + d, d);
  const int HW = P.H * P.W;
  for (int tile = t_begin; tile < t_end; ++tile) {
    const int k = tile - t_begin;
    const int cur = k % C::NBUF;
    // ops younger than tile `tile`'s DMAs: the DMAs of the (NBUF-2) later tiles still in flight + the stores of
    // the previous tiles issued after them (exact counts; see conv3_ws_kernel)
    const int later = min(C::NBUF - 2, t_end - 1 - tile);      // later tiles whose DMAs are already issued
    const int st_tiles = min(k, C::NBUF - 1);                  // previous tiles whose stores are younger
    if constexpr (C::NBUF == 3) {
      // issue order: ... DMA(t) | stores(t-2) | DMA(t+1) | stores(t-1) |  -> younger than DMA(t):
      if (later >= 1) {
        if (st_tiles >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * C::NST + C::NDMA) : "memory");
        else if (st_tiles == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NST + C::NDMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NDMA) : "memory");
      } else {
        if (st_tiles >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * C::NST) : "memory");
        else if (st_tiles == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NST) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    } else {
      // NBUF == 2: DMA(t) was issued during tile t-1, before stores(t-1)
      if (st_tiles >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NST) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (tile + C::NBUF - 1 < t_end) dma(tile + C::NBUF - 1, (k + C::NBUF - 1) % C::NBUF);

    f32x16 acc[C::PXT];
#pragma unroll
    for (int pt = 0; pt < C::PXT; ++pt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[pt][r] = 0.f;
    const char* pb = smem + cur * C::A_BYTES + l31 * C::RSTR + hh * 16;
    {
      // pixel fragments requested two K-groups ahead of their MFMAs, pinned (see conv3_ws_kernel)
      constexpr int DEPTH = 2;
      bf16x8 ring[DEPTH + 1][C::PXT];
#pragma unroll
      for (int i = 0; i < DEPTH && i < C::KGN; ++i)
#pragma unroll
        for (int pt = 0; pt < C::PXT; ++pt)
          ring[i][pt] = *reinterpret_cast<const bf16x8*>(pb + pt * 32 * C::RSTR + i * 32);
#pragma unroll
      for (int kg = 0; kg < C::KGN; ++kg) {
        if (kg + DEPTH < C::KGN) {
#pragma unroll
          for (int pt = 0; pt < C::PXT; ++pt)
            ring[(kg + DEPTH) % (DEPTH + 1)][pt] =
                *reinterpret_cast<const bf16x8*>(pb + pt * 32 * C::RSTR + (kg + DEPTH) * 32);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pt = 0; pt < C::PXT; ++pt)
          acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[kg], ring[kg % (DEPTH + 1)][pt], acc[pt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    // ---- epilogue: scatter to (2y+zk, 2x+zl); exactly NST buffer stores per wave
#pragma unroll
    for (int pt = 0; pt < C::PXT; ++pt) {
      const long long px = (long long)tile * C::TP + pt * 32 + l31;
      const bool ok = px < total_px;
      const int n = (int)(px / HW), rem = (int)(px - (long long)n * HW);
      const int y = rem / P.W, x = rem - y * P.W;
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        // v_permlane32_swap: the lower half-wave gives its g-odd run for the upper one's g-even run; each lane
        // then owns 8 consecutive rows (16gp + 8hh ..) and writes 16 bytes
        bf16x4 xa, xb;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          xa[j] = (bf16_t)(acc[pt][8 * gp + j] + bias4[2 * gp][j]);
          xb[j] = (bf16_t)(acc[pt][8 * gp + 4 + j] + bias4[2 * gp + 1][j]);
        }
        const u32x2 ua = __builtin_bit_cast(u32x2, xa), ub = __builtin_bit_cast(u32x2, xb);
        const auto s0 = __builtin_amdgcn_permlane32_swap(ua[0], ub[0], false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(ua[1], ub[1], false, false);
        const int r = cg * 256 + wave * 32 + 16 * gp + 8 * hh;
        const int z = r / P.Cout, co = r - z * P.Cout;
        const long long opix = ((long long)n * 2 * P.H + 2 * y + (z >> 1)) * (2 * P.W) + 2 * x + (z & 1);
        const unsigned vo = ok ? (unsigned)((opix * P.Cout + co) * 2) : OOB;
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{s0[0], s1[0], s0[1], s1[1]}, yrs, vo, 0, 0);
      }
    }
  }
}

template <int CIN>
int32_t launch_convt_ws(ConvTParams P, hipStream_t s) {
  using C = CfgTW<CIN>;
  auto kern = convt_ws_kernel<CIN>;
  unet_set_max_lds(reinterpret_cast<const void*>(kern), C::LDS);
  const long long total_px = (long long)P.N * P.H * P.W;
  P.tiles = (int)cdiv64(total_px, C::TP);
  const int nCg = 4 * P.Cout / 256;
  int tpb = (int)cdiv64((long long)P.tiles * nCg, unet_cu_budget());
  if (tpb < 2) tpb = 2;
  P.tiles_per_block = tpb;
  const long long ranges8 = cdiv64(cdiv64(P.tiles, tpb), 8) * 8;
  const double flops = 2.0 * total_px * 4.0 * P.Cout * CIN;
  ProfScope prof(UNET_K_CONVT_FWD, flops, s, "convt_ws_kernel");
  hipLaunchKernelGGL(kern, dim3((unsigned)(ranges8 * nCg)), dim3(512), C::LDS, s, P);
  return unet_check_launch("convt_ws_kernel");
}


// ------------------------------------------------------------------------------------------------------
// convt_dgrad_ws_kernel<COUT>: data gradient of the wide transposed convolutions, same streaming design as
// convt_ws_kernel.  dx[p][ci] = sum_{z,co} dy[(2y+zk, 2x+zl)][co] * w[ci][z][co]: GEMM rows = CIN = 2*COUT,
// K = 4*COUT gathered from the 2x2 sub-positions of dy (the gather happens in the DMA's per-lane source address,
// the LDS row of a pixel is its 4*COUT K-vector).  Weights (32 rows x K per wave) live in registers.
template <int COUT>
struct CfgTD {
  static constexpr int CIN = 2 * COUT, K = 4 * COUT;
  static constexpr int TP = (COUT <= 64) ? 128 : 64;
  static constexpr int ROWP = K / 8 + 1;
  static constexpr int RSTR = ROWP * 16;
  static constexpr int PIECES = TP * ROWP;
  static constexpr int NWAVE = 8;
  static constexpr int NINSTR = (PIECES + 63) / 64;
  static constexpr int NDMA = (NINSTR + NWAVE - 1) / NWAVE;
  static constexpr int A_BYTES = NINSTR * 1024;
  static constexpr int NBUF = 2;
  static constexpr int CT_BASE = NBUF * A_BYTES + 1024;         // BNB: [scale | shift | mean][CIN], then the wave exchange
  static constexpr int LDS = CT_BASE + 3 * CIN * 4 + 2 * 2 * CIN * 4;
  static constexpr int RW = CIN / 32;                           // waves along rows (4 or 8)
  static constexpr int PW = NWAVE / RW;                         // waves along pixels (2 or 1)
  static constexpr int PXT = TP / PW / 32;                      // MFMA pixel tiles per wave (2)
  static constexpr int KGN = K / 16;
  static constexpr int NST = 2 * PXT;
};

// BNB: the gradient this kernel produces is d loss / d a of a conv-BatchNorm-ReLU layer (the DoubleConv in front of the
// Up block, src/model.py:14-19 -> :51): the epilogue loads that layer's raw output y with the store offsets, applies the
// ReLU mask relu'(scale*y+shift), stores dz and keeps per-lane running sums of dz and dz*(y-mean) over the block's
// tiles; one cross-lane / cross-wave reduction at the end -> one ordered partial per block (deterministic).
template <int COUT, bool BNB = false>
__global__ __launch_bounds__(512, 1) void convt_dgrad_ws_kernel(const ConvTParams P) {
  using C = CfgTD<COUT>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int wr = wave % C::RW, wp = wave / C::RW;
  const int row_lane = wr * 32 + l31;
  const int t_begin = blockIdx.x * P.tiles_per_block;
  const int t_end = min(t_begin + P.tiles_per_block, P.tiles);
  if (t_begin >= t_end) {
    if (BNB) for (int i = tid; i < 2 * C::CIN; i += 512) P.stats[(size_t)blockIdx.x * 2 * C::CIN + i] = 0.f;
    return;
  }
  float* const ctab = reinterpret_cast<float*>(smem + C::CT_BASE);
  if constexpr (BNB) {
    for (int i = tid; i < 3 * C::CIN; i += 512)
      ctab[i] = (i < C::CIN ? P.bn_scale : (i < 2 * C::CIN ? P.bn_shift : P.bn_mean))[i % C::CIN];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // (the tile loop's raw barrier does not wait for LDS writes)
  }
  float rs0[2][2][4], rs1[2][2][4];               // BNB: running sums [16-channel group][run][row] of this lane's channels
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int j = 0; j < 4; ++j) { rs0[a][b][j] = 0.f; rs1[a][b][j] = 0.f; }

  bf16x8 wreg[C::KGN];
  {
    const bf16_t* wpk = reinterpret_cast<const bf16_t*>(P.w);
#pragma unroll
    for (int kg = 0; kg < C::KGN; ++kg)
      wreg[kg] = *reinterpret_cast<const bf16x8*>(wpk + (size_t)row_lane * C::K + kg * 16 + hh * 8);
    __builtin_amdgcn_s_waitcnt(0x0F70);
  }

  constexpr unsigned OOB = 0xFFFFFFF0u;
  const long long total_px = (long long)P.N * P.H * P.W;       // dx pixels (half-resolution grid)
  const int HW = P.H * P.W;
  // P.x = dy [N][2H][2W][COUT], P.y = dx [N][H][W][CIN]
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)P.x, (short)0, (int)std::min<long long>(total_px * 4 * COUT * 2, 0x7FFFFFFFLL), 0x00020000);
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)P.y, (short)0, (int)std::min<long long>(total_px * C::CIN * 2, 0x7FFFFFFFLL), 0x00020000);
  const __amdgpu_buffer_rsrc_t bnrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(BNB ? P.bn_y : P.x), (short)0, (int)std::min<long long>(total_px * C::CIN * 2, 0x7FFFFFFFLL), 0x00020000);
  (void)bnrs;
  typedef __attribute__((address_space(3))) void lds_void;

  int d_row[C::NDMA], d_z[C::NDMA], d_c[C::NDMA];
#pragma unroll
  for (int j = 0; j < C::NDMA; ++j) {
    const int q = (j * C::NWAVE + wave) * 64 + lane;
    const int row = q / C::ROWP, pc = q - row * C::ROWP;
    d_row[j] = (row < C::TP && pc < C::K / 8) ? row : -1;
    d_z[j] = pc / (COUT / 8);
    d_c[j] = (pc % (COUT / 8)) * 16;
  }
  auto dma = [&](int tile, int buf, bool live = true) {
    const long long p0 = (long long)tile * C::TP;
#pragma unroll
    for (int j = 0; j < C::NDMA; ++j) {
      const int idx = j * C::NWAVE + wave;
      const long long px = p0 + d_row[j];
      const bool ok = live && d_row[j] >= 0 && px < total_px;
      const int n = (int)(px / HW), rem = (int)(px - (long long)n * HW);
      const int y = rem / P.W, x = rem - y * P.W;
      const long long ipix = ((long long)n * 2 * P.H + 2 * y + (d_z[j] >> 1)) * (2 * P.W) + 2 * x + (d_z[j] & 1);
      const unsigned vo = ok ? (unsigned)(ipix * (COUT * 2) + d_c[j]) : OOB;
      char* dst = (live && idx < C::NINSTR) ? smem + buf * C::A_BYTES + idx * 1024 : smem + C::NBUF * C::A_BYTES;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void*)dst, 16, vo, 0, 0, 0);
    }
  };

  dma(t_begin, 0);
  for (int tile = t_begin; tile < t_end; ++tile) {
    const int k = tile - t_begin;
    const int cur = k & 1;
    // DMA(t) was issued during tile t-1 BEFORE stores(t-1): exactly NST younger ops
    if (k >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NST) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // BNB: this tile's y values (same offsets as the stores) are requested FIRST, then the next tile's DMAs, so that the
    // epilogue's wait for them leaves those DMAs in flight
    u32x4 yv[BNB ? C::PXT : 1][2];
    unsigned ovo[C::PXT][2];
#pragma unroll
    for (int pt = 0; pt < C::PXT; ++pt) {
      const long long px = (long long)tile * C::TP + (wp * C::PXT + pt) * 32 + l31;
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        ovo[pt][gp] = px < total_px ? (unsigned)((px * C::CIN + wr * 32 + 16 * gp + 8 * hh) * 2) : OOB;
        // (inline asm: hipcc does not count LDS-DMA instructions, so its own wait for a builtin load issued in front of
        //  the next tile's DMAs would be vmcnt(3) -- draining those DMAs every tile; the wait is hand-counted below)
        if constexpr (BNB)
          asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(yv[pt][gp]) : "v"(ovo[pt][gp]), "s"(bnrs) : "memory");
      }
    }
    if (BNB || tile + 1 < t_end) dma(tile + 1, cur ^ 1, tile + 1 < t_end);   // (BNB: always NDMA instructions -> one wait form)

    f32x16 acc[C::PXT];
#pragma unroll
    for (int pt = 0; pt < C::PXT; ++pt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[pt][r] = 0.f;
    const char* pb = smem + cur * C::A_BYTES + (wp * C::PXT * 32 + l31) * C::RSTR + hh * 16;
    {
      // pixel fragments requested two K-groups ahead of their MFMAs, pinned (see conv3_ws_kernel)
      constexpr int DEPTH = BNB ? 1 : 2;            // (the fused form needs the registers for its sums and y values)
      bf16x8 ring[DEPTH + 1][C::PXT];
#pragma unroll
      for (int i = 0; i < DEPTH && i < C::KGN; ++i)
#pragma unroll
        for (int pt = 0; pt < C::PXT; ++pt)
          ring[i][pt] = *reinterpret_cast<const bf16x8*>(pb + pt * 32 * C::RSTR + i * 32);
#pragma unroll
      for (int kg = 0; kg < C::KGN; ++kg) {
        if (kg + DEPTH < C::KGN) {
#pragma unroll
          for (int pt = 0; pt < C::PXT; ++pt)
            ring[(kg + DEPTH) % (DEPTH + 1)][pt] =
                *reinterpret_cast<const bf16x8*>(pb + pt * 32 * C::RSTR + (kg + DEPTH) * 32);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pt = 0; pt < C::PXT; ++pt)
          acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[kg], ring[kg % (DEPTH + 1)][pt], acc[pt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (BNB) {
      // the y loads are older than the next tile's DMAs: leave exactly those in flight
      static_assert(C::PXT == 2, "wait statement names 4 destinations");
      asm volatile("s_waitcnt vmcnt(%4)" : "+v"(yv[0][0]), "+v"(yv[0][1]), "+v"(yv[1][0]), "+v"(yv[1][1]) : "n"(C::NDMA));
    }
#pragma unroll
    for (int pt = 0; pt < C::PXT; ++pt) {
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {               // 16-byte stores (see convt_ws_kernel)
        bf16x4 xa, xb;
        if constexpr (BNB) {
          // y comes in with the store's 8-consecutive-channel layout: un-swap it to the accumulator's two 4-row runs
          const u32x4 o = yv[pt][gp];
          const auto o0 = __builtin_amdgcn_permlane32_swap(o[0], o[2], false, false);
          const auto o1 = __builtin_amdgcn_permlane32_swap(o[1], o[3], false, false);
          const bf16x4 ya = __builtin_bit_cast(bf16x4, u32x2{o0[0], o1[0]});
          const bf16x4 yb = __builtin_bit_cast(bf16x4, u32x2{o0[1], o1[1]});
          const bool ok = ovo[pt][gp] != OOB;
          const int cb = wr * 32 + 16 * gp + 4 * hh;
          const f32x4 sca = *reinterpret_cast<const f32x4*>(ctab + cb), scb = *reinterpret_cast<const f32x4*>(ctab + cb + 8);
          const f32x4 sha = *reinterpret_cast<const f32x4*>(ctab + C::CIN + cb);
          const f32x4 shb = *reinterpret_cast<const f32x4*>(ctab + C::CIN + cb + 8);
          const f32x4 mua = *reinterpret_cast<const f32x4*>(ctab + 2 * C::CIN + cb);
          const f32x4 mub = *reinterpret_cast<const f32x4*>(ctab + 2 * C::CIN + cb + 8);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float fa = (float)ya[j], fb = (float)yb[j];
            xa[j] = (bf16_t)((ok && fmaf(fa, sca[j], sha[j]) > 0.f) ? acc[pt][8 * gp + j] : 0.f);
            xb[j] = (bf16_t)((ok && fmaf(fb, scb[j], shb[j]) > 0.f) ? acc[pt][8 * gp + 4 + j] : 0.f);
            const float qa = (float)xa[j], qb = (float)xb[j];                // dz as stored
            rs0[gp][0][j] += qa;
            rs1[gp][0][j] = fmaf(qa, fa - mua[j], rs1[gp][0][j]);
            rs0[gp][1][j] += qb;
            rs1[gp][1][j] = fmaf(qb, fb - mub[j], rs1[gp][1][j]);
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) { xa[j] = (bf16_t)acc[pt][8 * gp + j]; xb[j] = (bf16_t)acc[pt][8 * gp + 4 + j]; }
        }
        const u32x2 ua = __builtin_bit_cast(u32x2, xa), ub = __builtin_bit_cast(u32x2, xb);
        const auto s0 = __builtin_amdgcn_permlane32_swap(ua[0], ub[0], false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(ua[1], ub[1], false, false);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{s0[0], s1[0], s0[1], s1[1]}, yrs, ovo[pt][gp], 0, 0);
      }
    }
  }
  if constexpr (BNB) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the last tile's dummy DMAs)
    // the block's partial: sums over the 32 pixel lanes of a half-wave (fixed butterfly order), then over the PW pixel
    // waves through LDS, one store per (statistic, channel)
    float* ex = ctab + 3 * C::CIN;                  // [PW][2][CIN]
#pragma unroll
    for (int gp = 0; gp < 2; ++gp)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float a = rs0[gp][t][j], b = rs1[gp][t][j];
#pragma unroll
          for (int m = 1; m < 32; m <<= 1) { a += __shfl_xor(a, m); b += __shfl_xor(b, m); }
          if (l31 == 0) {
            const int ch = wr * 32 + 16 * gp + 8 * t + 4 * hh + j;
            ex[(wp * 2 + 0) * C::CIN + ch] = a;
            ex[(wp * 2 + 1) * C::CIN + ch] = b;
          }
        }
    __syncthreads();
    for (int i = tid; i < 2 * C::CIN; i += 512) {
      float t = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < C::PW; ++w2) t += ex[(w2 * 2 + i / C::CIN) * C::CIN + i % C::CIN];
      P.stats[(size_t)blockIdx.x * 2 * C::CIN + i] = t;
    }
  }
}

template <int COUT>
int32_t launch_convt_dgrad_ws(ConvTParams P, hipStream_t s, int* n_parts = nullptr) {
  using C = CfgTD<COUT>;
  const bool bnb = P.bn_y != nullptr;
  auto kern = (bnb && COUT == 64) ? convt_dgrad_ws_kernel<COUT, (COUT == 64)> : convt_dgrad_ws_kernel<COUT, false>;
  unet_set_max_lds(reinterpret_cast<const void*>(kern), C::LDS);
  const long long total_px = (long long)P.N * P.H * P.W;
  P.tiles = (int)cdiv64(total_px, C::TP);
  int tpb = (int)cdiv64(P.tiles, std::min(256, unet_cu_budget()));     // (partial buffer: 256 parts)
  if (tpb < 2) tpb = 2;
  P.tiles_per_block = tpb;
  const long long blocks = cdiv64(P.tiles, tpb);
  if (n_parts) *n_parts = (int)blocks;
  const double flops = 2.0 * total_px * 4.0 * COUT * C::CIN;
  ProfScope prof(UNET_K_CONVT_DGRAD, flops, s, bnb ? "convt_dgrad_ws_bnbwd_kernel" : "convt_dgrad_ws_kernel");
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), C::LDS, s, P);
  return unet_check_launch("convt_dgrad_ws_kernel");
}

template <typename T, int TAPS, int BN, int KG>
int32_t launch(const IgemmParams& Pin, int kclass, hipStream_t s) {
  using C = Cfg<T, TAPS, BN, KG>;
  IgemmParams P = Pin;
  P.stats = nullptr;
  auto kern = igemm_kernel<T, TAPS, BN, KG>;
  unet_set_max_lds(reinterpret_cast<const void*>(kern), C::LDS);
  const long long blocks = (long long)P.N * P.tilesY * P.tilesX * P.nCo * P.nZ;
  UNET_REQUIRE(blocks > 0 && blocks < (1LL << 31), UNET_ERR_UNSUPPORTED, "igemm: grid of %lld blocks", blocks);
  const double flops = 2.0 * P.N * P.H * P.W * (double)P.Cout * P.Ctot * TAPS * P.gtaps * P.nZ;
  ProfScope prof(kclass, flops, s, "igemm_kernel");
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), C::LDS, s, P);
  return unet_check_launch("igemm_kernel");
}

// The specialised 3x3 kernels address one image plane of every view with 32-bit buffer offsets (descriptor
// num_records and voffset): a plane of 2 GiB or more (e.g. 4096x4096x64 bf16) must take the generic kernel, whose
// addressing is 64-bit.
template <typename T>
inline bool planes_fit_32bit(const IgemmParams& P) {
  auto ok = [](long long h, long long w, long long c) { return h * w * c * (long long)sizeof(T) < 0x7FFFFFFFLL; };
  return ok(P.src[0].H, P.src[0].W, P.src[0].C) && ok(P.src[1].H, P.src[1].W, P.src[1].C) &&
         ok(P.dst[0].H, P.dst[0].W, P.dst[0].C) && ok(P.dst[1].H, P.dst[1].W, P.dst[1].C);
}

template <typename T, int TAPS>
int32_t dispatch(IgemmParams& P, int kclass, hipStream_t s, int* stat_parts = nullptr) {
  if (stat_parts) *stat_parts = 0;
  constexpr int CK4 = 4 * ET<T>::KGC;
  UNET_REQUIRE(P.Cout % 64 == 0, UNET_ERR_UNSUPPORTED, "igemm: c_out %d is not a multiple of 64", P.Cout);
  UNET_REQUIRE(P.Ctot % ET<T>::KGC == 0, UNET_ERR_UNSUPPORTED, "igemm: input channels %d not a multiple of %d",
               P.Ctot, ET<T>::KGC);
  const bool big = (P.Cout % 128 == 0);
  const bool k4 = (P.Ctot % CK4 == 0) && (P.src[1].C == 0 || P.src[0].C % CK4 == 0);
  if (!k4)
    UNET_REQUIRE(P.src[1].C == 0 || P.src[0].C % ET<T>::KGC == 0, UNET_ERR_UNSUPPORTED,
                 "igemm: concat split %d not chunk aligned", P.src[0].C);
  P.nCo = P.Cout / (big ? 128 : 64);
  P.tilesX = cdiv(P.W, TW);
  P.tilesY = cdiv(P.H, TH);
  const char impl = unet_tuning().conv_impl;    // UNET_CONV_IMPL: '0' = generic igemm_kernel, '2' no weight-stationary, '3' register-staged
  const bool small = planes_fit_32bit<T>(P);
  const bool use3 = small && impl != '0';
  if constexpr (TAPS == 9 && sizeof(T) == 2) {
    // 64-channel inputs: weight-stationary streaming kernel (impl "2" forces it off)
    const bool ws_ok = small && P.Ctot == 64 && P.src[1].C == 0 && impl != '0' && impl != '2';
    if (ws_ok) return launch_ws(P, kclass, s, stat_parts);
  }
  if constexpr (TAPS == 9 && sizeof(T) == 2) {
    // deep layers (>= 4 input chunks: below that the un-overlapped prologue of the one block per CU costs more
    // than it saves): both operands by LDS-DMA, 512-thread blocks (impl "3" = the register-staged kernels)
    const bool dma_ok = small && k4 && P.Ctot >= 128 && P.H % 16 == 0 && P.W % 16 == 0 && impl != '0' && impl != '3';
    if (dma_ok) return big ? launch_pdma<128>(P, kclass, s, stat_parts) : launch_pdma<64>(P, kclass, s, stat_parts);
  }
  if constexpr (TAPS == 9) {
    if (!use3 || P.Ctot < 2 * CK4) {
      if (big) return k4 ? launch<T, TAPS, 128, 4>(P, kclass, s) : launch<T, TAPS, 128, 1>(P, kclass, s);
      return k4 ? launch<T, TAPS, 64, 4>(P, kclass, s) : launch<T, TAPS, 64, 1>(P, kclass, s);
    }
    if (big) return k4 ? launch3<T, 128, 4>(P, kclass, s, stat_parts) : launch3<T, 128, 1>(P, kclass, s, stat_parts);
    return k4 ? launch3<T, 64, 4>(P, kclass, s, stat_parts) : launch3<T, 64, 1>(P, kclass, s, stat_parts);
  } else {
    if (big) return k4 ? launch<T, TAPS, 128, 4>(P, kclass, s) : launch<T, TAPS, 128, 1>(P, kclass, s);
    return k4 ? launch<T, TAPS, 64, 4>(P, kclass, s) : launch<T, TAPS, 64, 1>(P, kclass, s);
  }
}

inline DView in_view(const unet_view& v) { return DView{(const char*)v.ptr, v.c, v.h, v.w, v.off_y, v.off_x}; }
inline DViewW out_view(const unet_view& v) { return DViewW{(char*)v.ptr, v.c, v.h, v.w, v.off_y, v.off_x}; }

}  // namespace

extern "C" int32_t unet_conv3x3(int32_t dtype, int32_t n, int32_t h, int32_t w, const unet_view src[2],
                                const void* w_packed, int32_t c_out, const unet_view dst[2],
                                int32_t dst_split, int32_t accumulate, int32_t kclass, void* stream) {
  UNET_REQUIRE(src && dst && w_packed && src[0].ptr && dst[0].ptr, UNET_ERR_BAD_ARG, "unet_conv3x3: null pointer");
  UNET_REQUIRE(n > 0 && h > 0 && w > 0 && c_out > 0, UNET_ERR_BAD_ARG, "unet_conv3x3: bad dims");
  UNET_REQUIRE(dst_split == c_out || dst[1].ptr, UNET_ERR_BAD_ARG, "unet_conv3x3: dst[1] missing");
  UNET_REQUIRE(dst_split % 64 == 0 && dst_split > 0 && dst_split <= c_out, UNET_ERR_UNSUPPORTED,
               "unet_conv3x3: dst_split %d", dst_split);
  IgemmParams P{};
  P.src[0] = in_view(src[0]);
  P.src[1] = src[1].ptr ? in_view(src[1]) : DView{nullptr, 0, 0, 0, 0, 0};
  P.dst[0] = out_view(dst[0]);
  P.dst[1] = dst[1].ptr ? out_view(dst[1]) : DViewW{nullptr, 0, 0, 0, 0, 0};
  P.N = n; P.H = h; P.W = w;
  P.Ctot = P.src[0].C + P.src[1].C;
  P.Cout = c_out;
  P.wK = P.Ctot;
  P.w = (const char*)w_packed;
  P.bias = nullptr;
  P.dst_split = dst_split;
  P.accumulate = accumulate;
  P.imul = 1; P.gtaps = 1; P.omul = 1; P.nZ = 1;
  if (kclass < 0 || kclass >= UNET_K_COUNT) kclass = UNET_K_CONV_FWD;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == UNET_BF16) return dispatch<bf16_t, 9>(P, kclass, s);
  if (dtype == UNET_F32) return dispatch<float, 9>(P, kclass, s);
  unet_set_error("unet_conv3x3: dtype %d", dtype);
  return UNET_ERR_BAD_ARG;
}

extern "C" int32_t unet_conv3x3_bias_relu(int32_t dtype, int32_t n, int32_t h, int32_t w, const unet_view src[2],
                                          const void* w_packed, int32_t c_out, void* y, const float* bias,
                                          int32_t relu, void* stream) {
  UNET_REQUIRE(src && w_packed && src[0].ptr && y && bias, UNET_ERR_BAD_ARG, "unet_conv3x3_bias_relu: null pointer");
  UNET_REQUIRE(n > 0 && h > 0 && w > 0 && c_out > 0, UNET_ERR_BAD_ARG, "unet_conv3x3_bias_relu: bad dims");
  IgemmParams P{};
  P.src[0] = in_view(src[0]);
  P.src[1] = src[1].ptr ? in_view(src[1]) : DView{nullptr, 0, 0, 0, 0, 0};
  P.dst[0] = DViewW{(char*)y, c_out, h, w, 0, 0};
  P.dst[1] = DViewW{nullptr, 0, 0, 0, 0, 0};
  P.N = n; P.H = h; P.W = w;
  P.Ctot = P.src[0].C + P.src[1].C;
  P.Cout = c_out;
  P.wK = P.Ctot;
  P.w = (const char*)w_packed;
  P.bias = bias;
  P.relu = relu;
  P.dst_split = c_out;
  P.imul = 1; P.gtaps = 1; P.omul = 1; P.nZ = 1;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == UNET_BF16) return dispatch<bf16_t, 9>(P, UNET_K_CONV_FWD, s);
  if (dtype == UNET_F32) return dispatch<float, 9>(P, UNET_K_CONV_FWD, s);
  unet_set_error("unet_conv3x3_bias_relu: dtype %d", dtype);
  return UNET_ERR_BAD_ARG;
}

extern "C" size_t unet_conv3x3_stats_max_parts(int32_t n, int32_t h, int32_t w) {
  const size_t tiles = (size_t)n * cdiv(h, TH) * cdiv(w, TW);
  return tiles > 1024 ? tiles : 1024;
}

extern "C" int32_t unet_conv3x3_stats(int32_t dtype, int32_t n, int32_t h, int32_t w, const unet_view src[2],
                                      const void* w_packed, int32_t c_out, void* y, float* partial,
                                      int32_t* n_parts, void* stream) {
  UNET_REQUIRE(src && w_packed && src[0].ptr && y && partial && n_parts, UNET_ERR_BAD_ARG,
               "unet_conv3x3_stats: null pointer");
  UNET_REQUIRE(n > 0 && h > 0 && w > 0 && c_out > 0, UNET_ERR_BAD_ARG, "unet_conv3x3_stats: bad dims");
  IgemmParams P{};
  P.src[0] = in_view(src[0]);
  P.src[1] = src[1].ptr ? in_view(src[1]) : DView{nullptr, 0, 0, 0, 0, 0};
  P.dst[0] = DViewW{(char*)y, c_out, h, w, 0, 0};
  P.dst[1] = DViewW{nullptr, 0, 0, 0, 0, 0};
  P.N = n; P.H = h; P.W = w;
  P.Ctot = P.src[0].C + P.src[1].C;
  P.Cout = c_out;
  P.wK = P.Ctot;
  P.w = (const char*)w_packed;
  P.dst_split = c_out;
  P.imul = 1; P.gtaps = 1; P.omul = 1; P.nZ = 1;
  P.stats = unet_tuning().fused_stats == '0' ? nullptr : partial;      // (UNET_FUSED_STATS=0: always the streaming pass)
  hipStream_t s = (hipStream_t)stream;
  int parts = 0;
  int32_t rc;
  if (dtype == UNET_BF16) rc = dispatch<bf16_t, 9>(P, UNET_K_CONV_FWD, s, &parts);
  else if (dtype == UNET_F32) rc = dispatch<float, 9>(P, UNET_K_CONV_FWD, s, &parts);
  else { unet_set_error("unet_conv3x3_stats: dtype %d", dtype); return UNET_ERR_BAD_ARG; }
  if (rc) return rc;
  if (parts == 0)   // this kernel variant has no fused statistics: one streaming pass over y instead
    rc = unet_internal_bn_partials(dtype, y, (int64_t)n * h * w, c_out, partial, &parts, s);
  *n_parts = parts;
  return rc;
}

// ---- data gradient of a 3x3 convolution fused with the ReLU mask and the BatchNorm-backward sums of the layer that
// produced the convolution's input (the internal activation of DoubleConv, src/model.py:14-19)
namespace {
inline bool dgrad_bnrelu_pdma_ok(int dtype, int n, int h, int w, int c_in_gemm, int c_out_gemm) {
  (void)n;
  return dtype == UNET_BF16 && c_in_gemm >= 128 && c_in_gemm % 64 == 0 && c_out_gemm % 64 == 0 && h % 16 == 0 &&
         w % 16 == 0 && (long long)h * w * c_out_gemm * 2 < 0x7FFFFFFFLL && (long long)h * w * c_in_gemm * 2 < 0x7FFFFFFFLL;
}
// 64 -> 64 (the full-resolution level): the weight-stationary streaming kernel, any frame size
inline bool dgrad_bnrelu_ws_ok(int dtype, int n, int h, int w, int c_in_gemm, int c_out_gemm) {
  (void)n;
  return dtype == UNET_BF16 && c_in_gemm == 64 && c_out_gemm == 64 && (long long)h * w * 64 * 2 < 0x7FFFFFFFLL &&
         unet_tuning().ws_stats != '0';
}
}  // namespace

extern "C" int32_t unet_conv3x3_dgrad_bnrelu_supported(int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c_dy,
                                                       int32_t c_dx) {
  if (unet_tuning().dgrad_bn == '0') return 0;            // (UNET_DGRAD_BN=0: never)
  return (dgrad_bnrelu_pdma_ok(dtype, n, h, w, c_dy, c_dx) || dgrad_bnrelu_ws_ok(dtype, n, h, w, c_dy, c_dx)) ? 1 : 0;
}

extern "C" int32_t unet_conv3x3_dgrad_bnrelu(int32_t dtype, int32_t n, int32_t h, int32_t w, const void* dy, int32_t c_dy,
                                             const void* w_packed, int32_t c_dx, const void* y_prev,
                                             const float* bn_scale, const float* bn_shift, const float* bn_mean,
                                             void* dz, float* partial, int32_t* n_parts, void* stream) {
  UNET_REQUIRE(dy && w_packed && y_prev && bn_scale && bn_shift && bn_mean && dz && partial && n_parts, UNET_ERR_BAD_ARG,
               "unet_conv3x3_dgrad_bnrelu: null pointer");
  UNET_REQUIRE(n > 0 && h > 0 && w > 0, UNET_ERR_BAD_ARG, "unet_conv3x3_dgrad_bnrelu: bad dims");
  UNET_REQUIRE(unet_conv3x3_dgrad_bnrelu_supported(dtype, n, h, w, c_dy, c_dx), UNET_ERR_UNSUPPORTED,
               "unet_conv3x3_dgrad_bnrelu: %d -> %d channels at %dx%d (dtype %d) is not covered; use unet_conv3x3 + "
               "unet_bn_relu_bwd", c_dy, c_dx, h, w, dtype);
  IgemmParams P{};
  P.src[0] = DView{(const char*)dy, c_dy, h, w, 0, 0};
  P.src[1] = DView{nullptr, 0, 0, 0, 0, 0};
  P.dst[0] = DViewW{(char*)dz, c_dx, h, w, 0, 0};
  P.dst[1] = DViewW{nullptr, 0, 0, 0, 0, 0};
  P.N = n; P.H = h; P.W = w;
  P.Ctot = c_dy;
  P.Cout = c_dx;
  P.wK = c_dy;
  P.w = (const char*)w_packed;
  P.dst_split = c_dx;
  P.imul = 1; P.gtaps = 1; P.omul = 1; P.nZ = 1;
  P.stats = partial;
  P.bn_y = (const char*)y_prev;
  P.bn_scale = bn_scale; P.bn_shift = bn_shift; P.bn_mean = bn_mean;
  int parts = 0;
  hipStream_t s = (hipStream_t)stream;
  int32_t rc;
  if (dgrad_bnrelu_pdma_ok(dtype, n, h, w, c_dy, c_dx))
    rc = (c_dx % 128 == 0) ? launch_pdma<128>(P, UNET_K_CONV_DGRAD, s, &parts) : launch_pdma<64>(P, UNET_K_CONV_DGRAD, s, &parts);
  else
    rc = launch_ws(P, UNET_K_CONV_DGRAD, s, &parts);
  *n_parts = parts;
  return rc;
}

extern "C" int32_t unet_convt2x2_fwd(int32_t dtype, int32_t n, int32_t h, int32_t w, const void* x,
                                     int32_t c_in, const void* w_packed, const float* bias, void* y,
                                     int32_t c_out, void* stream) {
  UNET_REQUIRE(x && w_packed && y, UNET_ERR_BAD_ARG, "unet_convt2x2_fwd: null pointer");
  UNET_REQUIRE(n > 0 && h > 0 && w > 0, UNET_ERR_BAD_ARG, "unet_convt2x2_fwd: bad dims");
  {
    const long long out_bytes = (long long)n * 4 * h * w * c_out * 2;
    if (dtype == UNET_BF16 && c_in == 2 * c_out && (c_in == 128 || c_in == 256) && out_bytes < 0x7FFFFFFFLL &&
        unet_tuning().convt_impl != '0') {                     // (UNET_CONVT_IMPL=0: generic igemm path)
      ConvTParams T{(const char*)x, (char*)y, (const char*)w_packed, bias, n, h, w, c_out, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr};
      return c_in == 128 ? launch_convt_ws<128>(T, (hipStream_t)stream) : launch_convt_ws<256>(T, (hipStream_t)stream);
    }
  }
  if (unet_internal_convt_gemm_ok(0, dtype, n, h, w, c_in, c_out))     // deep levels: one LDS-DMA GEMM (convt_gemm.hip)
    return unet_internal_convt_gemm(0, n, h, w, x, w_packed, bias, y, c_in, c_out, (hipStream_t)stream);
  IgemmParams P{};
  P.src[0] = DView{(const char*)x, c_in, h, w, 0, 0};
  P.dst[0] = DViewW{(char*)y, c_out, 2 * h, 2 * w, 0, 0};
  P.N = n; P.H = h; P.W = w;
  // one GEMM with 4*c_out rows (row = z*c_out + co, the packed layout [4][c_out][c_in] read as one matrix):
  // the input tile is staged once for all four sub-positions
  P.Ctot = c_in; P.Cout = 4 * c_out; P.wK = c_in;
  P.w = (const char*)w_packed;
  P.bias = bias;
  P.dst_split = 4 * c_out;
  P.imul = 1; P.gtaps = 1; P.omul = 2; P.nZ = 1; P.zdiv = c_out;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == UNET_BF16) return dispatch<bf16_t, 1>(P, UNET_K_CONVT_FWD, s);
  if (dtype == UNET_F32) return dispatch<float, 1>(P, UNET_K_CONVT_FWD, s);
  unet_set_error("unet_convt2x2_fwd: dtype %d", dtype);
  return UNET_ERR_BAD_ARG;
}

extern "C" int32_t unet_convt2x2_dgrad(int32_t dtype, int32_t n, int32_t h, int32_t w, const void* dy,
                                       int32_t c_out, const void* w_packed, void* dx, int32_t c_in,
                                       void* stream) {
  UNET_REQUIRE(dy && w_packed && dx, UNET_ERR_BAD_ARG, "unet_convt2x2_dgrad: null pointer");
  UNET_REQUIRE(n > 0 && h > 0 && w > 0, UNET_ERR_BAD_ARG, "unet_convt2x2_dgrad: bad dims");
  {
    const long long in_bytes = (long long)n * 4 * h * w * c_out * 2;
    if (dtype == UNET_BF16 && c_in == 2 * c_out && (c_out == 64 || c_out == 128) && in_bytes < 0x7FFFFFFFLL &&
        unet_tuning().convt_impl != '0') {
      ConvTParams T{(const char*)dy, (char*)dx, (const char*)w_packed, nullptr, n, h, w, c_out, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr};
      return c_out == 64 ? launch_convt_dgrad_ws<64>(T, (hipStream_t)stream)
                         : launch_convt_dgrad_ws<128>(T, (hipStream_t)stream);
    }
  }
  if (unet_internal_convt_gemm_ok(1, dtype, n, h, w, c_in, c_out))
    return unet_internal_convt_gemm(1, n, h, w, dy, w_packed, nullptr, dx, c_in, c_out, (hipStream_t)stream);
  IgemmParams P{};
  P.src[0] = DView{(const char*)dy, c_out, 2 * h, 2 * w, 0, 0};
  P.dst[0] = DViewW{(char*)dx, c_in, h, w, 0, 0};
  P.N = n; P.H = h; P.W = w;
  P.Ctot = c_out; P.Cout = c_in; P.wK = 4 * c_out;
  P.w = (const char*)w_packed;
  P.bias = nullptr;
  P.dst_split = c_in;
  P.imul = 2; P.gtaps = 4; P.omul = 1; P.nZ = 1;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == UNET_BF16) return dispatch<bf16_t, 1>(P, UNET_K_CONVT_DGRAD, s);
  if (dtype == UNET_F32) return dispatch<float, 1>(P, UNET_K_CONVT_DGRAD, s);
  unet_set_error("unet_convt2x2_dgrad: dtype %d", dtype);
  return UNET_ERR_BAD_ARG;
}

// ---- data gradient of a transposed convolution fused with the ReLU mask and the BatchNorm-backward sums of the layer that
// produced its input (the DoubleConv in front of an Up block, src/model.py:14-19 -> :51)
extern "C" int32_t unet_convt2x2_dgrad_bnrelu_supported(int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c_in,
                                                        int32_t c_out) {
  const long long in_bytes = (long long)n * 4 * h * w * c_out * 2;
  // (c_out == 128 -- 128 weight registers per lane -- has no room for the running sums: 78 spills; not offered)
  return (dtype == UNET_BF16 && c_in == 2 * c_out && c_out == 64 && in_bytes < 0x7FFFFFFFLL &&
          unet_tuning().convt_impl != '0' && unet_tuning().dgrad_bn != '0') ? 1 : 0;
}

extern "C" size_t unet_convt2x2_dgrad_bnrelu_max_parts(void) { return 256; }

extern "C" int32_t unet_convt2x2_dgrad_bnrelu(int32_t dtype, int32_t n, int32_t h, int32_t w, const void* dy, int32_t c_out,
                                              const void* w_packed, const void* y_prev, const float* bn_scale,
                                              const float* bn_shift, const float* bn_mean, void* dz, int32_t c_in,
                                              float* partial, int32_t* n_parts, void* stream) {
  UNET_REQUIRE(dy && w_packed && y_prev && bn_scale && bn_shift && bn_mean && dz && partial && n_parts, UNET_ERR_BAD_ARG,
               "unet_convt2x2_dgrad_bnrelu: null pointer");
  UNET_REQUIRE(n > 0 && h > 0 && w > 0, UNET_ERR_BAD_ARG, "unet_convt2x2_dgrad_bnrelu: bad dims");
  UNET_REQUIRE(unet_convt2x2_dgrad_bnrelu_supported(dtype, n, h, w, c_in, c_out), UNET_ERR_UNSUPPORTED,
               "unet_convt2x2_dgrad_bnrelu: %d <- %d channels at %dx%d (dtype %d) is not covered; use unet_convt2x2_dgrad + "
               "unet_bn_relu_bwd", c_in, c_out, h, w, dtype);
  ConvTParams T{(const char*)dy, (char*)dz, (const char*)w_packed, nullptr, n, h, w, c_out, 0, 0,
                (const char*)y_prev, bn_scale, bn_shift, bn_mean, partial};
  int parts = 0;
  const int32_t rc = launch_convt_dgrad_ws<64>(T, (hipStream_t)stream, &parts);
  *n_parts = parts;
  return rc;
}

#ifdef PDMA_STAMPS
extern "C" void unet_debug_set_buffer(void* p) { g_pdma_debug = p; }
#endif
