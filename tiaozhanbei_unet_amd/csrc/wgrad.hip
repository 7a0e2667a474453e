// Weight-gradient GEMMs on MFMA for gfx950: the reduction runs over PIXELS.
//
//   conv3x3 :  dW[co][ci][r][s]  = sum_{n,y,x} dY[n,y,x,co] * X[n,y+r-1,x+s-1,ci]     (TAPS=9, S=1, PAD=1)
//   convT2x2:  dW[ci][co][k][l]  = sum_{n,i,j} X[n,i,j,ci]  * dY[n,2i+k,2j+l,co]      (TAPS=4, S=2, PAD=0)
// (autograd of nn.Conv2d / nn.ConvTranspose2d, /root/reference/src/model.py:14,17,51, reached from
//  total_loss.backward() at src/train_utils.py:132.)
//
// Both are "out[r][c][tap] = sum_p R[p][r] * Cn[p*S + tap - PAD][c]" with a row tensor R and a column
// tensor Cn.  Block = 64 rows x 64 cols x all taps (each wave 32x32 x TAPS fp32 accumulators), looping
// over a range of pixel tiles (split-K).  Per tile the R tile and the halo'd Cn patch are staged ONCE in
// LDS and reused by every tap.  The MFMA K dimension is the pixel index, which is the SLOW axis of NHWC,
// so bf16 fragments are fetched with ds_read_b64_tr_b16 (hardware transpose, gfx950); fp32 fragments are
// plain ds_read_b32 (one pixel per lane-half).  Partial slabs go to a caller workspace and are summed in
// a fixed order by reduce_kernel -> bitwise reproducible gradients.
#include <stdlib.h>

#include <algorithm>

#include "common.h"

namespace {

struct WView { const char* p; int C, H, W, oy, ox; };

#ifdef PDMA_STAMPS
void* g_wgrad_debug = nullptr;   // diagnostic build only (unet_debug_set_buffer_wgrad)
#endif
struct WgradParams {
  void* debug;     // diagnostic build: s_memtime sums per wave (tools/wgrad_stamps.py); nullptr otherwise
  WView rt;        // row tensor, geometry = frame
  WView ct[2];     // column tensor(s); channel tile below ct[0].C reads ct[0], else ct[1]
  int N, H, W;     // frame (pixel grid of the row tensor)
  int Crow, Ccol;  // GEMM rows / cols (multiples of 64)
  float* partial;  // [split][TAPS][Crow][Ccol]
  int split, tilesPerSplit, tilesX, tilesY, nR, nC;
  int xcd_chunk;   // > 0: XCD-aware block order (wgrad_block_coords), blocks per XCD; 0: plain order
};

// Block -> (split, row tile, column tile).  Workgroups go round-robin over the 8 XCDs (blockIdx % 8), each with its own
// L2.  XCD-aware order: XCD x owns the contiguous range [x*chunk, (x+1)*chunk) of the split-major list, i.e. whole
// pixel ranges with ALL their (row, column) channel tiles -- the blocks that stream the same dY / X pixels run on one
// L2 at the same time, so those bytes leave HBM/MALL once (plain order: the column tiles of one pixel range sit on
// different XCDs and dY is fetched up to 8 times).  With fewer than 8 splits an XCD takes an 8x8 square of tiles.
__device__ inline bool wgrad_block_coords(const WgradParams& P, int& sp, int& rtile, int& ctile) {
  int b = blockIdx.x;
  const int T = P.nR * P.nC;
  if (P.xcd_chunk > 0) {
    const int q = b >> 3;
    b = (b & 7) * P.xcd_chunk + q;
    if (q >= P.xcd_chunk || b >= P.split * T) return false;
    sp = b / T;
    int t = b - sp * T;
    if (((P.nR | P.nC) & 7) == 0) {
      const int tb = t >> 6, in = t & 63, cbn = P.nC >> 3;
      rtile = (tb / cbn) * 8 + (in >> 3);
      ctile = (tb % cbn) * 8 + (in & 7);
    } else {
      ctile = t % P.nC;
      rtile = t / P.nC;
    }
    return true;
  }
  ctile = b % P.nC;  b /= P.nC;
  rtile = b % P.nR;
  sp = b / P.nR;
  return true;
}

template <typename T, int TAPS>
struct WCfg {
  static constexpr int S = (TAPS == 9) ? 1 : 2;
  static constexpr int R = (TAPS == 9) ? 3 : 2;
  static constexpr int PAD = (TAPS == 9) ? 1 : 0;
  static constexpr int TH = (TAPS == 9) ? 8 : 4, TW = 16;
  static constexpr int NPIX = TH * TW;
  static constexpr int HH = (TH - 1) * S + R, HW = (TW - 1) * S + R;
  static constexpr int ROWB = 64 * ET<T>::ES;                       // 64 channels per tile row
  static constexpr int STR = (sizeof(T) == 2) ? ROWB + 64 : ROWB + 16;  // bf16: 192 B (tr-read conflict free)
  static constexpr int PPR = ROWB / 16;
  static constexpr int R_BYTES = NPIX * STR;
  static constexpr int C_BYTES = HH * HW * STR;
  static constexpr int LDS = R_BYTES + C_BYTES;
  static constexpr int NRP = (NPIX * PPR + 255) / 256;
  static constexpr int NCP = (HH * HW * PPR + 255) / 256;
};

__device__ inline bf16x8 tr_frag(const char* base, int off0, int off1) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + off0));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + off1));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

template <typename T, int TAPS>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? 2 : 1) void wgrad_kernel(const WgradParams P) {
  using C = WCfg<T, TAPS>;
  using E = ET<T>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sR = smem;
  char* const sC = smem + C::R_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave & 1, wc = wave >> 1;
  const int l31 = lane & 31, hh = lane >> 5;

  int sp, rtile, ctile;
  if (!wgrad_block_coords(P, sp, rtile, ctile)) return;

  // column source for this channel tile
  int cch = ctile * 64;
  const WView CS = (cch < P.ct[0].C) ? P.ct[0] : P.ct[1];
  if (cch >= P.ct[0].C) cch -= P.ct[0].C;
  const int rch = rtile * 64;

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // ---- per-lane fragment addressing
  // bf16 (ds_read_b64_tr_b16): group g = lane>>4 covers channel block 16*(g&1) and k base 8*(g>>1);
  // lane i=lane&15 supplies the address of pixel row q=i>>2, channels 4*(i&3)..+3 and receives channel i.
  // fp32: lane reads pixel (2*step + hh), channel l31.
  int r_lane, c_lane, kq;
  if constexpr (sizeof(T) == 2) {
    const int g = lane >> 4, i = lane & 15;
    kq = 8 * (g >> 1) + (i >> 2);                       // pixel offset inside the 16-pixel k group (first read)
    const int chb = (16 * (g & 1) + 4 * (i & 3)) * 2;   // channel byte offset inside the wave's 32 channels
    r_lane = wr * 64 + chb;
    c_lane = wc * 64 + chb;
  } else {
    kq = hh;
    r_lane = (wr * 32 + l31) * 4;
    c_lane = (wc * 32 + l31) * 4;
  }

  const int ntiles = P.N * P.tilesY * P.tilesX;
  const int t_begin = sp * P.tilesPerSplit;
  const int t_end = min(t_begin + P.tilesPerSplit, ntiles);

  for (int tile = t_begin; tile < t_end; ++tile) {
    int t = tile;
    const int txi = t % P.tilesX;  t /= P.tilesX;
    const int tyi = t % P.tilesY;
    const int n = t / P.tilesY;
    const int ty0 = tyi * C::TH, tx0 = txi * C::TW;

    // ---- stage (global -> regs -> LDS), zero outside the tensors
    u32x4 rreg[C::NRP], creg[C::NCP];
#pragma unroll
    for (int i = 0; i < C::NRP; ++i) {
      const int id = tid + i * 256;
      rreg[i] = u32x4{0u, 0u, 0u, 0u};
      if (id < C::NPIX * C::PPR) {
        const int pix = id / C::PPR, part = id % C::PPR;
        const int y = ty0 + pix / C::TW, x = tx0 + pix % C::TW;
        if (y < P.H && x < P.W) {
          const size_t e = ((size_t)(n * P.rt.H + y) * P.rt.W + x) * P.rt.C + rch;
          rreg[i] = *reinterpret_cast<const u32x4*>(P.rt.p + e * E::ES + part * 16);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < C::NCP; ++i) {
      const int id = tid + i * 256;
      creg[i] = u32x4{0u, 0u, 0u, 0u};
      if (id < C::HH * C::HW * C::PPR) {
        const int pix = id / C::PPR, part = id % C::PPR;
        const int hy = pix / C::HW, hx = pix - hy * C::HW;
        const int y = ty0 * C::S + hy - C::PAD - CS.oy, x = tx0 * C::S + hx - C::PAD - CS.ox;
        if (y >= 0 && y < CS.H && x >= 0 && x < CS.W) {
          const size_t e = ((size_t)(n * CS.H + y) * CS.W + x) * CS.C + cch;
          creg[i] = *reinterpret_cast<const u32x4*>(CS.p + e * E::ES + part * 16);
        }
      }
    }
    __syncthreads();   // previous tile fully consumed
#pragma unroll
    for (int i = 0; i < C::NRP; ++i) {
      const int id = tid + i * 256;
      if (id < C::NPIX * C::PPR)
        *reinterpret_cast<u32x4*>(sR + (id / C::PPR) * C::STR + (id % C::PPR) * 16) = rreg[i];
    }
#pragma unroll
    for (int i = 0; i < C::NCP; ++i) {
      const int id = tid + i * 256;
      if (id < C::HH * C::HW * C::PPR)
        *reinterpret_cast<u32x4*>(sC + (id / C::PPR) * C::STR + (id % C::PPR) * 16) = creg[i];
    }
    __syncthreads();

    // ---- MFMA over the tile's pixels: one k group = one tile row of 16 pixels
#pragma unroll 1
    for (int ty = 0; ty < C::TH; ++ty) {
      if constexpr (sizeof(T) == 2) {
        const int rp = (ty * C::TW + kq) * C::STR + r_lane;
        const bf16x8 fa = tr_frag(sR, rp, rp + 4 * C::STR);
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
          const int ky = tap / C::R, kx = tap % C::R;
          const int cp = ((ty * C::S + ky) * C::HW + kq * C::S + kx) * C::STR + c_lane;
          const bf16x8 fb = tr_frag(sC, cp, cp + 4 * C::S * C::STR);
          acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[tap], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int st = 0; st < 8; ++st) {
          const int px = 2 * st + kq;
          const float fa = *reinterpret_cast<const float*>(sR + (ty * C::TW + px) * C::STR + r_lane);
#pragma unroll
          for (int tap = 0; tap < TAPS; ++tap) {
            const int ky = tap / C::R, kx = tap % C::R;
            const float fb = *reinterpret_cast<const float*>(
                sC + ((ty * C::S + ky) * C::HW + px * C::S + kx) * C::STR + c_lane);
            acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[tap], 0, 0, 0);
          }
        }
      }
    }
  }

  // ---- partial slab: D[row i][col j], col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int col = ctile * 64 + wc * 32 + l31;
#pragma unroll
  for (int tap = 0; tap < TAPS; ++tap) {
    float* o = P.partial + ((size_t)(sp * TAPS + tap) * P.Crow) * P.Ccol;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = rtile * 64 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
      o[(size_t)row * P.Ccol + col] = acc[tap][r];
    }
  }
}

// out[r][c][tap] (fp32, r < rows_out, c < cols_out) = sum_s partial[s][tap][r][c], fixed order.
// One thread per (r, c): slab reads are coalesced along c, the TAPS results of a thread are contiguous.
template <int TAPS>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial,
                                                           float* __restrict__ out, int split, int Crow, int Ccol,
                                                           int rows_out, int cols_out) {
  const long long total = (long long)rows_out * cols_out;
  const size_t plane = (size_t)Crow * Ccol, slab = (size_t)TAPS * plane;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cols_out), r = (int)(i / cols_out);
    const float* p = partial + (size_t)r * Ccol + c;
    float s[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) s[t] = 0.f;
    for (int k = 0; k < split; ++k) {
#pragma unroll
      for (int t = 0; t < TAPS; ++t) s[t] += p[(size_t)k * slab + (size_t)t * plane];
    }
#pragma unroll
    for (int t = 0; t < TAPS; ++t) out[i * TAPS + t] = s[t];
  }
}


// ------------------------------------------------------------------------------------------------------
// wgrad_dma_kernel: bf16 conv3x3 weight gradient with LDS-DMA staging (buffer_load ... lds).
// Same math and tiling as wgrad_kernel<bf16,9>; differences:
//  * the dY tile and the halo'd X patch of tile t+1 are DMA'd into the second LDS buffer while tile t runs
//    its 72 MFMAs per wave -- no staging registers (180 instead of 220 VGPRs), global latency hidden;
//  * LDS rows are the natural 128 B (64 channels) with a 16-byte-piece XOR swizzle, piece ^= ((row>>1)&1)<<2,
//    applied on the DMA's per-lane SOURCE address and on the transposed reads; four consecutive pixel rows
//    of a ds_read_b64_tr_b16 then sit in four different 64-byte bank quarters (conflict-free) and a buffer is
//    39 KB instead of 59 KB, so two blocks still fit a CU with double buffering.
struct WDma {
  static constexpr int TH = 8, TW = 16, NPIX = 128, HH = 10, HW = 18;
  static constexpr int R_INSTR = NPIX * 8 / 64;                 // 16 x 1 KiB
  static constexpr int C_INSTR = (HH * HW * 8 + 63) / 64;       // 23 x 1 KiB (last one half used)
  static constexpr int R_BYTES = R_INSTR * 1024, C_BYTES = C_INSTR * 1024;
  static constexpr int BUF = R_BYTES + C_BYTES;
  static constexpr int NINSTR = R_INSTR + C_INSTR;              // 39
  static constexpr int NDMA = (NINSTR + 3) / 4;                 // 10 per wave (one surplus -> dummy KiB)
  static constexpr int LDS = 2 * BUF + 1024;
};

__device__ inline bf16x8 tr_frag2(const char* base, int off0, int off1) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + off0));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + off1));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

#ifdef PDMA_STAMPS
#define WG_STAMP(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); wg_st[i] += t_ - wg_prev; wg_prev = t_; }
#else
#define WG_STAMP(i)
#endif

template <bool REUSE>
__global__ __launch_bounds__(256, 2) void wgrad_dma_kernel(const WgradParams P) {
  using C = WDma;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave & 1, wc = wave >> 1;
  const int l31 = lane & 31, hh = lane >> 5;

  int sp, rtile, ctile;
  if (!wgrad_block_coords(P, sp, rtile, ctile)) return;
  int cch = ctile * 64;
  const WView CS = (cch < P.ct[0].C) ? P.ct[0] : P.ct[1];
  if (cch >= P.ct[0].C) cch -= P.ct[0].C;
  const int rch = rtile * 64;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // transposed-read lane geometry (see wgrad_kernel): kq = pixel offset in the 16-pixel k group,
  // lane byte offset inside the wave's 64 bytes of channels
  const int g = lane >> 4, i16 = lane & 15;
  const int kq = 8 * (g >> 1) + (i16 >> 2);
  const int chb = (16 * (g & 1) + 4 * (i16 & 3)) * 2;
  const int r_piece = (wr * 64 + chb) >> 4, r_sub = (wr * 64 + chb) & 15;
  const int c_piece = (wc * 64 + chb) >> 4, c_sub = (wc * 64 + chb) & 15;
  const int r_swz = (((kq >> 1) & 1) << 2);                    // row = ty*16 + kq (+4): bit 1 of kq
  int c_base[2][3];
#pragma unroll
  for (int par = 0; par < 2; ++par)
#pragma unroll
    for (int sx = 0; sx < 3; ++sx)
      c_base[par][sx] = (kq + sx) * 128 + ((c_piece ^ (((((kq + sx) >> 1) ^ par) & 1) << 2)) << 4) + c_sub;

  // ---- DMA descriptors: instruction ii = j*4 + wave; < R_INSTR -> dY tile, else X patch
  constexpr unsigned OOB = 0xFFFFFFF0u;
  int d_code[C::NDMA];      // row | piece << 12 | isC << 16    (-1: surplus)
#pragma unroll
  for (int j = 0; j < C::NDMA; ++j) {
    const int ii = j * 4 + wave;
    int code = -1;
    if (ii < C::R_INSTR) {
      const int row = ii * 8 + (lane >> 3), pp = lane & 7;
      code = row | ((pp ^ (((row >> 1) & 1) << 2)) << 12);
    } else if (ii < C::NINSTR) {
      const int q = (ii - C::R_INSTR) * 64 + lane;
      const int row = q >> 3, pp = q & 7;
      if (row < C::HH * C::HW) code = row | ((pp ^ (((row >> 1) & 1) << 2)) << 12) | (1 << 16);
    }
    d_code[j] = code;
  }
  const unsigned r_img = (unsigned)P.rt.H * P.rt.W * P.rt.C * 2u;
  const unsigned c_img = (unsigned)CS.H * CS.W * CS.C * 2u;
  typedef __attribute__((address_space(3))) void lds_void;

  auto dma = [&](int tile, int buf) {
    int t = tile;
    const int txi = t % P.tilesX;  t /= P.tilesX;
    const int tyi = t % P.tilesY;
    const int n = t / P.tilesY;
    const int ty0 = tyi * C::TH, tx0 = txi * C::TW;
    const __amdgpu_buffer_rsrc_t rr =
        __builtin_amdgcn_make_buffer_rsrc((void*)(P.rt.p + (size_t)n * r_img), (short)0, (int)r_img, 0x00020000);
    const __amdgpu_buffer_rsrc_t rc =
        __builtin_amdgcn_make_buffer_rsrc((void*)(CS.p + (size_t)n * c_img), (short)0, (int)c_img, 0x00020000);
#pragma unroll
    for (int j = 0; j < C::NDMA; ++j) {
      const int ii = j * 4 + wave;                             // wave-uniform
      const int code = d_code[j];
      const int row = code & 4095, piece = (code >> 12) & 15;
      char* dst = smem + 2 * C::BUF;                           // dummy KiB for the surplus instruction
      if (ii < C::R_INSTR) {
        dst = smem + buf * C::BUF + ii * 1024;
        const int y = ty0 + (row >> 4), x = tx0 + (row & 15);
        const bool ok = y < P.H && x < P.W;
        const unsigned vo = ok ? (unsigned)(((y * P.rt.W + x) * P.rt.C + rch) * 2 + piece * 16) : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rr, (lds_void*)dst, 16, vo, 0, 0, 0);
      } else {
        if (ii < C::NINSTR) dst = smem + buf * C::BUF + C::R_BYTES + (ii - C::R_INSTR) * 1024;
        const int hy = row / C::HW, hx = row - hy * C::HW;
        const int y = ty0 + hy - 1 - CS.oy, x = tx0 + hx - 1 - CS.ox;
        const bool ok = code >= 0 && y >= 0 && y < CS.H && x >= 0 && x < CS.W;
        const unsigned vo = ok ? (unsigned)(((y * CS.W + x) * CS.C + cch) * 2 + piece * 16) : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rc, (lds_void*)dst, 16, vo, 0, 0, 0);
      }
    }
  };

  const int ntiles = P.N * P.tilesY * P.tilesX;
  const int t_begin = sp * P.tilesPerSplit;
  const int t_end = min(t_begin + P.tilesPerSplit, ntiles);

#ifdef PDMA_STAMPS
  unsigned long long wg_st[4] = {0, 0, 0, 0}, wg_prev = __builtin_amdgcn_s_memtime();
  const unsigned long long wg_t0 = wg_prev, wg_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  if (t_begin < t_end) dma(t_begin, 0);
  for (int tile = t_begin; tile < t_end; ++tile) {
    const int buf = (tile - t_begin) & 1;
    WG_STAMP(3)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's share of tile `tile` has landed
    WG_STAMP(0)
    __builtin_amdgcn_s_barrier();                               // ... everyone's has; buffer buf^1 is free
    WG_STAMP(1)
    if (tile + 1 < t_end) dma(tile + 1, buf ^ 1);
    WG_STAMP(2)
    const char* sR = smem + buf * C::BUF;
    const char* sC = sR + C::R_BYTES;
    if constexpr (REUSE) {
      // the X fragment of (tile row ty, tap row r, tap column s) is the fragment of (ty+1, r-1, s): keep the
      // three live patch rows x three column shifts in registers and fetch only the new patch row per ty
      // (30 + 8 instead of 72 + 8 transposed fragment reads per tile: the LDS array was the limiter)
      // patch row stride 18 is even, so the swizzle bit of row prow*18 + kq + sx is ((kq+sx)>>1 ^ prow) & 1:
      // two lane addresses per column shift (even / odd patch row), the row itself is an immediate offset
      auto load_c = [&](int prow, int sx) {
        const int cp = c_base[prow & 1][sx] + prow * (C::HW * 128);
        return tr_frag2(sC, cp, cp + 4 * 128);                  // rows +4: same swizzle bit
      };
      bf16x8 fb[3][3];
#pragma unroll
      for (int sx = 0; sx < 3; ++sx) { fb[0][sx] = load_c(0, sx); fb[1][sx] = load_c(1, sx); }
#pragma unroll
      for (int ty = 0; ty < C::TH; ++ty) {
#pragma unroll
        for (int sx = 0; sx < 3; ++sx) fb[(ty + 2) % 3][sx] = load_c(ty + 2, sx);
        const int rrow = ty * C::TW + kq;
        const int rp = rrow * 128 + ((r_piece ^ r_swz) << 4) + r_sub;
        const bf16x8 fa = tr_frag2(sR, rp, rp + 4 * 128);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
          acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb[(ty + tap / 3) % 3][tap % 3], acc[tap], 0, 0, 0);
      }
    } else {
#pragma unroll 4
      for (int ty = 0; ty < C::TH; ++ty) {
        const int rrow = ty * C::TW + kq;
        const int rp = rrow * 128 + ((r_piece ^ r_swz) << 4) + r_sub;
        const bf16x8 fa = tr_frag2(sR, rp, rp + 4 * 128);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int crow = (ty + tap / 3) * C::HW + kq + tap % 3;
          const int cp = crow * 128 + ((c_piece ^ (((crow >> 1) & 1) << 2)) << 4) + c_sub;
          const bf16x8 fb = tr_frag2(sC, cp, cp + 4 * 128);       // rows +4: same swizzle bit
          acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[tap], 0, 0, 0);
        }
      }
    }
  }

#ifdef PDMA_STAMPS
  WG_STAMP(3)
  if (P.debug && lane == 0 && blockIdx.x < 512) {
    unsigned long long* o = (unsigned long long*)P.debug + ((size_t)blockIdx.x * 4 + wave) * 8;
    for (int i = 0; i < 4; ++i) o[i] = wg_st[i];
    o[4] = (unsigned long long)(t_end - t_begin);
    o[5] = ((__builtin_amdgcn_s_memtime() - wg_t0) << 20) / (__builtin_amdgcn_s_memrealtime() - wg_r0 + 1);
  }
#endif
  const int col = ctile * 64 + wc * 32 + l31;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    float* o = P.partial + ((size_t)(sp * 9 + tap) * P.Crow) * P.Ccol;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = rtile * 64 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
      o[(size_t)row * P.Ccol + col] = acc[tap][r];
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// wgrad16_kernel<NRH, NCH, PAIRED> (round 3): the bf16 conv3x3 weight gradient on v_mfma_f32_16x16x32_bf16.
//
// Why: the 32x32x16 kernel above ran at 1.23-1.58 GHz in-kernel (profiles/r02_wgrad_stamps.txt) -- the chip holds a
// higher clock on the 16x16x32 shape (MI355X_MICROARCH.md, DVFS give-back item 7; stamped here: 1.76-1.90 GHz) -- and
// ~36 % of a tile went to its ten one-KiB LDS-DMA issues per wave.  Here:
//  * one 512-thread block per CU owns 128 x 64 (NRH=2, NCH=1) or 64 x 128 (1, 2) gradient channels x 9 taps: the X patch
//    (or the dY tile) is staged once for twice the MFMAs -- 62 instead of 78 KiB of LDS-DMA per 18.9 MFLOP;
//  * K = 32 pixels per MFMA = one 32-pixel tile ROW (tile 4 x 32; PAIRED, W <= 16: 16 pixels of image n + 16 of image
//    n+1), so the X fragment of (patch row p, column shift s) still serves the three (tile row, tap row) pairs with
//    ty + r = p; the loop walks PATCH rows: per row 6 X fragments + 2 dY fragments (a 4-slot ring of dY rows), 12..36
//    MFMAs; 36 independent accumulator chains (144 VGPRs) as before;
//  * LDS rows are the natural 128 B (64 channels of one pixel); a patch row is padded to PW = 48 / 40 / 32 LDS rows so
//    that the 16-byte-piece XOR swizzle -- bits 1 and 3 of the LDS row index select one of four 32-byte bank groups, on
//    the DMA's per-lane SOURCE address and on the transposed reads -- leaves the 8 pixel rows x 32 B of a
//    ds_read_b64_tr_b16 half-wave in 8 different bank groups for every column shift (SQ_LDS_BANK_CONFLICT = 0), and
//    every fragment address is one of 12 (PW/8 even) or 24 per-lane constants + an immediate: NO address arithmetic in
//    the loop.  That matters on this MFMA shape: a 16-cycle MFMA holds the SIMD's vector issue for 8 cycles, so two
//    waves per SIMD leave 8 cycles per MFMA for everything else; the first version (2 VALU per fragment read) was
//    vector-ISSUE-bound (SQ counters: 1.43 VALU per MFMA, 54 % MFMA busy);
//  * the two waves of a SIMD issue their LDS-DMAs at opposite ends of a tile (see the loop).
// Same split-K slabs / ordered reduction as above: bitwise reproducible, and bit-identical to the 32x32x16 kernel's sums
// only up to fp32 summation order (K is walked in a different order).
template <int NRH, int NCH, bool PAIRED>
struct W16 {
  static constexpr int PW = PAIRED ? 32 : (NCH == 1 ? 48 : 40); // LDS rows per patch row (18 / 34 of them used)
  static constexpr bool PX = ((PW / 8) & 1) != 0;               // bit 3 of the LDS row flips with the patch row
  static constexpr int IMGROWS = 6 * PW;                        // LDS rows of one image's patch
  static constexpr int XROWS = (PAIRED ? 2 : 1) * IMGROWS;
  static constexpr int R_HALF = 128 * 128, C_HALF = XROWS * 128;
  static constexpr int IPR = PAIRED ? 3 : 5;                    // DMA pieces (8 LDS rows) per patch row that hold data
  static constexpr int R_INSTR = 16, C_INSTR = (PAIRED ? 2 : 1) * 6 * IPR;   // one-KiB DMA instructions per half
  static constexpr int BUF = NRH * R_HALF + NCH * C_HALF;
  static constexpr int LDS = 2 * BUF;                           // (paired: exactly the CU's 160 KiB)
  static constexpr int WR = 2 * NRH, WC = 2 * NCH;              // waves along gradient rows / columns (32 channels each)
  static_assert(WR * WC == 8, "eight waves");
  static_assert(LDS <= 160 * 1024, "LDS");
  static_assert(5 * PW * 128 + 12 * 128 + 512 < 65536, "fragment offsets are 16-bit immediates");
};

template <int NRH, int NCH, bool PAIRED>
__global__ __launch_bounds__(512, 1) void wgrad16_kernel(const WgradParams P) {
  using C = W16<NRH, NCH, PAIRED>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void lds_void;
  constexpr unsigned OOB = 0xFFFFFFF0u;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // wave-uniform by construction: keep it scalar
  const int wr = wave % C::WR, wc = wave / C::WR;

  int sp, rtile, ctile;
  if (!wgrad_block_coords(P, sp, rtile, ctile)) return;
  sp = __builtin_amdgcn_readfirstlane(sp);
  rtile = __builtin_amdgcn_readfirstlane(rtile);
  ctile = __builtin_amdgcn_readfirstlane(ctile);
  const int rch = rtile * 64 * NRH;
  WView CS[NCH];
  int cch[NCH];
#pragma unroll
  for (int h = 0; h < NCH; ++h) {
    const int c = (ctile * NCH + h) * 64;
    const bool first = c < P.ct[0].C;
    CS[h] = first ? P.ct[0] : P.ct[1];
    cch[h] = first ? c : c - P.ct[0].C;
  }

  f32x4 acc[9][2][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[t][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- transposed-read lane geometry.  16x16x32 operands: lane = (k group g = lane >> 4: k = 8g .. 8g+7) x (channel
  // lane & 15).  ds_read_b64_tr_b16: lane i of a 16-lane group ADDRESSES pixel row (i >> 2) of the group's 4 rows,
  // channels 4 (i & 3) .. +3, and RECEIVES the four rows of channel i; two reads (rows +0..3, +4..7) = the 8 k values.
  const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, cq = (i16 & 3) * 8;
  int a_off[2];                                   // dY: LDS row = ty*32 + 8g + 4j + q -> swizzle (q >> 1) | (g & 1) << 1
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const int cg = (wr & 1) * 2 + cb;
    a_off[cb] = (wr >> 1) * C::R_HALF + (8 * g + q) * 128 + ((cg ^ ((q >> 1) | ((g & 1) << 1))) << 5) + cq;
  }
  // X: LDS row = image * IMGROWS + prow * PW + 8g' + (s + 4j + q); swizzle bit 0 = bit 1 of (s + 4j + q), bit 1 =
  // bit0(g') ^ bit 3 of (s + 4j + q) [^ (prow & 1) when PW/8 is odd: a second set of constants for the odd patch rows]
  const int gp = PAIRED ? (g & 1) : g;
  int b_off[C::PX ? 2 : 1][3][2][2];              // [patch-row parity][shift][read][16-channel block]
#pragma unroll
  for (int par = 0; par < (C::PX ? 2 : 1); ++par)
#pragma unroll
    for (int sx = 0; sx < 3; ++sx)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          const int t = sx + 4 * j + q;
          const int f = ((t >> 1) & 1) | ((((gp & 1) ^ (t >> 3) ^ par) & 1) << 1);
          const int cg = (wc & 1) * 2 + nb;
          b_off[par][sx][j][nb] = NRH * C::R_HALF + (wc >> 1) * C::C_HALF +
                                  ((PAIRED ? (g >> 1) * C::IMGROWS : 0) + 8 * gp + t) * 128 + ((cg ^ f) << 5) + cq;
        }

  // ---- DMA.  One instruction (piece) = 8 LDS rows (pixels) x 128 B; lane -> row lane >> 3, 16-byte position lane & 7.
  // A piece belongs to one wave, so every quantity except the lane's (row, position) is wave-uniform and lives in SGPRs:
  // per piece a scalar base (soffset), a per-lane constant (pixel * channel stride + swizzled piece; bit 3 of the LDS row
  // = a property of the piece) and a column range check.
  // The two waves of a SIMD (w, w + 4) issue their pieces at opposite ends of a tile (see the loop): the EARLY waves
  // (4-7) before their MFMAs -- which they therefore start late -- the LATE waves (0-3) after theirs.  The early waves
  // take fewer pieces (NE each, from the end of the list), so that both waves of a SIMD reach the end of the tile
  // together (W16_NE: stamped / timed at 4, 5, 6, 8)
#ifndef W16_NE
#define W16_NE 8
#endif
  constexpr int NT = NRH * C::R_INSTR + NCH * C::C_INSTR;      // pieces per tile: 62 (128x64 wide), 68 (paired), 76 (64x128)
  constexpr int NE = W16_NE;
  constexpr int NLATE = (NT - 4 * NE + 3) / 4;
  // (measured: the waves that compute FIRST must be the ones that win the SIMD's issue arbitration -- the older waves
  //  0-3; the roles swapped, or the early waves raised with s_setprio, cost 8-13 %)
  const bool late_wave = wave < 4;
  const unsigned r_img = (unsigned)P.rt.H * P.rt.W * P.rt.C * 2u;
  unsigned c_img[NCH];
#pragma unroll
  for (int h = 0; h < NCH; ++h) c_img[h] = (unsigned)CS[h].H * CS[h].W * CS[h].C * 2u;

  // the tile being FETCHED: (image [pair], tile row, tile column), advanced by scalar compares -- no divisions in the loop
  int d_txi, d_tyi, d_n;
  {
    int t = sp * P.tilesPerSplit;
    d_txi = __builtin_amdgcn_readfirstlane(t % P.tilesX);  t = __builtin_amdgcn_readfirstlane(t / P.tilesX);
    d_tyi = __builtin_amdgcn_readfirstlane(t % P.tilesY);
    d_n = __builtin_amdgcn_readfirstlane(t / P.tilesY);
  }
  auto next_tile = [&]() {
    if (++d_txi == P.tilesX) {
      d_txi = 0;
      if (++d_tyi == P.tilesY) { d_tyi = 0; ++d_n; }
    }
  };
  auto dma = [&](int buf) {
    const int n0 = d_n * (PAIRED ? 2 : 1), ty0 = d_tyi * 4, tx0 = PAIRED ? 0 : d_txi * 32;
    const int nimg = (PAIRED && n0 + 1 < P.N) ? 2 : 1;
    // the lane-dependent constants are re-derived per tile (a few VALU ops) instead of living in VGPRs across the MFMA
    // loop: the opaque copy keeps LICM from hoisting them
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int lx = ln >> 3;
    const int pz = (ln & 7) ^ (((ln >> 4) & 1) << 1);           // piece ^ (bit 1 of the row) << 1
    auto x_piece = [&](int i2) {                                // piece i2 of the X patch (both halves counted through)
      const int h = NCH == 1 ? 0 : i2 / C::C_INSTR, li = NCH == 1 ? i2 : i2 % C::C_INSTR;
      // (scalar selects: indexing CS[] with the run-time h would put the view into scratch memory)
      const bool h1 = NCH > 1 && h == 1;
      WView S;
      S.p = h1 ? CS[NCH - 1].p : CS[0].p;  S.C = h1 ? CS[NCH - 1].C : CS[0].C;  S.H = h1 ? CS[NCH - 1].H : CS[0].H;
      S.W = h1 ? CS[NCH - 1].W : CS[0].W;  S.oy = h1 ? CS[NCH - 1].oy : CS[0].oy;  S.ox = h1 ? CS[NCH - 1].ox : CS[0].ox;
      const unsigned cimg = h1 ? c_img[NCH - 1] : c_img[0];
      const int cchh = h1 ? cch[NCH - 1] : cch[0];
      const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(S.p + (size_t)n0 * cimg), (short)0, (int)(cimg * nimg), 0x00020000);
      const int img = PAIRED ? li / (6 * C::IPR) : 0, lr = PAIRED ? li % (6 * C::IPR) : li;
      const int prow = lr / C::IPR, pc0 = (lr % C::IPR) * 8;    // patch row, first of the 8 patch columns
      const int row0 = img * C::IMGROWS + prow * C::PW + pc0;   // first LDS row of the piece
      const int y = ty0 + prow - 1 - S.oy, x0 = tx0 + pc0 - 1 - S.ox;
      const bool rowok = y >= 0 && y < S.H && img < nimg;
      // scalar part: the image row (never negative once rowok); the column (x0 may be -1) stays in the lane part
      const unsigned so = (unsigned)img * cimg + (unsigned)((y * S.W * S.C + cchh) * 2);
      const unsigned lane_off = (unsigned)(x0 + lx) * (unsigned)(S.C * 2) + (unsigned)((pz ^ (((row0 >> 3) & 1) << 2)) << 4);
      const bool colok = (unsigned)(x0 + lx) < (unsigned)S.W && pc0 + lx < (PAIRED ? 18 : 34);
      const unsigned vo = (rowok && colok) ? lane_off : OOB;
      char* dst = smem + buf * C::BUF + NRH * C::R_HALF + h * C::C_HALF + row0 * 128;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rc, (lds_void*)dst, 16, vo, rowok ? so : 0u, 0, 0);
    };
    const __amdgpu_buffer_rsrc_t rr =
        __builtin_amdgcn_make_buffer_rsrc((void*)(P.rt.p + (size_t)n0 * r_img), (short)0, (int)(r_img * nimg), 0x00020000);
    const unsigned lstr = (unsigned)lx * (unsigned)(P.rt.C * 2);
    auto y_piece = [&](int li) {                                // piece li of the dY tile: LDS rows li*8 .. +7
      const int h = li / C::R_INSTR, l16 = li % C::R_INSTR;
      const int ty = l16 >> 2, c0 = (l16 & 3) * 8;              // tile row, first of the 8 columns (0..31)
      const int img = PAIRED ? (c0 >> 4) : 0;
      const int y = ty0 + ty, x0 = tx0 + (PAIRED ? (c0 & 15) : c0);
      const bool rowok = y < P.H && img < nimg;
      const unsigned so = (unsigned)img * r_img + (unsigned)(((y * P.rt.W + x0) * P.rt.C + rch + h * 64) * 2);
      const unsigned lane_off = lstr + (unsigned)((pz ^ ((l16 & 1) << 2)) << 4);      // (bit 3 of the LDS row = l16 & 1)
      const unsigned vo = (rowok && x0 + lx < P.W) ? lane_off : OOB;
      char* dst = smem + buf * C::BUF + h * C::R_HALF + l16 * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rr, (lds_void*)dst, 16, vo, rowok ? so : 0u, 0, 0);
    };
    // pieces 0 .. NT-1 = the dY tile, then the X patch; the late waves take the first NT - 4 NE, the early waves the rest
    if (late_wave) {
#pragma unroll
      for (int j = 0; j < NLATE; ++j) {
        const int idx = (wave & 3) * NLATE + j;
        if (idx >= NT - 4 * NE) break;
        if (idx < NRH * C::R_INSTR) y_piece(idx);
        else x_piece(idx - NRH * C::R_INSTR);
      }
    } else {
#pragma unroll
      for (int j = 0; j < NE; ++j) {
        const int idx = NT - 4 * NE + (wave & 3) * NE + j;
        if (idx < NRH * C::R_INSTR) y_piece(idx);
        else x_piece(idx - NRH * C::R_INSTR);
      }
    }
  };

  const int ngroups = PAIRED ? (P.N + 1) / 2 : P.N;
  const int ntiles = ngroups * P.tilesY * P.tilesX;
  const int t_begin = sp * P.tilesPerSplit;
  const int t_end = min(t_begin + P.tilesPerSplit, ntiles);

#ifdef PDMA_STAMPS
  unsigned long long wg_st[4] = {0, 0, 0, 0}, wg_prev = __builtin_amdgcn_s_memtime();
  const unsigned long long wg_t0 = wg_prev, wg_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  if (t_begin < t_end) dma(0);
  for (int tile = t_begin; tile < t_end; ++tile) {
    const int buf = (tile - t_begin) & 1;
    WG_STAMP(3)
    next_tile();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's share of tile `tile` has landed
    WG_STAMP(0)
    __builtin_amdgcn_s_barrier();                               // ... everyone's has; buffer buf^1 is free
    WG_STAMP(1)
    // The next tile's DMAs: a wave in its burst of 8-10 one-KiB issues (140-250 cycles EACH beside the partner's MFMAs
    // and fragment reads, stamped) feeds no MFMAs.  The two waves of a SIMD (w, w + 4) therefore issue at opposite ends
    // of a tile -- the X waves right here, before their MFMAs, the dY waves after theirs -- so that one of the two
    // always has MFMAs to issue (stamps: the earlier the dY waves issued, the more the two bursts overlapped).
    const bool more = tile + 1 < t_end;
#ifndef W16_NO_DMA
    if (more && !late_wave) { dma(buf ^ 1); WG_STAMP(2) }
#endif
    const char* sb = smem;                                      // (a_off / b_off carry the buffer's offset)
    auto ldA = [&](int ty, int cb) {
      const int o = a_off[cb] + ty * 4096;
      return tr_frag2(sb, o, o + 512);
    };
    auto ldB = [&](int prow, int sx, int nb) {
      const int par = C::PX ? (prow & 1) : 0;
      return tr_frag2(sb, b_off[par][sx][0][nb] + prow * (C::PW * 128), b_off[par][sx][1][nb] + prow * (C::PW * 128));
    };
    bf16x8 fa[4][2], fb[3][2];                                  // dY rows: ring of 4 (rows p, p-1, p-2 live, p+1 in flight)
    // A tile = 18 slots (patch row p, column shift sx) of 4 n_r(p) MFMAs, n_r = 1 2 3 3 2 1.  Every fragment is requested
    // two or three slots before its first use, into registers retired by then: slot (p, 0) fetches the X fragments of
    // (p, 2) and dY row p + 1, slot (p, 1) those of (p + 1, 0), slot (p, 2) those of (p + 1, 1).  Within a slot the
    // ds_reads are INTERLEAVED one per MFMA (sched_group_barrier): a ds_read between two MFMAs is nearly free, a burst
    // of eight after a group stalls the matrix pipe of a wave that is alone on its SIMD while its partner issues DMAs
    // (hipcc left alone sinks the reads to their first use and waits there: half the MFMA rate).
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) fa[0][cb] = ldA(0, cb);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) fb[0][nb] = ldB(0, 0, nb);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) fb[1][nb] = ldB(0, 1, nb);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < 6; ++p) {
#pragma unroll
      for (int sx = 0; sx < 3; ++sx) {
        int nld = 0;
        if (sx == 0) {
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) fb[2][nb] = ldB(p, 2, nb);
          nld += 4;
          if (p + 1 < 4) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) fa[(p + 1) & 3][cb] = ldA(p + 1, cb);
            nld += 4;
          }
        } else if (p + 1 < 6) {
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) fb[sx - 1][nb] = ldB(p + 1, sx - 1, nb);
          nld += 4;
        }
        int nmf = 0;
#pragma unroll
        for (int r = 2; r >= 0; --r) {                          // oldest dY row first
          const int ty = p - r;
          if (ty < 0 || ty > 3) continue;
#pragma unroll
          for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
              acc[r * 3 + sx][cb][nb] =     // D[X channel][dY channel]: a lane's 4 registers = 4 consecutive X channels
                  __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[sx][nb], fa[ty & 3][cb], acc[r * 3 + sx][cb][nb], 0, 0, 0);
          nmf += 4;
        }
        // pipeline of the slot: MFMA, ds_read, MFMA, ds_read, ... then whatever is left of either
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          if (i < nmf) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (i < nld) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#ifndef W16_NO_DMA
    if (more && late_wave) { WG_STAMP(3) dma(buf ^ 1); WG_STAMP(2) }
#endif
    // the fragment addresses move to the other buffer IN PLACE (a second set of registers does not fit beside 200
    // accumulator / fragment registers)
    {
      const int delta = buf ? -C::BUF : C::BUF;
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) a_off[cb] += delta;
#pragma unroll
      for (int par = 0; par < (C::PX ? 2 : 1); ++par)
#pragma unroll
        for (int sx = 0; sx < 3; ++sx)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) b_off[par][sx][j][nb] += delta;
    }
  }

#ifdef PDMA_STAMPS
  WG_STAMP(3)
  const unsigned long long wg_e0 = __builtin_amdgcn_s_memtime();
#endif
  // ---- partial slab [tap][dY channel][X channel].  The MFMAs ran with A = X, B = dY: D rows (lane >> 4) * 4 + reg = 4
  // CONSECUTIVE X channels, column lane & 15 = the dY channel -> one 16-byte store per accumulator (36 per lane; with the
  // operands the other way round it took 144 four-byte stores, and the store tail is issue-bound)
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    float* o = P.partial + ((size_t)(sp * 9 + tap) * P.Crow) * P.Ccol;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const int row = rch + wr * 32 + cb * 16 + i16;
        const int col = ctile * 64 * NCH + wc * 32 + nb * 16 + g * 4;
        *reinterpret_cast<f32x4*>(o + (size_t)row * P.Ccol + col) = acc[tap][cb][nb];
      }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef PDMA_STAMPS
  if (P.debug && lane == 0 && blockIdx.x < 256) {
    unsigned long long* o = (unsigned long long*)P.debug + ((size_t)blockIdx.x * 8 + wave) * 8;
    for (int i = 0; i < 4; ++i) o[i] = wg_st[i];
    o[4] = (unsigned long long)(t_end - t_begin);
    o[5] = ((__builtin_amdgcn_s_memtime() - wg_t0) << 20) / (__builtin_amdgcn_s_memrealtime() - wg_r0 + 1);
    o[6] = __builtin_amdgcn_s_memtime() - wg_e0;               // partial-slab stores, drained
    o[7] = __builtin_amdgcn_s_memtime() - wg_t0;
  }
#endif
}

// Many splits, few outputs (the 64-channel layers: 512 slabs of 9x64x64): one block per (row, 64 columns,
// tap), 4 slab-lanes per output, combined in a fixed order through LDS.
template <int TAPS>
__global__ __launch_bounds__(256) void wgrad_reduce_wide_kernel(const float* __restrict__ partial,
                                                                float* __restrict__ out, int split, int Crow,
                                                                int Ccol, int rows_out, int cols_out) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, ks = threadIdx.x >> 6;
  int b = blockIdx.x;
  const int tap = b % TAPS;  b /= TAPS;
  const int nchunk = (cols_out + 63) / 64;
  const int c = (b % nchunk) * 64 + cl, r = b / nchunk;
  const size_t plane = (size_t)Crow * Ccol, slab = (size_t)TAPS * plane;
  float s = 0.f;
  if (c < cols_out) {
    const float* p = partial + (size_t)tap * plane + (size_t)r * Ccol + c;
    int k = ks;
    for (; k + 12 < split; k += 16) {
      const float a0 = p[(size_t)k * slab], a1 = p[(size_t)(k + 4) * slab], a2 = p[(size_t)(k + 8) * slab],
                  a3 = p[(size_t)(k + 12) * slab];
      s += a0; s += a1; s += a2; s += a3;
    }
    for (; k < split; k += 4) s += p[(size_t)k * slab];
  }
  red[ks][cl] = s;
  __syncthreads();
  if (ks == 0 && c < cols_out)
    out[((size_t)r * cols_out + c) * TAPS + tap] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}

struct Plan { int tilesX, tilesY, nR, nC, split, tilesPerSplit; size_t bytes; };

template <int TAPS>
Plan make_plan(int n, int h, int w, int crow, int ccol) {
  Plan p;
  constexpr int TH = (TAPS == 9) ? 8 : 4;
  p.tilesX = cdiv(w, 16);
  p.tilesY = cdiv(h, TH);
  p.nR = crow / 64;
  p.nC = ccol / 64;
  const long long ntiles = (long long)n * p.tilesY * p.tilesX;
  long long want = 2LL * unet_cu_budget() / ((long long)p.nR * p.nC);      // two 256-thread blocks per CU
  if (want < 1) want = 1;
  if (want > ntiles) want = ntiles;
  p.tilesPerSplit = (int)cdiv64(ntiles, want);
  p.split = (int)cdiv64(ntiles, p.tilesPerSplit);
  p.bytes = (size_t)p.split * TAPS * crow * ccol * sizeof(float);
  return p;
}

// wgrad16_kernel: 128 x 64 (crow % 128 == 0) or 64 x 128 (crow == 64, ccol % 128 == 0, wide frames) channel tiles, 4 x 32
// pixel tiles (PAIRED: two 16-wide images), one block per CU -> 256 blocks
inline int wgrad16_variant(int w, int crow, int ccol) {          // 0: not served; 1: <2,1,wide>; 2: <2,1,paired>; 3: <1,2,wide>
  if (crow % 128 == 0) return w <= 16 ? 2 : 1;
  if (crow == 64 && ccol % 128 == 0 && w > 16) return 3;
  return 0;
}
inline Plan make_plan16(int n, int h, int w, int crow, int ccol) {
  Plan p;
  const int v = wgrad16_variant(w, crow, ccol);
  p.tilesX = v == 2 ? 1 : cdiv(w, 32);
  p.tilesY = cdiv(h, 4);
  p.nR = v == 3 ? crow / 64 : crow / 128;
  p.nC = v == 3 ? ccol / 128 : ccol / 64;
  const long long ntiles = (long long)(v == 2 ? (n + 1) / 2 : n) * p.tilesY * p.tilesX;
  long long want = (long long)unet_cu_budget() / ((long long)p.nR * p.nC);   // one 512-thread block per CU
  if (want < 1) want = 1;
  if (want > ntiles) want = ntiles;
  p.tilesPerSplit = (int)cdiv64(ntiles, want);
  p.split = (int)cdiv64(ntiles, p.tilesPerSplit);
  p.bytes = (size_t)p.split * 9 * crow * ccol * sizeof(float);
  return p;
}

template <int NRH, int NCH, bool PAIRED>
void launch16(const WgradParams& P, long long blocks, hipStream_t s) {
  using C = W16<NRH, NCH, PAIRED>;
  auto kern = wgrad16_kernel<NRH, NCH, PAIRED>;
  unet_set_max_lds(reinterpret_cast<const void*>(kern), C::LDS);
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), C::LDS, s, P);
}

template <typename T, int TAPS>
int32_t run(WgradParams& P, const Plan& pl, float* out, int rows_out, int cols_out, int kclass, hipStream_t s) {
  using C = WCfg<T, TAPS>;
  auto kern = wgrad_kernel<T, TAPS>;
  unet_set_max_lds(reinterpret_cast<const void*>(kern), C::LDS);
  P.tilesX = pl.tilesX; P.tilesY = pl.tilesY; P.nR = pl.nR; P.nC = pl.nC;
  P.split = pl.split; P.tilesPerSplit = pl.tilesPerSplit;
  long long blocks = (long long)pl.split * pl.nR * pl.nC;
  P.xcd_chunk = 0;
  if (unet_tuning().wgrad_xcd != '0' && blocks >= 16) {         // UNET_WGRAD_XCD=0: plain block order
    P.xcd_chunk = (int)cdiv64(blocks, 8);
    blocks = 8LL * P.xcd_chunk;
  }
  const double flops = 2.0 * P.N * P.H * P.W * (double)P.Crow * P.Ccol * TAPS;
  const char impl = unet_tuning().wgrad_impl;   // UNET_WGRAD_IMPL: 0 register-staged, 1 the 32x32x16 LDS-DMA kernel, 2 that without reuse
  int split_used = pl.split;
  {
    // algorithmic bytes: both activation tensors once + the fp32 gradient once (split-K slabs are overhead, not counted)
    const double alg_bytes = (double)P.N * P.H * P.W * ((double)P.Crow + P.Ccol * (TAPS == 9 ? 1.0 : 4.0)) * sizeof(T) +
                             4.0 * TAPS * rows_out * cols_out;
    // wgrad16_kernel: default for the forms it wins on (A/B per shape, profiles/r03_wgrad16.txt); UNET_WGRAD_IMPL=3
    // forces it wherever it applies, 1 / 2 select the 32x32x16 kernel
    int v16 = 0;
    if constexpr (sizeof(T) == 2 && TAPS == 9) {
      v16 = (impl == 0 || impl == '3') ? wgrad16_variant(P.W, P.Crow, P.Ccol) : 0;
      if (impl == 0 && v16 == 3) v16 = 0;
    }
    ProfScope prof(kclass, flops, s,
                   v16 ? "wgrad16_kernel (+ reduce)"
                       : ((sizeof(T) == 2 && TAPS == 9) ? "wgrad_dma_kernel (+ reduce)" : "wgrad_kernel (+ reduce)"), alg_bytes);
    if constexpr (sizeof(T) == 2 && TAPS == 9) {
      if (v16) {
        const Plan p16 = make_plan16(P.N, P.H, P.W, P.Crow, P.Ccol);
        P.tilesX = p16.tilesX; P.tilesY = p16.tilesY; P.nR = p16.nR; P.nC = p16.nC;
        P.split = p16.split; P.tilesPerSplit = p16.tilesPerSplit;
        split_used = p16.split;
        long long b16 = (long long)p16.split * p16.nR * p16.nC;
        P.xcd_chunk = 0;
        if (unet_tuning().wgrad_xcd != '0' && b16 >= 16) { P.xcd_chunk = (int)cdiv64(b16, 8); b16 = 8LL * P.xcd_chunk; }
        if (v16 == 1) launch16<2, 1, false>(P, b16, s);
        else if (v16 == 2) launch16<2, 1, true>(P, b16, s);
        else launch16<1, 2, false>(P, b16, s);
      } else if (impl != '0') {
        unet_set_max_lds(reinterpret_cast<const void*>(wgrad_dma_kernel<true>), WDma::LDS);
        unet_set_max_lds(reinterpret_cast<const void*>(wgrad_dma_kernel<false>), WDma::LDS);
        if (impl == '2')                                        // "2": every tap re-reads its X fragment
          hipLaunchKernelGGL(wgrad_dma_kernel<false>, dim3((unsigned)blocks), dim3(256), WDma::LDS, s, P);
        else
          hipLaunchKernelGGL(wgrad_dma_kernel<true>, dim3((unsigned)blocks), dim3(256), WDma::LDS, s, P);
      } else {
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), C::LDS, s, P);
      }
    } else {
      hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), C::LDS, s, P);
    }
    int32_t rc = unet_check_launch("wgrad_kernel");
    if (rc) return rc;
    if (split_used >= 16) {
      const long long rb = (long long)rows_out * ((cols_out + 63) / 64) * TAPS;
      hipLaunchKernelGGL(wgrad_reduce_wide_kernel<TAPS>, dim3((unsigned)rb), dim3(256), 0, s, (const float*)P.partial,
                         out, split_used, P.Crow, P.Ccol, rows_out, cols_out);
    } else {
      const long long total = (long long)rows_out * cols_out;
      const int rb = (int)std::min<long long>(cdiv64(total, 256), 4096);
      hipLaunchKernelGGL(wgrad_reduce_kernel<TAPS>, dim3(rb), dim3(256), 0, s, (const float*)P.partial, out, split_used,
                         P.Crow, P.Ccol, rows_out, cols_out);
    }
  }
  return unet_check_launch("wgrad_reduce_kernel");
}

inline int pad64(int c) { return (c + 63) / 64 * 64; }

}  // namespace

extern "C" size_t unet_conv3x3_wgrad_workspace(int32_t n, int32_t h, int32_t w, int32_t c_in, int32_t c_out) {
  size_t bytes = make_plan<9>(n, h, w, pad64(c_out), pad64(c_in)).bytes;
  if (wgrad16_variant(w, pad64(c_out), pad64(c_in)))
    bytes = std::max(bytes, make_plan16(n, h, w, pad64(c_out), pad64(c_in)).bytes);
  return bytes;
}

extern "C" int32_t unet_conv3x3_wgrad(int32_t dtype, int32_t n, int32_t h, int32_t w, const unet_view src[2],
                                      const void* dy, int32_t c_out, float* dw, int32_t c_in_param,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  UNET_REQUIRE(src && src[0].ptr && dy && dw && workspace, UNET_ERR_BAD_ARG, "unet_conv3x3_wgrad: null pointer");
  UNET_REQUIRE(n > 0 && h > 0 && w > 0, UNET_ERR_BAD_ARG, "unet_conv3x3_wgrad: bad dims");
  const int ctot = src[0].c + (src[1].ptr ? src[1].c : 0);
  UNET_REQUIRE(c_out % 64 == 0, UNET_ERR_UNSUPPORTED, "unet_conv3x3_wgrad: c_out %d not a multiple of 64", c_out);
  // the image layer (8/16 padded channels) reads a 64-channel column tile: the caller must have padded
  // the tensor to 64 channels or given c % 64 == 0 views.
  UNET_REQUIRE(ctot % 64 == 0 && (!src[1].ptr || src[0].c % 64 == 0), UNET_ERR_UNSUPPORTED,
               "unet_conv3x3_wgrad: input channels %d(+%d) must be multiples of 64", src[0].c,
               src[1].ptr ? src[1].c : 0);
  UNET_REQUIRE(c_in_param <= ctot, UNET_ERR_BAD_ARG, "unet_conv3x3_wgrad: c_in_param %d > %d", c_in_param, ctot);
  const Plan pl = make_plan<9>(n, h, w, c_out, ctot);
  const size_t need = unet_conv3x3_wgrad_workspace(n, h, w, ctot, c_out);
  UNET_REQUIRE(workspace_bytes >= need, UNET_ERR_WORKSPACE, "unet_conv3x3_wgrad: workspace %zu < %zu", workspace_bytes, need);
  WgradParams P{};
#ifdef PDMA_STAMPS
  P.debug = g_wgrad_debug;
#endif
  P.rt = WView{(const char*)dy, c_out, h, w, 0, 0};
  P.ct[0] = WView{(const char*)src[0].ptr, src[0].c, src[0].h, src[0].w, src[0].off_y, src[0].off_x};
  P.ct[1] = src[1].ptr ? WView{(const char*)src[1].ptr, src[1].c, src[1].h, src[1].w, src[1].off_y, src[1].off_x}
                       : WView{nullptr, 0, 0, 0, 0, 0};
  P.N = n; P.H = h; P.W = w;
  P.Crow = c_out; P.Ccol = ctot;
  P.partial = (float*)workspace;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == UNET_BF16) return run<bf16_t, 9>(P, pl, dw, c_out, c_in_param, UNET_K_CONV_WGRAD, s);
  if (dtype == UNET_F32) return run<float, 9>(P, pl, dw, c_out, c_in_param, UNET_K_CONV_WGRAD, s);
  unet_set_error("unet_conv3x3_wgrad: dtype %d", dtype);
  return UNET_ERR_BAD_ARG;
}

namespace {

// ------------------------------------------------------------------------------------------------------
// convt_wgrad_ws_kernel<CIN>: weight (and bias) gradient of the wide transposed convolutions (up3: 256->128 @64x64,
// up4: 128->64 @128x128), streaming.  dW[ci][z][co] = sum_p X[p][ci] * dYg[p][z*Cout + co], dYg[p] = the four
// 2x2 sub-positions of dY gathered into one 4*Cout row (the gather is the DMA's per-lane source address), so the
// whole output is a [Cin] x [4*Cout] matrix that a block keeps in REGISTERS (4 waves x (64 x 128) fp32) while
// 32-pixel tiles of X and dYg stream through LDS by LDS-DMA (double buffered).  K = pixels is the slow NHWC axis:
// both operands are read with ds_read_b64_tr_b16; rows are 256 / 512 B, 16-byte pieces XOR-swizzled by (row & 3) << 2
// (on the DMA source address and on the reads) so the four pixel rows of a transposed read sit in four different
// 64-byte bank quarters.  The bias gradient (column sums of dYg) rides along as VALU adds on the B fragments.
// HBM-bound: every byte of X and dY is read once (twice for Cin = 256, where the output is split 2 x 2 over blocks).
// Split-K partial slabs [split][Cin + 1][4*Cout] are summed in a fixed order -> bitwise reproducible.
struct CtwParams {
  const char* x; const char* dy; float* part;
  int N, H, W, tiles, tiles_per_split;
};
template <int CIN>
struct CfgCTW {
  static constexpr int COUT = CIN / 2, NC = 2 * CIN;
  static constexpr int TP = 32;
  static constexpr int X_BYTES = TP * 256, D_BYTES = TP * 512, BUF = X_BYTES + D_BYTES;   // 8 + 16 KiB
  static constexpr int NDMA = BUF / 1024 / 4;                                              // 6 per wave
  static constexpr int RB = CIN / 128, CB = NC / 256;
  static constexpr int LDS = 2 * BUF;
};

template <int CIN>
__global__ __launch_bounds__(256, 2) void convt_wgrad_ws_kernel(const CtwParams P) {
  using C = CfgCTW<CIN>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void lds_void;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave & 1, wc = wave >> 1;
  const int l31 = lane & 31, hh = lane >> 5;
  const int rb = blockIdx.y / C::CB, cb = blockIdx.y % C::CB;
  const int sp = blockIdx.x;
  const int t_begin = sp * P.tiles_per_split, t_end = min(t_begin + P.tiles_per_split, P.tiles);

  f32x16 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float cs[4] = {0.f, 0.f, 0.f, 0.f};            // column sums (bias gradient), wr == 0 waves only

  // transposed-read lane geometry (as in wgrad_dma_kernel)
  const int g = lane >> 4, i16 = lane & 15;
  const int kq = 8 * (g >> 1) + (i16 >> 2);
  const int chb = (16 * (g & 1) + 4 * (i16 & 3)) * 2;
  const int swz = (kq & 3) << 2;                 // rows kq, kq+4, kq+16, kq+20: same (row & 3)
  int aoff[2], boff[4];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int c = (wr * 64 + mt * 32) * 2 + chb;
    aoff[mt] = kq * 256 + (((c >> 4) ^ swz) << 4) + (c & 15);
  }
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int c = (wc * 128 + nt * 32) * 2 + chb;
    boff[nt] = C::X_BYTES + kq * 512 + (((c >> 4) ^ swz) << 4) + (c & 15);
  }

  // DMA geometry: instruction ii = j*4 + wave; ii < 8: X (4 pixel rows of 256 B), else dYg (2 pixel rows of 512 B)
  int d_row[C::NDMA];
  unsigned d_off[C::NDMA];                       // byte offset inside the pixel's source row (X) / code (dYg)
#pragma unroll
  for (int j = 0; j < C::NDMA; ++j) {
    const int ii = j * 4 + wave;
    if (ii < 8) {
      const int row = ii * 4 + (lane >> 4), piece = lane & 15;
      d_row[j] = row;
      d_off[j] = (unsigned)(rb * 256 + ((piece ^ ((row & 3) << 2)) << 4));
    } else {
      const int row = (ii - 8) * 2 + (lane >> 5), piece = lane & 31;
      const int n0 = cb * 256 + ((piece ^ ((row & 3) << 2)) << 3);       // first of this piece's 8 columns
      const int z = n0 / C::COUT, co = n0 - z * C::COUT;
      d_row[j] = row;
      d_off[j] = (unsigned)(z | (co << 8));
    }
  }
  const long long x_total = (long long)P.N * P.H * P.W * CIN * 2;
  const long long dy_total = 2 * x_total;        // 4*H*W*COUT*2 bytes per image = 2 * H*W*CIN*2
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)P.x, (short)0, (int)std::min<long long>(x_total, 0x7FFFFFFFLL), 0x00020000);
  const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)P.dy, (short)0, (int)std::min<long long>(dy_total, 0x7FFFFFFFLL), 0x00020000);
  // W % 32 == 0: a tile lies inside one image row; W == 16 (the deepest level): a tile is two whole rows of one image
  // (H is even there), a lane's pixel row r sits in tile row r >> 4
  const bool narrow = P.W < C::TP;
  const int tiles_row = narrow ? 1 : P.W / C::TP;

  auto dma = [&](int tile, int buf) {
    const int ry = narrow ? tile * 2 : tile / tiles_row;                     // ry = n*H + y (wave-uniform)
    const int x0 = narrow ? 0 : (tile - ry * tiles_row) * C::TP;
    const int n = ry / P.H, y = ry - n * P.H;
    const long long xpix0 = (long long)ry * P.W + x0;
    const long long dpix0 = ((long long)n * 2 * P.H + 2 * y) * (2 * P.W) + 2 * x0;
#pragma unroll
    for (int j = 0; j < C::NDMA; ++j) {
      const int ii = j * 4 + wave;
      char* dst = smem + buf * C::BUF + ii * 1024;
      if (ii < 8) {
        const unsigned vo = (unsigned)((xpix0 + d_row[j]) * (CIN * 2)) + d_off[j];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void*)dst, 16, vo, 0, 0, 0);
      } else {
        const int z = d_off[j] & 255, co = d_off[j] >> 8;
        // (narrow: pixel row r = 16 * dy + x -> two output rows further down per dy)
        const int rr = narrow ? (d_row[j] >> 4) * (4 * P.W) + 2 * (d_row[j] & 15) : 2 * d_row[j];
        const long long dp = dpix0 + (long long)(z >> 1) * (2 * P.W) + rr + (z & 1);
        const unsigned vo = (unsigned)(dp * (C::COUT * 2) + co * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(drs, (lds_void*)dst, 16, vo, 0, 0, 0);
      }
    }
  };

  if (t_begin < t_end) dma(t_begin, 0);
  for (int tile = t_begin; tile < t_end; ++tile) {
    const int buf = (tile - t_begin) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's share of the tile has landed
    __builtin_amdgcn_s_barrier();                               // ... everyone's has; the other buffer is free
    if (tile + 1 < t_end) dma(tile + 1, buf ^ 1);
    const char* sb = smem + buf * C::BUF;
#pragma unroll
    for (int ks = 0; ks < C::TP / 16; ++ks) {
      bf16x8 fa[2], fb[4];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) fa[mt] = tr_frag2(sb, aoff[mt] + ks * 16 * 256, aoff[mt] + ks * 16 * 256 + 4 * 256);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) fb[nt] = tr_frag2(sb, boff[nt] + ks * 16 * 512, boff[nt] + ks * 16 * 512 + 4 * 512);
      if (wr == 0 && rb == 0) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int e = 0; e < 8; ++e) cs[nt] += (float)fb[nt][e];
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt], fb[nt], acc[mt][nt], 0, 0, 0);
    }
  }

  float* slab = P.part + (size_t)sp * (CIN + 1) * C::NC;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int col = cb * 256 + wc * 128 + nt * 32 + l31;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rb * 128 + wr * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        slab[(size_t)row * C::NC + col] = acc[mt][nt][r];
      }
    if (wr == 0 && rb == 0) {
      const float t = cs[nt] + __shfl_xor(cs[nt], 32);          // the two pixel halves of the fragment
      if (hh == 0) slab[(size_t)CIN * C::NC + col] = t;
    }
  }
}

// dw[ci][co][z] = sum_s part[s][ci][z*Cout + co], fixed order.  grid (Cin, NC / 64), 256 threads = 64 columns x 4
// split lanes.  Row Cin of the slabs holds the column sums of dYg: grid row Cin folds its four z into db[co].
__global__ __launch_bounds__(256) void convt_wgrad_ws_reduce_kernel(const float* __restrict__ part, int nsplit, int CIN,
                                                                    float* __restrict__ dw, float* __restrict__ db) {
  __shared__ float red[4][64];
  const int NC = 2 * CIN, COUT = CIN / 2;
  const int row = blockIdx.x, c = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const size_t stride = (size_t)(CIN + 1) * NC;
  float s0 = 0.f, s1 = 0.f;
  if (row < CIN) {
    const int col = blockIdx.y * 64 + c;
    const float* p = part + (size_t)row * NC + col;
    int s = sl;
    for (; s + 4 < nsplit; s += 8) { s0 += p[(size_t)s * stride]; s1 += p[(size_t)(s + 4) * stride]; }
    if (s < nsplit) s0 += p[(size_t)s * stride];
  } else {
    // bias: blocks y < Cout / 64 only; co = y*64 + c, the four z columns are summed per split in z order
    const int co = blockIdx.y * 64 + c;
    if (co < COUT) {
      const float* p = part + (size_t)CIN * NC + co;
      for (int s = sl; s < nsplit; s += 4) {
        const float* q = p + (size_t)s * stride;
        s0 += ((q[0] + q[COUT]) + q[2 * COUT]) + q[3 * COUT];
      }
    }
  }
  red[sl][c] = s0 + s1;
  __syncthreads();
  if (sl == 0) {
    const float t = ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
    if (row < CIN) {
      const int col = blockIdx.y * 64 + c;
      const int z = col / COUT, co = col - z * COUT;
      dw[((size_t)row * COUT + co) * 4 + z] = t;
    } else if (blockIdx.y * 64 + c < COUT) {
      db[blockIdx.y * 64 + c] = t;
    }
  }
}

template <int CIN>
int32_t launch_convt_wgrad_ws(const void* x, const void* dy, int n, int h, int w, float* dw, float* db, void* workspace,
                              size_t workspace_bytes, hipStream_t s) {
  using C = CfgCTW<CIN>;
  const long long px = (long long)n * h * w;
  CtwParams P{(const char*)x, (const char*)dy, (float*)workspace, n, h, w, (int)(px / C::TP), 0};
  const int yb = C::RB * C::CB;
  int nsplit = 2 * unet_cu_budget() / yb;
  if (nsplit < 1) nsplit = 1;
  if (nsplit > P.tiles) nsplit = P.tiles;
  P.tiles_per_split = (P.tiles + nsplit - 1) / nsplit;
  nsplit = (P.tiles + P.tiles_per_split - 1) / P.tiles_per_split;
  const size_t need = (size_t)nsplit * (CIN + 1) * C::NC * sizeof(float);
  UNET_REQUIRE(workspace_bytes >= need, UNET_ERR_WORKSPACE, "unet_convt2x2_wgrad: workspace %zu < %zu", workspace_bytes, need);
  auto kern = convt_wgrad_ws_kernel<CIN>;
  unet_set_max_lds(reinterpret_cast<const void*>(kern), C::LDS);
  {
    ProfScope prof(UNET_K_CONVT_WGRAD, 2.0 * px * 4.0 * C::COUT * CIN, s, "convt_wgrad_ws_kernel (+ reduce)");
    hipLaunchKernelGGL(kern, dim3(nsplit, yb), dim3(256), C::LDS, s, P);
    int32_t rc = unet_check_launch("convt_wgrad_ws_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(convt_wgrad_ws_reduce_kernel, dim3(CIN + 1, C::NC / 64), dim3(256), 0, s, (const float*)workspace,
                       nsplit, CIN, dw, db);
  }
  return unet_check_launch("convt_wgrad_ws_reduce_kernel");
}

inline bool convt_wgrad_ws_ok(int dtype, int n, int h, int w, int c_in, int c_out) {
  const long long xb = (long long)n * h * w * c_in * 2;
  const bool deep = (c_in == 512 || c_in == 1024) && unet_tuning().convt_impl != '3';   // (3: deep levels generic)
  return dtype == UNET_BF16 && (c_in == 128 || c_in == 256 || deep) && c_out * 2 == c_in &&
         (w % 32 == 0 || (w == 16 && h % 2 == 0)) && ((long long)n * h * w) % 32 == 0 &&
         2 * xb < 0x7FFFFFFFLL && unet_tuning().convt_impl != '0';      // (UNET_CONVT_IMPL=0: generic kernels)
}
inline size_t convt_wgrad_ws_bytes(int n, int h, int w, int c_in) {
  const long long tiles = (long long)n * h * w / 32;
  const int yb = (c_in / 128) * (c_in / 128);
  long long nsplit = 2 * unet_cu_budget() / yb;
  if (nsplit < 1) nsplit = 1;
  if (nsplit > tiles) nsplit = tiles;
  return (size_t)nsplit * (c_in + 1) * (2 * c_in) * sizeof(float);
}

}  // namespace

extern "C" size_t unet_convt2x2_wgrad_workspace(int32_t n, int32_t h, int32_t w, int32_t c_in, int32_t c_out) {
  size_t bytes = make_plan<4>(n, h, w, pad64(c_in), pad64(c_out)).bytes;
  if (convt_wgrad_ws_ok(UNET_BF16, n, h, w, c_in, c_out)) bytes = std::max(bytes, convt_wgrad_ws_bytes(n, h, w, c_in));
  return bytes;
}

extern "C" int32_t unet_convt2x2_wgrad(int32_t dtype, int32_t n, int32_t h, int32_t w, const void* x,
                                       int32_t c_in, const void* dy, int32_t c_out, float* dw, float* db,
                                       void* workspace, size_t workspace_bytes, void* stream) {
  UNET_REQUIRE(x && dy && dw && db && workspace, UNET_ERR_BAD_ARG, "unet_convt2x2_wgrad: null pointer");
  UNET_REQUIRE(n > 0 && h > 0 && w > 0, UNET_ERR_BAD_ARG, "unet_convt2x2_wgrad: bad dims");
  UNET_REQUIRE(c_in % 64 == 0 && c_out % 64 == 0, UNET_ERR_UNSUPPORTED,
               "unet_convt2x2_wgrad: channels %d -> %d must be multiples of 64", c_in, c_out);
  if (convt_wgrad_ws_ok(dtype, n, h, w, c_in, c_out)) {
    hipStream_t st = (hipStream_t)stream;
    if (c_in == 128) return launch_convt_wgrad_ws<128>(x, dy, n, h, w, dw, db, workspace, workspace_bytes, st);
    if (c_in == 256) return launch_convt_wgrad_ws<256>(x, dy, n, h, w, dw, db, workspace, workspace_bytes, st);
    if (c_in == 512) return launch_convt_wgrad_ws<512>(x, dy, n, h, w, dw, db, workspace, workspace_bytes, st);
    return launch_convt_wgrad_ws<1024>(x, dy, n, h, w, dw, db, workspace, workspace_bytes, st);
  }
  const Plan pl = make_plan<4>(n, h, w, c_in, c_out);
  UNET_REQUIRE(workspace_bytes >= pl.bytes, UNET_ERR_WORKSPACE, "unet_convt2x2_wgrad: workspace %zu < %zu",
               workspace_bytes, pl.bytes);
  WgradParams P{};
  P.rt = WView{(const char*)x, c_in, h, w, 0, 0};
  P.ct[0] = WView{(const char*)dy, c_out, 2 * h, 2 * w, 0, 0};
  P.ct[1] = WView{nullptr, 0, 0, 0, 0, 0};
  P.N = n; P.H = h; P.W = w;
  P.Crow = c_in; P.Ccol = c_out;
  P.partial = (float*)workspace;
  hipStream_t s = (hipStream_t)stream;
  int32_t rc;
  if (dtype == UNET_BF16) rc = run<bf16_t, 4>(P, pl, dw, c_in, c_out, UNET_K_CONVT_WGRAD, s);
  else if (dtype == UNET_F32) rc = run<float, 4>(P, pl, dw, c_in, c_out, UNET_K_CONVT_WGRAD, s);
  else { unet_set_error("unet_convt2x2_wgrad: dtype %d", dtype); return UNET_ERR_BAD_ARG; }
  if (rc) return rc;
  // bias gradient: column sums of dy; the partial slabs above are consumed (stream order), reuse them
  return unet_internal_colsum(dtype, dy, (int64_t)n * 4 * h * w, c_out, db, (float*)workspace, workspace_bytes, s);
}

#ifdef PDMA_STAMPS
extern "C" void unet_debug_set_buffer_wgrad(void* p) { g_wgrad_debug = p; }
#endif
