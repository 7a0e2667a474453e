// The first convolution of the network, inc.double_conv.0 (nn.Conv2d(n_channels, 64, 3, padding=1) at
// /root/reference/src/model.py:14 reached from UNet.forward :98 / AnomalyUNet.forward :190), bf16 mode.
//
// With n_channels <= 3 the whole reduction is K = 9*Cin <= 27 -> ONE 32-deep MFMA step.  Padding the image
// to 64 channels (the generic path) costs a 268 MB padded copy per step at bs=32 256x256 plus two 64-channel
// convolutions' worth of MFMA time; here the kernels read the caller's fp32 NCHW image directly:
//
//   first_fwd_kernel   D[co][px] = W[co][k] * Xcol[px][k], k = tap*Cin + ci.  A wave owns 16-pixel row segments;
//                      each lane gathers its 8 K-values (buffer loads, out-of-image = out-of-range offset = 0),
//                      4 x v_mfma_f32_16x16x32_bf16 give 64 channels x 16 pixels, v_permlane16_swap regroups
//                      the accumulators into 16-byte NHWC stores.  BatchNorm batch statistics of the stored
//                      (bf16-rounded) values accumulate in registers and leave as ONE ordered partial per block.
//                      Bound by the 64-channel output write (HBM).
//   first_wgrad_kernel dW[co][k] = sum_px dY[px][co] * Xcol[px][k]: K = pixels.  Each wave streams 16-pixel
//                      dY row segments (16 x 128 B) through a wave-private LDS double buffer by LDS-DMA and reads
//                      them transposed (ds_read_b64_tr_b16); the Xcol operand is 8 consecutive pixels per lane
//                      straight from the fp32 image.  2 x v_mfma_f32_32x32x16_bf16 per segment; block partials
//                      [64][32] are summed in a fixed order by first_wgrad_reduce_kernel -> bitwise reproducible.
//                      Bound by the dY read (HBM).
#include <stdlib.h>

#include <algorithm>

#include "common.h"

namespace {

constexpr unsigned OOB = 0xFFFFFFF0u;
constexpr int FIRST_CO = 64;

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

// Position of a 16-pixel row segment; the kernels walk segments with a constant stride, so positions advance by
// carries instead of 64-bit divisions (which cost more than the segment's arithmetic).
struct SegPos { int seg, y, n; };
__device__ __forceinline__ SegPos seg_step(SegPos p, int dseg, int dy, int dn, int segs_row, int H) {
  p.seg += dseg;
  int c = p.seg >= segs_row;
  p.seg -= c ? segs_row : 0;
  p.y += dy + c;
  c = p.y >= H;
  p.y -= c ? H : 0;
  p.n += dn + c;
  return p;
}
struct SegWalk { SegPos first; int dseg, dy, dn; };
__device__ __forceinline__ SegWalk seg_walk(long long u0, long long stride, int segs_row, int H) {
  SegWalk w;
  const long long r0 = u0 / segs_row, rs = stride / segs_row;
  w.first = SegPos{(int)(u0 - r0 * segs_row), (int)(r0 % H), (int)std::min<long long>(r0 / H, 1 << 30)};
  w.dseg = (int)(stride - rs * segs_row);
  w.dy = (int)(rs % H);
  w.dn = (int)std::min<long long>(rs / H, 1 << 28);
  return w;
}

struct FirstParams {
  const float* x;        // [N][CI][H][W] fp32
  const float* w;        // [64][CI][3][3] fp32
  char* y;               // [N][H][W][64] bf16 (fwd: output; wgrad: dY)
  float* part;           // fwd: BN partials [blocks][2][64] (may be NULL); wgrad: [blocks][64][32]
  int N, CI, H, W;
  const char* y_raw;     // wgrad<BNB>: the layer's raw conv output (P.y is then the ReLU-masked gradient dz)
  const float* coefs;    // wgrad<BNB>: [3][64] A, B, K of dy = A*dz + B*y + K
};

// ------------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(256, 3) void first_fwd_kernel(const FirstParams P) {
  __shared__ float red[4][2][FIRST_CO];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, kb = lane >> 4;
  const int K = 9 * P.CI;
  const int HW = P.H * P.W;

  // A fragments: W[co = ct*16 + l15][k = kb*8 + e], k = tap*CI + ci
  bf16x8 wa[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = kb * 8 + e;
      float v = 0.f;
      if (k < K) {
        const int tap = k / P.CI, ci = k - tap * P.CI;
        v = P.w[((ct * 16 + l15) * P.CI + ci) * 9 + tap];
      }
      wa[ct][e] = (bf16_t)v;
    }
  // gather geometry of this lane's 8 K-values
  int rel[8], reli[8], dyv[8], dxv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = kb * 8 + e;
    const int tap = k < K ? k / P.CI : 0, ci = k < K ? k - tap * P.CI : 0;
    dyv[e] = k < K ? tap / 3 - 1 : 1 << 20;          // invalid k: never inside the image
    dxv[e] = tap % 3 - 1;
    rel[e] = (ci * HW + dyv[e] * P.W + dxv[e]) * 4;
    reli[e] = k < K ? rel[e] : 0;                    // interior path: padding K reads the centre pixel (weight is 0)
  }

  const int segs_row = P.W >> 4;
  const long long stride = (long long)gridDim.x * 4;
  const unsigned img_bytes = (unsigned)P.CI * HW * 4u;

  auto gather = [&](SegPos q, float (&xv)[8]) {
    const int x = q.seg * 16 + l15;
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc((void*)(P.x + (size_t)q.n * P.CI * HW), (short)0, (int)img_bytes, 0x00020000);
    const int base = (q.y * P.W + x) * 4;
    if (q.y > 0 && q.y + 1 < P.H && q.seg > 0 && q.seg + 1 < segs_row) {     // wave-uniform: no border in reach
#pragma unroll
      for (int e = 0; e < 8; ++e)
        xv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (unsigned)(base + reli[e]), 0, 0));
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const bool ok = (unsigned)(q.y + dyv[e]) < (unsigned)P.H && (unsigned)(x + dxv[e]) < (unsigned)P.W;
        xv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, ok ? (unsigned)(base + rel[e]) : OOB, 0, 0));
      }
    }
  };

  float bs[4][4], bq[4][4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int j = 0; j < 4; ++j) { bs[ct][j] = 0.f; bq[ct][j] = 0.f; }

  // A wave takes FU segments per iteration and keeps the next iteration's 8*FU gathers in flight behind the
  // current one's MFMAs and stores: the kernel is latency-bound otherwise (one segment = 4 MFMAs).
  constexpr int FU = 2;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const SegWalk wk = seg_walk(((long long)blockIdx.x * 4 + wave_u) * FU, stride * FU, segs_row, P.H);
  SegPos cur = wk.first;
  float xc[FU][8], xn[FU][8];
#pragma unroll
  for (int i = 0; i < FU; ++i) {
    const SegPos q = seg_step(cur, i, 0, 0, segs_row, P.H);
    if (q.n < P.N) gather(q, xc[i]);
  }
  while (cur.n < P.N) {
    const SegPos nxt = seg_step(cur, wk.dseg, wk.dy, wk.dn, segs_row, P.H);
#pragma unroll
    for (int i = 0; i < FU; ++i) {
      const SegPos q = seg_step(nxt, i, 0, 0, segs_row, P.H);
      if (q.n < P.N) gather(q, xn[i]);
    }
#pragma unroll
    for (int i = 0; i < FU; ++i) {
      const SegPos q = seg_step(cur, i, 0, 0, segs_row, P.H);
      if (q.n >= P.N) break;
      bf16x8 fb;
#pragma unroll
      for (int e = 0; e < 8; ++e) fb[e] = (bf16_t)xc[i][e];
      f32x4 acc[4];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ct], fb, acc[ct], 0, 0, 0);
      }
      const long long pix = ((long long)q.n * P.H + q.y) * P.W + q.seg * 16 + l15;
      bf16_t* orow = reinterpret_cast<bf16_t*>(P.y) + pix * FIRST_CO;
#pragma unroll
      for (int cp = 0; cp < 2; ++cp) {
        bf16x4 ra, rb;
#pragma unroll
        for (int j = 0; j < 4; ++j) { ra[j] = (bf16_t)acc[2 * cp][j]; rb[j] = (bf16_t)acc[2 * cp + 1][j]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {                 // statistics of the values as STORED
          const float qa = (float)ra[j], qb = (float)rb[j];
          bs[2 * cp][j] += qa;      bq[2 * cp][j] = fmaf(qa, qa, bq[2 * cp][j]);
          bs[2 * cp + 1][j] += qb;  bq[2 * cp + 1][j] = fmaf(qb, qb, bq[2 * cp + 1][j]);
        }
        // lane kb ends up with tile 2cp + (kb & 1), channels 8*(kb >> 1) .. +7 (see conv3_pdma_kernel)
        const u32x2 ua = __builtin_bit_cast(u32x2, ra), ub = __builtin_bit_cast(u32x2, rb);
        const auto s0 = __builtin_amdgcn_permlane16_swap(ua[0], ub[0], false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(ua[1], ub[1], false, false);
        *reinterpret_cast<u32x4*>(orow + cp * 32 + (kb & 1) * 16 + (kb >> 1) * 8) = u32x4{s0[0], s1[0], s0[1], s1[1]};
      }
    }
#pragma unroll
    for (int i = 0; i < FU; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) xc[i][e] = xn[i][e];
    cur = nxt;
  }

  if (P.part) {
    {
      float rv[32];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int j = 0; j < 4; ++j) { rv[ct * 4 + j] = bs[ct][j]; rv[16 + ct * 4 + j] = bq[ct][j]; }
      row16_sum_n(rv);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int j = 0; j < 4; ++j) { bs[ct][j] = rv[ct * 4 + j]; bq[ct][j] = rv[16 + ct * 4 + j]; }
    }
    if (l15 == 0) {
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          red[wave][0][ct * 16 + kb * 4 + j] = bs[ct][j];
          red[wave][1][ct * 16 + kb * 4 + j] = bq[ct][j];
        }
    }
    __syncthreads();
    if (tid < 2 * FIRST_CO) {
      const int q = tid / FIRST_CO, c = tid - q * FIRST_CO;
      const float t = ((red[0][q][c] + red[1][q][c]) + red[2][q][c]) + red[3][q][c];   // fixed order
      P.part[((size_t)blockIdx.x * 2 + q) * FIRST_CO + c] = t;
    }
  }
}

// ------------------------------------------------------------------------------------------------ wgrad
__device__ inline bf16x8 tr_frag2(const char* base, int off0, int off1) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + off0));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + off1));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// BNB (round 4): the BatchNorm-backward apply of this layer rides along.  The image layer is the one layer whose dy has a
// SINGLE consumer -- this kernel; there is no data gradient towards the image -- so the standalone pass
// dy = A*dz + B*y + K (read dz, read y, write dy: 805 MB at bs = 32, 256 x 256) disappears: the kernel streams the masked
// gradient dz AND the raw conv output y (segment for segment, side by side in the wave's LDS buffer), and a lane -- which
// holds 8 pixels of ONE channel after the transposed read -- forms dy with that channel's three coefficients and rounds
// it to bf16 exactly as the apply pass stored it: the same MFMA operands, bit-identical dW.
template <bool BNB>
__global__ __launch_bounds__(256, BNB ? 2 : 4) void first_wgrad_kernel(const FirstParams P) {
  // per wave: 2 x WU 2-KiB dY segments (16 pixels x 64 channels bf16) [BNB: + the y segment behind each]; reused for
  // the block reduction
  constexpr int SEGB = BNB ? 4096 : 2048;
  __shared__ __attribute__((aligned(16))) char smem[BNB ? 4 * 2 * 2 * 4096 : 4 * FIRST_CO * 32 * 4];   // >= 32 KiB
  typedef __attribute__((address_space(3))) void lds_void;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int K = 9 * P.CI;
  const int HW = P.H * P.W;
  constexpr int WU = 2;                           // segments per iteration (two more in flight behind them)
  char* const wbuf = smem + wave * (WU * 2 * SEGB);

  // B operand geometry: lane (n = l31 = k index, hh): pixels 8hh .. 8hh+7 of the segment
  const bool kvalid = l31 < K;
  const int tap = kvalid ? l31 / P.CI : 0, ci = kvalid ? l31 - tap * P.CI : 0;
  const int dy = tap / 3 - 1, dx = tap % 3 - 1;
  const int relb = (ci * HW + dy * P.W + dx + 8 * hh) * 4;

  // transposed-read geometry of the A operand (dY^T), as in wgrad_dma_kernel: rows = pixels, 128-byte rows,
  // 16-byte pieces XOR-swizzled by ((row >> 1) & 1) << 2 through the DMA's source address
  const int g = lane >> 4, i16 = lane & 15;
  const int kq = 8 * (g >> 1) + (i16 >> 2);
  const int chb = (16 * (g & 1) + 4 * (i16 & 3)) * 2;
  const int swz = ((kq >> 1) & 1) << 2;
  int aoff[2];
#pragma unroll
  for (int wr = 0; wr < 2; ++wr) {
    const int cb = wr * 64 + chb;
    aoff[wr] = kq * 128 + ((((cb >> 4) ^ swz)) << 4) + (cb & 15);
  }
  // DMA source of this lane's two 16-byte pieces per segment: instruction i covers pixel rows 8i .. 8i+7
  unsigned dsrc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = i * 8 + (lane >> 3), pp = lane & 7;
    dsrc[i] = (unsigned)(row * 128 + ((pp ^ (((row >> 1) & 1) << 2)) << 4));
  }

  const int segs_row = P.W >> 4;
  const long long stride = (long long)gridDim.x * 4;
  const unsigned img_bytes = (unsigned)P.CI * HW * 4u;
  const long long dy_total = (long long)P.N * HW * FIRST_CO * 2;
  // dY is addressed per segment through a 64-bit base + small offsets; one resource per segment
  auto issue = [&](SegPos q, int slot, float (&xv)[8]) {
    const int seg = q.seg, y = q.y, n = q.n;
    const long long pix0 = ((long long)n * P.H + y) * P.W + seg * 16;
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(P.y + pix0 * (FIRST_CO * 2)), (short)0, (int)std::min<long long>(2048, dy_total - pix0 * (FIRST_CO * 2)),
        0x00020000);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(drs, (lds_void*)(wbuf + slot * SEGB + i * 1024), 16, dsrc[i], 0, 0, 0);
    if constexpr (BNB) {
      const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(P.y_raw + pix0 * (FIRST_CO * 2)), (short)0, (int)std::min<long long>(2048, dy_total - pix0 * (FIRST_CO * 2)),
          0x00020000);
#pragma unroll
      for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(yrs, (lds_void*)(wbuf + slot * SEGB + 2048 + i * 1024), 16, dsrc[i], 0, 0, 0);
    }
    const __amdgpu_buffer_rsrc_t xrs =
        __builtin_amdgcn_make_buffer_rsrc((void*)(P.x + (size_t)n * P.CI * HW), (short)0, (int)img_bytes, 0x00020000);
    const int x0 = seg * 16;
    const int base = (y * P.W + x0) * 4 + relb;
    if (y > 0 && y + 1 < P.H && seg > 0 && seg + 1 < segs_row) {     // wave-uniform: no border in reach; the padding
      const unsigned b0 = kvalid ? (unsigned)base : 0u;               // columns k >= 9*CI read anything finite (never used)
#pragma unroll
      for (int e = 0; e < 8; ++e)
        xv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, b0 + e * 4, 0, 0));
    } else {
      const bool yok = kvalid && (unsigned)(y + dy) < (unsigned)P.H;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const bool ok = yok && (unsigned)(x0 + 8 * hh + e + dx) < (unsigned)P.W;
        xv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, ok ? (unsigned)(base + e * 4) : OOB, 0, 0));
      }
    }
  };

  f32x16 acc[2];
#pragma unroll
  for (int wr = 0; wr < 2; ++wr)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[wr][r] = 0.f;

  // every iteration issues exactly WU x (2 [BNB: 4] DMAs + 8 loads) -- segments past the end become out-of-range no-ops --
  // so the wait that retires the CURRENT iteration's operands is a constant vmcnt(NV*WU)
  constexpr int NV = BNB ? 12 : 10;
  float cA[2], cB[2], cK[2];                      // BNB: coefficients of this lane's channel in each 32-channel half
  if constexpr (BNB) {
#pragma unroll
    for (int wr = 0; wr < 2; ++wr) {
      const int ch = wr * 32 + 16 * (g & 1) + i16;
      cA[wr] = P.coefs[ch]; cB[wr] = P.coefs[FIRST_CO + ch]; cK[wr] = P.coefs[2 * FIRST_CO + ch];
    }
  }
  auto issue_or_skip = [&](SegPos q, int slot, float (&xv)[8]) {
    if (q.n < P.N) {
      issue(q, slot, xv);
    } else {
      const __amdgpu_buffer_rsrc_t nul = __builtin_amdgcn_make_buffer_rsrc((void*)P.x, (short)0, 0, 0x00020000);
#pragma unroll
      for (int i = 0; i < (BNB ? 4 : 2); ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(nul, (lds_void*)(wbuf + slot * SEGB + i * 1024), 16, OOB, 0, 0, 0);
#pragma unroll
      for (int e = 0; e < 8; ++e) xv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(nul, OOB, 0, 0));
    }
  };
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const SegWalk wk = seg_walk(((long long)blockIdx.x * 4 + wave_u) * WU, stride * WU, segs_row, P.H);
  SegPos cur = wk.first;
  float xc[WU][8], xn[WU][8];
  int buf = 0;
#pragma unroll
  for (int i = 0; i < WU; ++i) issue_or_skip(seg_step(cur, i, 0, 0, segs_row, P.H), i, xc[i]);
  while (cur.n < P.N) {
    const SegPos nxt = seg_step(cur, wk.dseg, wk.dy, wk.dn, segs_row, P.H);
#pragma unroll
    for (int i = 0; i < WU; ++i) issue_or_skip(seg_step(nxt, i, 0, 0, segs_row, P.H), (buf ^ 1) * WU + i, xn[i]);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NV * WU) : "memory");
#pragma unroll
    for (int i = 0; i < WU; ++i) {
      bf16x8 fb;
#pragma unroll
      for (int e = 0; e < 8; ++e) fb[e] = (bf16_t)xc[i][e];          // zeros for a skipped segment
      const char* sb = wbuf + (buf * WU + i) * SEGB;
#pragma unroll
      for (int wr = 0; wr < 2; ++wr) {
        bf16x8 fa = tr_frag2(sb, aoff[wr], aoff[wr] + 4 * 128);
        if constexpr (BNB) {
          const bf16x8 fy = tr_frag2(sb + 2048, aoff[wr], aoff[wr] + 4 * 128);
#pragma unroll
          for (int e = 0; e < 8; ++e)            // dy as bn_bwd_apply_premasked_kernel stores it
            fa[e] = (bf16_t)fmaf(cA[wr], (float)fa[e], fmaf(cB[wr], (float)fy[e], cK[wr]));
        }
        acc[wr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[wr], 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < WU; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) xc[i][e] = xn[i][e];
    buf ^= 1;
    cur = nxt;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // block reduction in a fixed wave order: red[wave][co][k]
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int wr = 0; wr < 2; ++wr)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
      red[(wave * FIRST_CO + co) * 32 + l31] = acc[wr][r];
    }
  __syncthreads();
  for (int i = tid; i < FIRST_CO * 32; i += 256) {
    const float t = ((red[i] + red[FIRST_CO * 32 + i]) + red[2 * FIRST_CO * 32 + i]) + red[3 * FIRST_CO * 32 + i];
    P.part[(size_t)blockIdx.x * (FIRST_CO * 32) + i] = t;
  }
}

// dw[co][ci][tap] = sum_b part[b][co][tap*CI + ci].  One block per output channel: 32 k x 8 partial-lanes, each lane
// sums its partials (b = lane, lane + 8, ...) in ascending order, the 8 lane sums are combined in a fixed order.
__global__ __launch_bounds__(256) void first_wgrad_reduce_kernel(const float* __restrict__ part, int nparts, int CI,
                                                                 float* __restrict__ dw) {
  __shared__ float red[8][32];
  const int co = blockIdx.x, k = threadIdx.x & 31, pl = threadIdx.x >> 5;
  float s0 = 0.f, s1 = 0.f;
  int b = pl;
  for (; b + 8 < nparts; b += 16) {
    s0 += part[(size_t)b * (FIRST_CO * 32) + co * 32 + k];
    s1 += part[(size_t)(b + 8) * (FIRST_CO * 32) + co * 32 + k];
  }
  if (b < nparts) s0 += part[(size_t)b * (FIRST_CO * 32) + co * 32 + k];
  red[pl][k] = s0 + s1;
  __syncthreads();
  if (threadIdx.x < 9 * CI) {
    const int kk = threadIdx.x, tap = kk / CI, ci = kk - tap * CI;
    float t = red[0][kk];
#pragma unroll
    for (int i = 1; i < 8; ++i) t += red[i][kk];
    dw[(co * CI + ci) * 9 + tap] = t;
  }
}

// resident blocks: 3 per CU for the forward kernel (register budget), 4 for the weight gradient; at most 1024 (the
// capacity of the BatchNorm partial buffer, unet_conv3x3_stats_max_parts)
int first_blocks(long long units, int per_iter, int per_cu) {
  long long b = (units + 4 * per_iter - 1) / (4 * per_iter);
  return (int)std::min<long long>(b, 256 * per_cu);
}

}  // namespace

extern "C" int32_t unet_conv3x3_first_supported(int32_t c_in, int32_t c_out, int32_t h, int32_t w) {
  return (c_in >= 1 && 9 * c_in <= 32 && c_out == FIRST_CO && h > 0 && w > 0 && w % 16 == 0) ? 1 : 0;
}

extern "C" int32_t unet_conv3x3_first_stats(int32_t n, int32_t h, int32_t w, const float* x, int32_t c_in,
                                            const float* weight, void* y, float* partial, int32_t* n_parts,
                                            void* stream) {
  UNET_REQUIRE(x && weight && y, UNET_ERR_BAD_ARG, "unet_conv3x3_first_stats: null pointer");
  UNET_REQUIRE(n > 0 && unet_conv3x3_first_supported(c_in, FIRST_CO, h, w), UNET_ERR_UNSUPPORTED,
               "unet_conv3x3_first_stats: n=%d c_in=%d h=%d w=%d (needs 9*c_in <= 32, w %% 16 == 0)", n, c_in, h, w);
  UNET_REQUIRE((long long)c_in * h * w * 4 < 0x7FFFFFFFLL, UNET_ERR_UNSUPPORTED, "unet_conv3x3_first_stats: image too large");
  const long long units = (long long)n * h * (w / 16);
  const int blocks = first_blocks(units, 2, 3);
  FirstParams P{x, weight, (char*)y, partial, n, c_in, h, w, nullptr, nullptr};
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(UNET_K_CONV_FWD, 2.0 * n * h * w * (double)FIRST_CO * c_in * 9, s);
  hipLaunchKernelGGL(first_fwd_kernel, dim3(blocks), dim3(256), 0, s, P);
  if (n_parts) *n_parts = partial ? blocks : 0;
  return unet_check_launch("first_fwd_kernel");
}

extern "C" size_t unet_conv3x3_first_wgrad_workspace(int32_t n, int32_t h, int32_t w) {
  const long long units = (long long)n * h * (w / 16 > 0 ? w / 16 : 1);
  return (size_t)first_blocks(units, 2, 4) * FIRST_CO * 32 * sizeof(float);
}

static int32_t first_wgrad_impl(int32_t n, int32_t h, int32_t w, const float* x, int32_t c_in, const void* dy, const void* y_raw,
                                const float* coefs, float* dw, void* workspace, size_t workspace_bytes, void* stream,
                                const char* what) {
  UNET_REQUIRE(x && dy && dw && workspace, UNET_ERR_BAD_ARG, "%s: null pointer", what);
  UNET_REQUIRE(n > 0 && unet_conv3x3_first_supported(c_in, FIRST_CO, h, w), UNET_ERR_UNSUPPORTED,
               "%s: n=%d c_in=%d h=%d w=%d", what, n, c_in, h, w);
  UNET_REQUIRE((long long)c_in * h * w * 4 < 0x7FFFFFFFLL, UNET_ERR_UNSUPPORTED, "%s: image too large", what);
  UNET_REQUIRE(workspace_bytes >= unet_conv3x3_first_wgrad_workspace(n, h, w), UNET_ERR_WORKSPACE, "%s: workspace too small", what);
  const long long units = (long long)n * h * (w / 16);
  const int blocks = first_blocks(units, 2, 4);
  FirstParams P{x, nullptr, (char*)const_cast<void*>(dy), (float*)workspace, n, c_in, h, w, (const char*)y_raw, coefs};
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(UNET_K_CONV_WGRAD, 2.0 * n * h * w * (double)FIRST_CO * c_in * 9, s);
  if (y_raw) hipLaunchKernelGGL(first_wgrad_kernel<true>, dim3(blocks), dim3(256), 0, s, P);
  else hipLaunchKernelGGL(first_wgrad_kernel<false>, dim3(blocks), dim3(256), 0, s, P);
  int32_t rc = unet_check_launch("first_wgrad_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(first_wgrad_reduce_kernel, dim3(FIRST_CO), dim3(256), 0, s, (const float*)workspace, blocks, c_in, dw);
  return unet_check_launch("first_wgrad_reduce_kernel");
}

extern "C" int32_t unet_conv3x3_first_wgrad(int32_t n, int32_t h, int32_t w, const float* x, int32_t c_in,
                                            const void* dy, float* dw, void* workspace, size_t workspace_bytes,
                                            void* stream) {
  return first_wgrad_impl(n, h, w, x, c_in, dy, nullptr, nullptr, dw, workspace, workspace_bytes, stream,
                          "unet_conv3x3_first_wgrad");
}

extern "C" int32_t unet_conv3x3_first_wgrad_bn(int32_t n, int32_t h, int32_t w, const float* x, int32_t c_in,
                                               const void* dz, const void* y, const float* coefs, float* dw,
                                               void* workspace, size_t workspace_bytes, void* stream) {
  UNET_REQUIRE(y && coefs, UNET_ERR_BAD_ARG, "unet_conv3x3_first_wgrad_bn: null pointer");
  return first_wgrad_impl(n, h, w, x, c_in, dz, y, coefs, dw, workspace, workspace_bytes, stream,
                          "unet_conv3x3_first_wgrad_bn");
}
