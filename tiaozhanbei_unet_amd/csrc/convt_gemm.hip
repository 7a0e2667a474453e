// convt_gemm_kernel: the deep transposed convolutions (up1: 1024 -> 512 @16x16, up2: 512 -> 256 @32x32; reference
// src/model.py:45-53, nn.ConvTranspose2d(in, in // 2, kernel_size=2, stride=2)) as ONE plain GEMM each way:
//   forward        y[(n, 2h+zy, 2w+zx)][co] = b[co] + sum_ci x[(n,h,w)][ci] * w[ci][co][zy][zx]
//                  -> D[4*Cout rows (z, co)][pixels] = Wp[(z,co)][ci] . X[pixel][ci]^T, K = Cin
//   data gradient  dx[(n,h,w)][ci] = sum_{z,co} dy[(n, 2h+zy, 2w+zx)][co] * w[ci][co][z]
//                  -> D[Cin rows][pixels] = Wd[ci][(z,co)] . dY[pixel][(z,co)]^T, K = 4*Cout (four gathered segments)
// on a 256 (rows: output features) x 256 (pixels) x 64 tile per 512-thread block, both operands staged by LDS-DMA.
//
// Schedule ("ping-pong"): waves 0-3 own rows 0-127, waves 4-7 rows 128-255 (wave = 128 rows x 64 pixels, 8 x 4
// accumulator tiles of v_mfma_f32_16x16x32_bf16 = 128 VGPRs); the two waves of a SIMD (w, w + 4) run one barrier apart:
//   LOAD(t,0): 16 fragment reads (rows 0-63 of the wave + its 64 pixels), DMA of the wave group's OWN weight half of
//              K-tile t+1                            | barrier | COMPUTE: 32 MFMAs | barrier
//   LOAD(t,1): 8 fragment reads (rows 64-127; the pixel fragments stay in registers), DMA of the pixel tile of K-tile
//              t+2 into the stage being consumed (its pixel half is dead after LOAD(t,0)), counted vmcnt(4)
//                                                    | barrier | COMPUTE: 32 MFMAs | barrier
// so one wave of every SIMD always has 32 MFMAs to issue from registers while its partner reads LDS and issues DMAs.
// LDS: 2 stages x (256 weight rows + 256 pixel rows) x 128 B, 16-byte pieces XOR-swizzled by (row >> 1) & 7 through
// the DMA's per-lane SOURCE address (conflict-free 16x16x32 fragment reads, same image as conv3_pdma's weight slabs).
// Hazards: a weight half is written and read by ONE wave group (program order + that group's barriers); a pixel tile is
// re-filled one LOAD after its last reads, which every wave retires (lgkmcnt(0)) before the barrier that ends its LOAD.
#include <algorithm>

#include "common.h"

namespace {

struct GemmTParams {
  const char* a;        // pixel operand: x (forward) / dy (data gradient), NHWC bf16
  const char* w;        // packed weights [rows][K] bf16
  char* out;            // y / dx, NHWC bf16
  const float* bias;    // forward: [Cout]
  int M;                // pixels of the low-resolution side: n * H * W
  int rows, K;          // GEMM rows (4*Cout / Cin) and depth (Cin / 4*Cout)
  int H, W, Cin, Cout;
  int nRt, nPt;         // row tiles, pixel tiles
};

constexpr int STAGE = 65536, PIX_BASE = 32768, DUMMY = 2 * STAGE, LDS_BYTES = 2 * STAGE + 1024;
constexpr unsigned OOB = 0xFFFFFFF0u;

template <bool DGRAD>
__global__ __launch_bounds__(512, 1) void convt_gemm_kernel(const GemmTParams P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void lds_void;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave >> 2, wq = wave & 3;
  const int l15 = lane & 15, kb = lane >> 4;

  // XCD x (= blockIdx % 8) owns a contiguous run of tiles; consecutive tiles share the pixel tile (row tile fastest)
  const int G = gridDim.x;
  const int logical = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  if (logical >= P.nRt * P.nPt) return;
  const int pt_i = logical / P.nRt, rt_i = logical - pt_i * P.nRt;
  const int r0 = rt_i * 256, m0 = pt_i * 256;
  const int nK = P.K / 64;
  const int HW = P.H * P.W;

  // ---- DMA geometry.  One wave-instruction = 8 rows x 128 B; lane -> (row = 8*instr + lane/8, LDS piece lane%8),
  // source piece = LDS piece ^ ((row >> 1) & 7).
  const long long a_bytes = DGRAD ? (long long)P.M * 4 * P.Cout * 2 : (long long)P.M * P.Cin * 2;
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)P.a, (short)0, (int)a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)P.w, (short)0, (int)((long long)P.rows * P.K * 2), 0x00020000);
  unsigned p_src[4], w_src[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    {   // pixel tile: 32 instructions over 8 waves
      const int row = (j * 8 + wave) * 8 + (lane >> 3), pos = lane & 7;
      const int m = m0 + row;
      const unsigned sw = (unsigned)((pos ^ ((row >> 1) & 7)) << 4);
      if (m >= P.M) p_src[j] = OOB;
      else if (!DGRAD) p_src[j] = (unsigned)m * (unsigned)P.Cin * 2u + sw;
      else {
        const int n = m / HW, r = m - n * HW;
        const int h = r / P.W, x = r - h * P.W;
        p_src[j] = (unsigned)(((n * 2 * P.H + 2 * h) * 2 * P.W + 2 * x)) * (unsigned)P.Cout * 2u + sw;
      }
    }
    {   // this wave group's weight half: 16 instructions over its 4 waves
      const int row = (j * 4 + wq) * 8 + (lane >> 3), pos = lane & 7;
      w_src[j] = (unsigned)(r0 + grp * 128 + row) * (unsigned)P.K * 2u + (unsigned)((pos ^ ((row >> 1) & 7)) << 4);
    }
  }
  auto dma_pix = [&](int kt, bool live) {
    unsigned soff;
    if (!DGRAD) soff = (unsigned)kt * 128u;
    else {
      const int k0 = kt * 64, z = k0 / P.Cout, c0 = k0 - z * P.Cout;
      soff = (unsigned)(((z >> 1) * 2 * P.W + (z & 1)) * P.Cout + c0) * 2u;
    }
    const int base = live ? (kt & 1) * STAGE + PIX_BASE + wave * 1024 : DUMMY;      // (uniform: M0)
    const int step = live ? 8192 : 0;
    if (!live) soff = 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_void*)(smem + base + j * step), 16, live ? p_src[j] : OOB, soff,
                                               0, 0);
  };
  auto dma_w = [&](int kt, bool live) {
    const int base = live ? (kt & 1) * STAGE + grp * 16384 + wq * 1024 : DUMMY;
    const int step = live ? 4096 : 0;
    const unsigned soff = live ? (unsigned)kt * 128u : 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)(smem + base + j * step), 16, live ? w_src[j] : OOB, soff,
                                               0, 0);
  };

  // ---- fragment addresses (bytes within a stage)
  int aoff[8][2], boff[4][2];
#pragma unroll
  for (int ct = 0; ct < 8; ++ct) {
    const int row = ct * 16 + l15;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) aoff[ct][ks] = grp * 16384 + row * 128 + (((ks * 4 + kb) ^ ((row >> 1) & 7)) << 4);
  }
#pragma unroll
  for (int pt = 0; pt < 4; ++pt) {
    const int row = wq * 64 + pt * 16 + l15;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) boff[pt][ks] = PIX_BASE + row * 128 + (((ks * 4 + kb) ^ ((row >> 1) & 7)) << 4);
  }

  f32x4 acc[8][4];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 fa[2][4], fb[2][4];

  // ---- prologue: K-tile 0 whole, the pixel tile of K-tile 1
  dma_pix(0, true);
  dma_w(0, true);
  dma_pix(1, nK > 1);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (grp) __builtin_amdgcn_s_barrier();                   // the stagger: waves 4-7 one barrier behind

  for (int kt = 0; kt < nK; ++kt) {
    const char* st = smem + (kt & 1) * STAGE;
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
      // ---- LOAD
      if (ph == 0) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int pt = 0; pt < 4; ++pt) fb[ks][pt] = *reinterpret_cast<const bf16x8*>(st + boff[pt][ks]);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) fa[ks][ct] = *reinterpret_cast<const bf16x8*>(st + aoff[ph * 4 + ct][ks]);
      __builtin_amdgcn_sched_barrier(0);
      if (ph == 0) {
        dma_w(kt + 1, kt + 1 < nK);
      } else {
        dma_pix(kt + 2, kt + 2 < nK);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // K-tile kt+1 has landed; only the 4 pieces just issued fly
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---- COMPUTE
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
          for (int pt = 0; pt < 4; ++pt)
            acc[ph * 4 + ct][pt] =
                __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ks][ct], fb[ks][pt], acc[ph * 4 + ct][pt], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (!grp) __builtin_amdgcn_s_barrier();                  // pairs with the stagger barrier of waves 4-7

  // ---- epilogue.  D of 16x16x32: column = lane & 15 (pixel), rows (lane >> 4) * 4 + reg.  v_permlane16_swap trades the
  // (kb odd) rows of tile ct for the (kb even) rows of tile ct + 1: a lane then holds 8 CONSECUTIVE rows -- tile
  // ct + (kb & 1), rows 8 * (kb >> 1) .. + 7 -- one 16-byte store per (pixel tile, tile pair).
  const long long o_bytes = DGRAD ? (long long)P.M * P.Cin * 2 : (long long)P.M * 4 * P.Cout * 2;
  const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)P.out, (short)0, (int)o_bytes, 0x00020000);
  const int rbase = r0 + grp * 128;                         // first GEMM row of this wave
  int z = 0, cbase = rbase;
  if (!DGRAD) { z = rbase / P.Cout; cbase = rbase - z * P.Cout; }
#pragma unroll
  for (int pt = 0; pt < 4; ++pt) {
    const int m = m0 + wq * 64 + pt * 16 + l15;
    unsigned pix_off = OOB;
    if (m < P.M) {
      if (DGRAD) pix_off = (unsigned)m * (unsigned)P.Cin * 2u;
      else {
        const int n = m / HW, r = m - n * HW;
        const int h = r / P.W, x = r - h * P.W;
        pix_off = (unsigned)((n * 2 * P.H + 2 * h + (z >> 1)) * 2 * P.W + 2 * x + (z & 1)) * (unsigned)P.Cout * 2u;
      }
    }
#pragma unroll
    for (int cp = 0; cp < 4; ++cp) {
      float va[4], vb[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { va[j] = acc[2 * cp][pt][j]; vb[j] = acc[2 * cp + 1][pt][j]; }
      if (!DGRAD && P.bias) {
        const float* bp = P.bias + cbase + cp * 32 + kb * 4;   // native layout: tile 2cp rows kb*4.., +16: tile 2cp+1
#pragma unroll
        for (int j = 0; j < 4; ++j) { va[j] += bp[j]; vb[j] += bp[16 + j]; }
      }
      bf16x4 ra, rb;
#pragma unroll
      for (int j = 0; j < 4; ++j) { ra[j] = (bf16_t)va[j]; rb[j] = (bf16_t)vb[j]; }
      const u32x2 ua = __builtin_bit_cast(u32x2, ra), ub = __builtin_bit_cast(u32x2, rb);
      const auto s0 = __builtin_amdgcn_permlane16_swap(ua[0], ub[0], false, false);
      const auto s1 = __builtin_amdgcn_permlane16_swap(ua[1], ub[1], false, false);
      const unsigned co = (unsigned)(cbase + cp * 32 + (kb & 1) * 16 + (kb >> 1) * 8);
      const unsigned vo = pix_off == OOB ? OOB : pix_off + co * 2u;
      __builtin_amdgcn_raw_buffer_store_b128(u32x4{s0[0], s1[0], s0[1], s1[1]}, o_rsrc, vo, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // drain the dummy DMAs before the wave ends
}

}  // namespace

bool unet_internal_convt_gemm_ok(int mode, int dtype, int n, int h, int w, int c_in, int c_out) {
  if (dtype != UNET_BF16 || unet_tuning().convt_impl == '0' || unet_tuning().convt_impl == '2') return false;
  const long long M = (long long)n * h * w;
  const long long big = std::max(M * c_in * 2, M * 4 * c_out * 2);
  if (big >= 0x7FFFFFFFLL || (long long)c_in * 4 * c_out * 2 >= 0x7FFFFFFFLL) return false;
  if (c_in % 64 != 0 || c_out % 64 != 0) return false;
  if (mode == 0) return c_out % 256 == 0 && c_in >= 128;                   // a row tile stays inside one (zy, zx)
  return c_in % 256 == 0 && c_out >= 64;
}

// mode 0: forward (a = x, out = y, w_packed = [4][c_out][c_in]); mode 1: data gradient (a = dy, out = dx,
// w_packed = [c_in][4 * c_out])
int32_t unet_internal_convt_gemm(int mode, int n, int h, int w, const void* a, const void* w_packed, const float* bias,
                                 void* out, int c_in, int c_out, hipStream_t s) {
  GemmTParams P{};
  P.a = (const char*)a; P.w = (const char*)w_packed; P.out = (char*)out; P.bias = bias;
  P.M = n * h * w; P.H = h; P.W = w; P.Cin = c_in; P.Cout = c_out;
  P.rows = mode == 0 ? 4 * c_out : c_in;
  P.K = mode == 0 ? c_in : 4 * c_out;
  P.nRt = P.rows / 256;
  P.nPt = cdiv(P.M, 256);
  const int tiles = P.nRt * P.nPt;
  const int blocks = cdiv(tiles, 8) * 8;
  const double flops = 2.0 * P.M * 4.0 * c_out * c_in;
  if (mode == 0) {
    auto kern = convt_gemm_kernel<false>;
    unet_set_max_lds(reinterpret_cast<const void*>(kern), LDS_BYTES);
    ProfScope prof(UNET_K_CONVT_FWD, flops, s, "convt_gemm_kernel");
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), LDS_BYTES, s, P);
  } else {
    auto kern = convt_gemm_kernel<true>;
    unet_set_max_lds(reinterpret_cast<const void*>(kern), LDS_BYTES);
    ProfScope prof(UNET_K_CONVT_DGRAD, flops, s, "convt_gemm_dgrad_kernel");
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), LDS_BYTES, s, P);
  }
  return unet_check_launch("convt_gemm_kernel");
}
