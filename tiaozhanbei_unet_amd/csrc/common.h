// Shared device/host helpers for libunet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/unet_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define WAVE 64

// ---- host side status handling ---------------------------------------------------------
void unet_set_error(const char* fmt, ...);
int32_t unet_check_launch(const char* what);

#define UNET_REQUIRE(cond, code, ...)  \
  do {                                 \
    if (!(cond)) {                     \
      unet_set_error(__VA_ARGS__);     \
      return (code);                   \
    }                                  \
  } while (0)

// ---- tuning hooks: the UNET_* environment variables are read ONCE (first use); unet_tuning_reload() re-reads them
// (same-process A/B tools).  Each field is the first character of the variable's value, 0 when unset.
struct UnetTuning { char conv_impl, conv_var, fused_stats, convt_impl, wgrad_impl, ws_stats, dgrad_bn, pdma_pp, ws_st, ws_mfma, wgrad_xcd, conv_xcd, ws_stg, pdma_stg, ew_var, pdma_pair; };
const UnetTuning& unet_tuning();
// CUs the persistent (one-block-per-CU, statically partitioned) kernels may count on: the device's multiprocessor count
// minus unet_set_reserved_cus() (a data-parallel run leaves a few CUs to the RCCL all-reduce kernels that overlap the
// backward pass -- with all 256 claimed, a conv launch would wait for the CUs a collective holds and a static partition
// then takes up to twice as long), rounded down to a multiple of 8 (XCDs), at least 8
int unet_cu_budget();
// opt a kernel in to `bytes` of dynamic LDS -- once per (device, kernel), cheap afterwards
void unet_set_max_lds(const void* kernel, int bytes);

// ---- per-class event timing (prof.cpp) ---------------------------------------------------
void unet_prof_begin(int kclass, hipStream_t s);
void unet_prof_end(int kclass, double flops, hipStream_t s, const char* kernel, double bytes);
struct ProfScope {
  int k; double f; hipStream_t s; const char* name;      // name: static string naming the (dominant) kernel of the bracket
  double bytes;                                          // algorithmic bytes of the launch: inputs + weights + outputs, each once
  ProfScope(int kclass, double flops, hipStream_t st, const char* kernel = nullptr, double alg_bytes = 0.0)
      : k(kclass), f(flops), s(st), name(kernel), bytes(alg_bytes) {
    unet_prof_begin(k, s);
  }
  ~ProfScope() { unet_prof_end(k, f, s, name, bytes); }
};

// the deep transposed convolutions as one LDS-DMA GEMM (convt_gemm.hip); mode 0 forward, 1 data gradient
bool unet_internal_convt_gemm_ok(int mode, int dtype, int n, int h, int w, int c_in, int c_out);
int32_t unet_internal_convt_gemm(int mode, int n, int h, int w, const void* a, const void* w_packed, const float* bias,
                                 void* out, int c_in, int c_out, hipStream_t s);

// deterministic per-channel sum of x[pixels][C] (bn.hip); ws is scratch
int32_t unet_internal_colsum(int dtype, const void* x, int64_t pixels, int C, float* out, float* ws,
                              size_t ws_bytes, hipStream_t s);

// BatchNorm partial sums [parts][2][C] of y[pixels][C] by one streaming pass (bn.hip); parts <= 1024
int32_t unet_internal_bn_partials(int dtype, const void* y, int64_t pixels, int C, float* part, int* n_parts,
                                  hipStream_t s);

// ---- element traits ----------------------------------------------------------------------
template <typename T> struct ET;
template <> struct ET<float> {
  static constexpr int ES = 4;          // bytes per element
  static constexpr int PIECE = 4;       // elements per 16-byte piece
  static constexpr int KGC = 8;         // channels per 32-byte k-group
  __device__ static inline float to_f(float v) { return v; }
  __device__ static inline float from_f(float v) { return v; }
};
template <> struct ET<bf16_t> {
  static constexpr int ES = 2;
  static constexpr int PIECE = 8;
  static constexpr int KGC = 16;
  __device__ static inline float to_f(bf16_t v) { return (float)v; }
  __device__ static inline bf16_t from_f(float v) { return (bf16_t)v; }
};

// 16-byte vector load/store of PIECE elements, converted to/from fp32
template <typename T> struct Vec;
template <> struct Vec<float> {
  static constexpr int N = 4;
  __device__ static inline void load(const float* p, float (&o)[4]) {
    f32x4 v = *reinterpret_cast<const f32x4*>(p);
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
  }
  __device__ static inline void store(float* p, const float (&o)[4]) {
    f32x4 v = {o[0], o[1], o[2], o[3]};
    *reinterpret_cast<f32x4*>(p) = v;
  }
  // streaming load: a tensor read for the LAST time (or not again for milliseconds) should not displace the hot
  // operands of its neighbours from L2 / the Infinity Cache
  __device__ static inline void load_nt(const float* p, float (&o)[4]) {
    f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
  }
};
template <> struct Vec<bf16_t> {
  static constexpr int N = 8;
  __device__ static inline void load(const bf16_t* p, float (&o)[8]) {
    bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
  }
  __device__ static inline void store(bf16_t* p, const float (&o)[8]) {
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (bf16_t)o[i];
    *reinterpret_cast<bf16x8*>(p) = v;
  }
  __device__ static inline void load_nt(const bf16_t* p, float (&o)[8]) {
    const u32x4 r = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    const bf16x8 v = __builtin_bit_cast(bf16x8, r);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
  }
};

__host__ __device__ static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
