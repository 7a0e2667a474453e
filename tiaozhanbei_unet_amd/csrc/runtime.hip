// Status reporting + per-kernel-class hipEvent timing for libunet_hip.so.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "common.h"

static thread_local char g_err[512] = "";

void unet_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int32_t unet_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    unet_set_error("%s: %s", what, hipGetErrorString(e));
    return UNET_ERR_LAUNCH;
  }
  return UNET_OK;
}

extern "C" int32_t unet_abi_version(void) { return UNET_ABI_VERSION; }
extern "C" const char* unet_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------ tuning hooks, LDS opt-in
namespace {
UnetTuning g_tuning{};
std::once_flag g_tuning_once;
void read_tuning() {
  auto first = [](const char* name) -> char { const char* v = getenv(name); return (v && v[0]) ? v[0] : (char)0; };
  g_tuning.conv_impl = first("UNET_CONV_IMPL");
  g_tuning.conv_var = first("UNET_CONV_VAR");
  g_tuning.fused_stats = first("UNET_FUSED_STATS");
  g_tuning.convt_impl = first("UNET_CONVT_IMPL");
  g_tuning.wgrad_impl = first("UNET_WGRAD_IMPL");
  g_tuning.ws_stats = first("UNET_WS_STATS");
  g_tuning.dgrad_bn = first("UNET_DGRAD_BN");
  g_tuning.pdma_pp = first("UNET_PDMA_PP");
  g_tuning.ws_st = first("UNET_WS_ST");
  g_tuning.ws_mfma = first("UNET_WS_MFMA");
  g_tuning.wgrad_xcd = first("UNET_WGRAD_XCD");
  g_tuning.conv_xcd = first("UNET_CONV_XCD");
  g_tuning.ws_stg = first("UNET_WS_STG");
  g_tuning.pdma_stg = first("UNET_PDMA_STG");
  g_tuning.ew_var = first("UNET_EW_VAR");
  g_tuning.pdma_pair = first("UNET_PDMA_PAIR");
}
std::mutex g_lds_mu;
std::vector<std::pair<int, const void*>> g_lds_done;
}  // namespace

const UnetTuning& unet_tuning() {
  std::call_once(g_tuning_once, read_tuning);
  return g_tuning;
}

extern "C" int32_t unet_tuning_reload(void) {
  (void)unet_tuning();
  read_tuning();
  return UNET_OK;
}

namespace {
std::atomic<int> g_reserved_cus{-1};             // -1: not set -> UNET_RESERVED_CUS (default 0)
int g_dev_cus[64] = {0};
}  // namespace

int unet_cu_budget() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) dev = 0;
  int cus = g_dev_cus[dev];
  if (cus <= 0) {
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    g_dev_cus[dev] = cus;
  }
  int r = g_reserved_cus.load();
  if (r < 0) {
    const char* v = getenv("UNET_RESERVED_CUS");
    r = v ? atoi(v) : 0;
    if (r < 0) r = 0;
    g_reserved_cus.store(r);
  }
  int b = (cus - r) / 8 * 8;
  return b < 8 ? 8 : b;
}

extern "C" int32_t unet_set_reserved_cus(int32_t n) {
  UNET_REQUIRE(n >= 0 && n < 1024, UNET_ERR_BAD_ARG, "unet_set_reserved_cus: %d", n);
  g_reserved_cus.store(n);
  return UNET_OK;
}

extern "C" int32_t unet_get_cu_budget(void) { return unet_cu_budget(); }

void unet_set_max_lds(const void* kernel, int bytes) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> l(g_lds_mu);
  for (auto& e : g_lds_done)
    if (e.first == dev && e.second == kernel) return;
  (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  g_lds_done.emplace_back(dev, kernel);
}

// ------------------------------------------------------------------------------ profiling
namespace {
struct Rec { int k; double flops; hipEvent_t a, b; const char* name; double bytes; };
struct KStat { const char* name; double ms; long long launches; double flops; double bytes; };
std::vector<KStat> g_kstats;                      // per-kernel aggregation of the last unet_prof_collect()
std::mutex g_mu;
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
thread_local hipEvent_t t_start = nullptr;

hipEvent_t get_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
}  // namespace

void unet_prof_begin(int, hipStream_t s) {
  if (!g_on) return;
  std::lock_guard<std::mutex> l(g_mu);
  t_start = get_event();
  if (t_start) (void)hipEventRecord(t_start, s);
}

void unet_prof_end(int k, double flops, hipStream_t s, const char* kernel, double bytes) {
  if (!g_on || !t_start) return;
  std::lock_guard<std::mutex> l(g_mu);
  hipEvent_t b = get_event();
  if (!b) return;
  (void)hipEventRecord(b, s);
  g_recs.push_back({k, flops, t_start, b, kernel, bytes});
  t_start = nullptr;
}

extern "C" int32_t unet_prof_enable(int32_t on) {
  std::lock_guard<std::mutex> l(g_mu);
  g_on = on != 0;
  return UNET_OK;
}

extern "C" int32_t unet_prof_collect(double* ms, int64_t* launches, double* flops) {
  UNET_REQUIRE(ms && launches && flops, UNET_ERR_BAD_ARG, "unet_prof_collect: null output");
  std::lock_guard<std::mutex> l(g_mu);
  for (int i = 0; i < UNET_K_COUNT; ++i) { ms[i] = 0; launches[i] = 0; flops[i] = 0; }
  g_kstats.clear();
  for (auto& r : g_recs) {
    float t = 0.f;
    if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess &&
        r.k >= 0 && r.k < UNET_K_COUNT) {
      ms[r.k] += t; launches[r.k] += 1; flops[r.k] += r.flops;
      if (r.name) {
        KStat* ks = nullptr;
        for (auto& e : g_kstats) if (e.name == r.name || strcmp(e.name, r.name) == 0) { ks = &e; break; }
        if (!ks) { g_kstats.push_back({r.name, 0.0, 0, 0.0, 0.0}); ks = &g_kstats.back(); }
        ks->ms += t; ks->launches += 1; ks->flops += r.flops; ks->bytes += r.bytes;
      }
    }
    g_pool.push_back(r.a);
    g_pool.push_back(r.b);
  }
  g_recs.clear();
  return UNET_OK;
}

extern "C" int32_t unet_prof_kernel_stats(int32_t index, const char** name, double* ms, int64_t* launches, double* flops) {
  UNET_REQUIRE(name && ms && launches && flops, UNET_ERR_BAD_ARG, "unet_prof_kernel_stats: null output");
  std::lock_guard<std::mutex> l(g_mu);
  UNET_REQUIRE(index >= 0 && index < (int)g_kstats.size(), UNET_ERR_BAD_ARG, "unet_prof_kernel_stats: index %d of %zu", index,
               g_kstats.size());
  *name = g_kstats[index].name; *ms = g_kstats[index].ms; *launches = g_kstats[index].launches; *flops = g_kstats[index].flops;
  return UNET_OK;
}

extern "C" int32_t unet_prof_kernel_bytes(int32_t index, double* bytes) {
  UNET_REQUIRE(bytes, UNET_ERR_BAD_ARG, "unet_prof_kernel_bytes: null output");
  std::lock_guard<std::mutex> l(g_mu);
  UNET_REQUIRE(index >= 0 && index < (int)g_kstats.size(), UNET_ERR_BAD_ARG, "unet_prof_kernel_bytes: index %d of %zu", index,
               g_kstats.size());
  *bytes = g_kstats[index].bytes;
  return UNET_OK;
}

// ------------------------------------------------------------------------------ diagnostic: CU occupier
// tools/cu_share_probe.py: `blocks` workgroups that each hold `lds_bytes` of LDS and spin for `microseconds` -- a
// stand-in for the RCCL all-reduce kernels a data-parallel run keeps resident on a few CUs during the backward pass
// (one block per channel, tens of KiB of LDS each).  A CU that hosts one cannot take a persistent conv block (125-160 KiB
// of LDS), so a launch sized for every CU runs a second round; unet_set_reserved_cus() sizes the launches for the rest.
namespace {
__global__ __launch_bounds__(256) void spin_kernel(unsigned long long ticks, int lds_words) {
  extern __shared__ unsigned spin_lds[];
  if (lds_words > 0) spin_lds[threadIdx.x % lds_words] = threadIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();          // 100 MHz
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (lds_words > 0 && spin_lds[0] == 0xFFFFFFFFu) __builtin_trap();       // (keeps the LDS allocation alive)
}
}  // namespace

extern "C" int32_t unet_debug_spin(int32_t blocks, int32_t lds_bytes, int32_t microseconds, void* stream) {
  UNET_REQUIRE(blocks > 0 && blocks <= 1024 && lds_bytes >= 0 && lds_bytes <= 64 * 1024 && microseconds > 0 &&
                   microseconds <= 2000000, UNET_ERR_BAD_ARG, "unet_debug_spin: bad arguments");
  hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(256), (size_t)lds_bytes, (hipStream_t)stream,
                     (unsigned long long)microseconds * 100ull, lds_bytes / 4);
  return unet_check_launch("spin_kernel");
}
