// Status reporting + per-kernel-class hipEvent timing for libunet_hip.so.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

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
