// Status reporting + per-kernel-class hipEvent timing for libunet_hip.so.
#include <stdarg.h>
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

// ------------------------------------------------------------------------------ profiling
namespace {
struct Rec { int k; double flops; hipEvent_t a, b; };
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

void unet_prof_end(int k, double flops, hipStream_t s) {
  if (!g_on || !t_start) return;
  std::lock_guard<std::mutex> l(g_mu);
  hipEvent_t b = get_event();
  if (!b) return;
  (void)hipEventRecord(b, s);
  g_recs.push_back({k, flops, t_start, b});
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
  for (auto& r : g_recs) {
    float t = 0.f;
    if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess &&
        r.k >= 0 && r.k < UNET_K_COUNT) {
      ms[r.k] += t; launches[r.k] += 1; flops[r.k] += r.flops;
    }
    g_pool.push_back(r.a);
    g_pool.push_back(r.b);
  }
  g_recs.clear();
  return UNET_OK;
}
