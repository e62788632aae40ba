// Version / error strings of the C ABI.
#include "common.h"

extern "C" int jg_version(void) { return 100; }

extern "C" const char* jg_strerror(int code) {
  switch (code) {
    case JG_OK: return "ok";
    case JG_ERR_BAD_ARG: return "bad argument (null pointer, misaligned leading dimension or unsupported shape)";
    case JG_ERR_UNSUPPORTED: return "unsupported configuration";
    case JG_ERR_LAUNCH: return "HIP launch failed";
    default: return "unknown error";
  }
}

// ---- dispatch switches --------------------------------------------------------------------------------
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace {
struct TuneDef { const char* env; int dflt; };
const TuneDef kTune[JG_TUNE_COUNT] = {
    {"JG_HALO_CFG", 0}, {"JG_WGRAD_HALO_CFG", 0}, {"JG_CONV_VARIANT", 6}, {"JG_WGRAD_VARIANT", 4}, {"JG_SINKHORN_GENERIC", 0},
    {"JG_CONV1X1", 1}, {"JG_GN_REVERSE", 1}, {"JG_HALO_DBG", 0}, {"JG_PERSIST64", 1}, {"JG_HALO_PIPE", 1}, {"JG_WGRAD_PIPE", 1}, {"JG_CONV_SPLITK", 1}, {"JG_CONV_SMALL_TILE", 1}, {"JG_GN_FUSED", 1}, {"JG_GN_FUSED_CAP", 256}, {"JG_GN_FUSED_DBG", 0}, {"JG_GN_FUSED_SLEEP", 4}, {"JG_WGRAD_LDS_PAD", 0},
    {"JG_LN_BWD_CAP", 256}, {"JG_DW_BWD_CAP", 512}, {"JG_DW_BWD_PPT", 8}, {"JG_CONV_KXK", 1}, {"JG_CONV_RING", 1}, {"JG_WGRAD_DEEP", 1}, {"JG_WGRAD_SW", 0}, {"JG_DETERMINISTIC", 0}, {"JG_WGRAD_GROUP_BLOCKS", 1024}, {"JG_SGEMM_SPLIT", 0}, {"JG_WGRAD_BIG", 1}, {"JG_DW_RUN", 1}};
std::atomic<int> g_tune[JG_TUNE_COUNT];
std::once_flag g_tune_once;
void tune_init() {
  for (int i = 0; i < JG_TUNE_COUNT; ++i) {
    const char* e = getenv(kTune[i].env);
    g_tune[i].store(e ? atoi(e) : kTune[i].dflt, std::memory_order_relaxed);
  }
}
}  // namespace

namespace {
thread_local const char* g_last_kernel = "";
}
void jg_note_kernel(const char* name) { g_last_kernel = name ? name : ""; }
extern "C" const char* jg_last_kernel(void) { return g_last_kernel; }

int jg_tune(int which) {
  std::call_once(g_tune_once, tune_init);
  return g_tune[which].load(std::memory_order_relaxed);
}

extern "C" int jg_set_tuning(const char* name, int value) {
  if (!name) return JG_ERR_BAD_ARG;
  std::call_once(g_tune_once, tune_init);
  for (int i = 0; i < JG_TUNE_COUNT; ++i)
    if (!strcmp(name, kTune[i].env)) {
      g_tune[i].store(value, std::memory_order_relaxed);
      return JG_OK;
    }
  return JG_ERR_BAD_ARG;
}

extern "C" int jg_get_tuning(const char* name) {
  if (!name) return -1;
  for (int i = 0; i < JG_TUNE_COUNT; ++i)
    if (!strcmp(name, kTune[i].env)) return jg_tune(i);
  return -1;
}
