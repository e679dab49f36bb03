// Error reporting, version and the dispatch options of libalvq.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>

#include "alvq_common.h"

namespace alvq {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// Dispatch options (which kernel variant serves a launch; results are the same within each mode's stated precision).
// Initialised once from the environment (ALVQ_<NAME>), changed at run time through alvq_set_option -- the hot path reads
// an atomic, never the environment.
struct OptionSpec { const char* name; const char* env; long def; };
static const OptionSpec kSpecs[OPT_COUNT] = {
    {"wide_min_tiles", "ALVQ_WIDE_MIN_TILES", 192},   // bf16: 256 x 256-tile kernels only from this many tiles (about one per CU)
    {"fx_rows", "ALVQ_FX_ROWS", 0},                   // f16mx conv row tile: 0 automatic, 128 / 256 forced
    {"fx_narrow", "ALVQ_FX_NARROW", 1},               // f16mx: 128-channel m-tile for fp32-NCL outputs of <= 128 channels
    {"conv_v2", "ALVQ_CONV_V2", 1},                   // bf16: the 256 x 256-tile kernels at all
    {"conv_k3", "ALVQ_CONV_K3", 1},                   // bf16: the shared-slab width-3 kernel (0: the generic 256 x 256 one)
    {"wgrad_v3", "ALVQ_WGRAD_V3", 3},                 // bf16 weight gradient without bias: v3 kernels for width 1 (1) / width 3 (2)
    {"vq_reg", "ALVQ_VQ_REG", 1},                     // quantiser argmin: x rows in registers (D <= 256); 0 = the LDS-stationary kernel
};
static std::atomic<long> g_opt[OPT_COUNT];
static std::atomic<int> g_opt_ready{0};
static void options_init() {
  if (g_opt_ready.load(std::memory_order_acquire)) return;
  for (int i = 0; i < OPT_COUNT; ++i) {
    const char* e = getenv(kSpecs[i].env);
    g_opt[i].store(e ? atol(e) : kSpecs[i].def, std::memory_order_relaxed);
  }
  g_opt_ready.store(1, std::memory_order_release);
}
long option(int id) {
  options_init();
  return g_opt[id].load(std::memory_order_relaxed);
}
}  // namespace alvq

extern "C" const char* alvq_version(void) { return "alvq 0.4.0 (gfx950)"; }
extern "C" const char* alvq_last_error(void) { return alvq::g_err; }

extern "C" int alvq_set_option(const char* name, int64_t value) {
  ALVQ_REQUIRE(name, ALVQ_EINVAL, "alvq_set_option: null name");
  alvq::options_init();
  for (int i = 0; i < alvq::OPT_COUNT; ++i)
    if (!strcmp(name, alvq::kSpecs[i].name)) {
      alvq::g_opt[i].store((long)value, std::memory_order_relaxed);
      return ALVQ_OK;
    }
  alvq::set_error("alvq_set_option: unknown option '%s'", name);
  return ALVQ_EINVAL;
}

extern "C" int64_t alvq_get_option(const char* name) {
  if (name) {
    alvq::options_init();
    for (int i = 0; i < alvq::OPT_COUNT; ++i)
      if (!strcmp(name, alvq::kSpecs[i].name)) return alvq::g_opt[i].load(std::memory_order_relaxed);
  }
  alvq::set_error("alvq_get_option: unknown option '%s'", name ? name : "(null)");
  return INT64_MIN;
}
