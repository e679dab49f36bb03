// Shared host/device helpers for libalvq (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/alvq.h"

namespace alvq {

void set_error(const char* fmt, ...);

// dispatch options (api.hip): environment-initialised, run-time settable through alvq_set_option
enum Option { OPT_WIDE_MIN_TILES, OPT_FX_ROWS, OPT_FX_NARROW, OPT_CONV_V2, OPT_CONV_K3, OPT_WGRAD_V3, OPT_VQ_REG, OPT_COUNT };
long option(int id);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return ALVQ_OK;
}

#define ALVQ_REQUIRE(cond, code, ...)  \
  do {                                 \
    if (!(cond)) {                     \
      alvq::set_error(__VA_ARGS__);    \
      return (code);                   \
    }                                  \
  } while (0)

// hipFuncSetAttribute applies to the current device only: a launch site remembers which devices it has configured
// (a process may drive more than one card; the kernels that ask for more than 64 KB of LDS fail to launch otherwise)
struct DeviceOnce {
  bool done[64] = {};
  bool need() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return true;
    if (done[d]) return false;
    done[d] = true;
    return true;
  }
};

constexpr int kWave = 64;  // CDNA wavefront
constexpr int kNumXcd = 8;

// Bijective XCD-aware remap (guide T1): hardware deals consecutive workgroup ids round-robin over the
// 8 XCDs; give each XCD a contiguous run of logical tile ids so neighbours share operands in its L2.
__device__ __forceinline__ int xcd_remap(int id, int n) {
  const int q = n / kNumXcd, r = n % kNumXcd;
  const int xcd = id % kNumXcd, local = id / kNumXcd;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + local;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

// defined in conv1d_f16mx.hip: device addresses (current device) of the fp16-range formats' range flag -- the bits raised
// since the last guarded alvq_adam_advance_f32 / since it was read -- and of the sticky word the former are folded into
int* fx_range_flag_ptr();
int* fx_range_sticky_ptr();

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace alvq
