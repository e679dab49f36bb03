// Probe (diagnostic tool, not product code): HBM write rate of the f16mx epilogue's store pattern against fully
// coalesced stores of the same bytes, all CUs storing at once (one 256 x 256 output tile per workgroup, 8 waves).
//   hipcc --offload-arch=gfx950 -O2 tools/store_pattern.hip -o acoustic_locating_vq-vae_amd/build/store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// tile = 256 rows x 256 channels; H plane [rows][Mp] u16, Q plane [rows][Mp/32][64 B].  MODE 0: the epilogue's pattern:
// lane (j = l & 31, h = l >> 5) of wave (wm = w >> 2, wn = w & 3) stores, for block (mi, ni): row wn*64 + ni*32 + j,
// channels wm*128 + mi*32 + 16 h .. +15 -> H 2 x 16 B, hi8 16 B, lo8 16 B.  MODE 1: same bytes, each wave-instruction a
// run of whole rows' segments: lane l -> row (l >> 3), 16-B piece (l & 7) of a 128-B line.
template <int MODE>
__global__ __launch_bounds__(512) void store_kernel(unsigned short* H, unsigned char* Q, int Mp, int mtiles, unsigned v) {
  const int tile = blockIdx.x, m0 = (tile % mtiles) * 256;
  const long r0 = (long)(tile / mtiles) * 256;
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const u32x4 val = {v, v + 1, v + 2, v + 3};
  if (MODE == 0) {
    const int j = l & 31, h = l >> 5, wm = w >> 2, wn = w & 3;
    for (int ni = 0; ni < 2; ++ni)
      for (int mi = 0; mi < 4; ++mi) {
        const long row = r0 + wn * 64 + ni * 32 + j;
        const int c = m0 + wm * 128 + mi * 32 + 16 * h;
        *(u32x4*)(H + row * Mp + c) = val;
        *(u32x4*)(H + row * Mp + c + 8) = val;
        unsigned char* q = Q + row * Mp * 2 + (long)(c >> 5) * 64 + (c & 31);
        *(u32x4*)q = val;
        *(u32x4*)(q + 32) = val;
      }
  } else {
    // H tile: 256 rows x 512 B = 4 lines per row; a wave stores 8 rows x 128 B per instruction
    for (int it = 0; it < 16; ++it) {          // 8 waves x 16 iterations x 8 rows = 1024 row-lines = 256 rows x 4
      const int idx = (w * 16 + it) * 8 + (l >> 3);
      const long row = r0 + (idx >> 2);
      const int line = idx & 3;
      *(u32x4*)((unsigned char*)(H + row * Mp + m0) + line * 128 + (l & 7) * 16) = val;
      *(u32x4*)(Q + row * Mp * 2 + (long)(m0 >> 5) * 64 + line * 128 + (l & 7) * 16) = val;
    }
  }
}

int main() {
  const int rows = 32256, Mp = 1024, mtiles = 4, tiles = (rows / 256) * mtiles;
  unsigned short* H; unsigned char* Q;
  CK(hipMalloc(&H, (size_t)rows * Mp * 2)); CK(hipMalloc(&Q, (size_t)rows * Mp * 2));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 2; ++mode)
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      for (int i = 0; i < 10; ++i) {
        if (mode == 0) hipLaunchKernelGGL(store_kernel<0>, dim3(tiles), dim3(512), 0, 0, H, Q, Mp, mtiles, (unsigned)i);
        else hipLaunchKernelGGL(store_kernel<1>, dim3(tiles), dim3(512), 0, 0, H, Q, Mp, mtiles, (unsigned)i);
      }
      CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("mode %d (%s): %.1f us per pass of %.0f MB = %.2f TB/s\n", mode, mode ? "whole 128-B lines per instruction" : "epilogue pattern (row per lane)",
             ms * 100.0, rows * (double)Mp * 4 / 1e6, rows * (double)Mp * 4 / (ms / 10 * 1e-3) / 1e12);
    }
  return 0;
}
