// Shared by the fp32 and bf16 weight-grad paths (each translation unit gets its own copy of the kernel).
#pragma once
#include "alvq_common.h"

namespace alvq {

// dw (+)= sum_s partial[s]; fixed summation order -> bitwise reproducible.
// OIK: dw[m][c][t].   IOK: dw[c][m][KW-1-t].
static __global__ void wgrad_reduce_kernel(const float* partial, float* dw, int splits, int KW, int M, int C, int w_layout,
                                    int accumulate) {
  const long total = (long)KW * M * C;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    // e indexes the OUTPUT (coalesced writes); decode to (m, c, t)
    int m, c, t;
    if (w_layout == ALVQ_W_OIK) {
      t = (int)(e % KW);
      c = (int)((e / KW) % C);
      m = (int)(e / ((long)KW * C));
    } else {
      const int tt = (int)(e % KW);
      t = KW - 1 - tt;
      m = (int)((e / KW) % M);
      c = (int)(e / ((long)KW * M));
    }
    const long src = ((long)t * M + m) * C + c;
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += partial[(long)k * total + src];
    dw[e] = accumulate ? dw[e] + s : s;
  }
}


}  // namespace alvq
