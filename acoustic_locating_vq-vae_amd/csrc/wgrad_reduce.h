// Shared by the fp32 and bf16 weight-grad paths (each translation unit gets its own copy of the kernel).
#pragma once
#include "alvq_common.h"

namespace alvq {

// dw (+)= sum_s partial[s]; fixed summation order -> bitwise reproducible.
// OIK: dw[m][c][t].   IOK: dw[c][m][KW-1-t].
// scale (device scalar or null): multiplied into the sum before it is stored / accumulated (undoes a loss scale).
// Body shared by the single launch and the batched one (wgrad_reduce_batch.hip): block `bid` of `nblk` 256-thread blocks.
// `stride`: elements between consecutive partials (KW * M * C unless the partials are padded, e.g. bias sums [splits][Mp]).
static __device__ __forceinline__ void wgrad_reduce_body(const float* partial, float* dw, int splits, int KW, int M, int C,
                                                         int w_layout, int accumulate, const float* scale, long stride, int bid,
                                                         int nblk) {
  const long total = (long)KW * M * C;
  const float sc = scale ? *scale : 1.f;
  if (w_layout == ALVQ_W_OIK && KW == 3) {
    // One (m, c) pair per thread, its three taps together: the partials [split][t][m][c] are read with lanes along c
    // (whole 256-byte runs per wave and tap, splits x 3 independent loads in flight) and the three outputs of a pair are
    // adjacent in dw[m][c][t].  Indexing the OUTPUT instead (below) made neighbouring lanes read three different tap
    // planes with a stride of three lanes: the batched reduction of a step ran at 3.9 TB/s.  Same sums, same split order.
    const long pairs = (long)M * C;
    for (long q = bid * 256L + threadIdx.x; q < pairs; q += (long)nblk * 256) {
      float s0 = 0.f, s1 = 0.f, s2 = 0.f;
      const float* p = partial + q;
      int k = 0;
      for (; k + 2 <= splits; k += 2) {
        const float* pa = p + (long)k * stride;
        const float* pb = pa + stride;
        const float a0 = pa[0], a1 = pa[pairs], a2 = pa[2 * pairs], b0 = pb[0], b1 = pb[pairs], b2 = pb[2 * pairs];
        s0 = (s0 + a0) + b0;
        s1 = (s1 + a1) + b1;
        s2 = (s2 + a2) + b2;
      }
      for (; k < splits; ++k) {
        const float* pa = p + (long)k * stride;
        s0 += pa[0];
        s1 += pa[pairs];
        s2 += pa[2 * pairs];
      }
      if (scale) { s0 *= sc; s1 *= sc; s2 *= sc; }
      float* o = dw + 3 * q;
      if (accumulate) { s0 += o[0]; s1 += o[1]; s2 += o[2]; }
      o[0] = s0; o[1] = s1; o[2] = s2;
    }
    return;
  }
  for (long e = bid * 256L + threadIdx.x; e < total; e += (long)nblk * 256) {
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
    int k = 0;
    for (; k + 4 <= splits; k += 4) {    // four loads in flight; same summation order as one at a time
      const float v0 = partial[(long)k * stride + src], v1 = partial[(long)(k + 1) * stride + src],
                  v2 = partial[(long)(k + 2) * stride + src], v3 = partial[(long)(k + 3) * stride + src];
      s = (((s + v0) + v1) + v2) + v3;
    }
    for (; k < splits; ++k) s += partial[(long)k * stride + src];
    if (scale) s *= sc;
    dw[e] = accumulate ? dw[e] + s : s;
  }
}
static __global__ void wgrad_reduce_kernel(const float* partial, float* dw, int splits, int KW, int M, int C, int w_layout,
                                    int accumulate, const float* scale) {
  wgrad_reduce_body(partial, dw, splits, KW, M, C, w_layout, accumulate, scale, (long)KW * M * C, blockIdx.x, gridDim.x);
}

// IOK output (ConvTranspose1d weights): the loop above would read the partials with a stride of C floats between
// neighbouring lanes.  Here a workgroup owns an RB (m) x 64 (c) tile for all taps: partials are read with lanes along c
// (their contiguous axis), summed in split order, and the tile is turned through LDS so that the writes are
// (m, tap) runs of one c -- the contiguous axis of dw[c][m][KW-1-t].
// RB rows (m) x 64 columns (c) per workgroup; RB = 16 gives a 1024 x 1024 weight 1024 workgroups of 256 threads with
// RB / 8 x KW 8-byte loads in flight each -- at RB = 32 (512 workgroups) the kernel ran at 2 TB/s.
template <int RB>
static __device__ __forceinline__ void wgrad_reduce_iok_body(float (&tile)[3][RB][65], const float* partial, float* dw, int splits,
                                                             int KW, int M, int C, int accumulate, const float* scale, int bid) {
  constexpr int RI = RB / 8;           // rows per thread
  const int ctiles = (C + 63) / 64;
  const int m0 = (bid / ctiles) * RB, c0 = (bid % ctiles) * 64;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const long plane = (long)M * C, total = (long)KW * plane;
  const bool pair = (C % 2 == 0);          // 8-byte loads need an even row stride
  // all (tap, row) sums of a thread advance together, one split at a time: RI x KW independent loads in flight
  float s0[3][RI], s1[3][RI];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int i = 0; i < RI; ++i) s0[t][i] = s1[t][i] = 0.f;
  const int c = c0 + 2 * tx;
  const bool full = pair && m0 + RB <= M && c0 + 64 <= C;     // interior tile: no per-element bounds, loads batch up
  if (full) {
    const float* p0 = partial + (long)(m0 + ty) * C + c;
    if (KW == 3) {
#pragma unroll 4
      for (int k = 0; k < splits; ++k) {
        float2 v[3][RI];
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int i = 0; i < RI; ++i) v[t][i] = *(const float2*)(p0 + (long)k * total + (long)t * plane + (long)(8 * i) * C);
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int i = 0; i < RI; ++i) {
            s0[t][i] += v[t][i].x;
            s1[t][i] += v[t][i].y;
          }
      }
    } else {
#pragma unroll 4
      for (int k = 0; k < splits; ++k) {
        float2 v[RI];
#pragma unroll
        for (int i = 0; i < RI; ++i) v[i] = *(const float2*)(p0 + (long)k * total + (long)(8 * i) * C);
#pragma unroll
        for (int i = 0; i < RI; ++i) {
          s0[0][i] += v[i].x;
          s1[0][i] += v[i].y;
        }
      }
    }
  } else
  for (int k = 0; k < splits; ++k) {
    const float* pk = partial + (long)k * total + c;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      if (t >= KW) break;
#pragma unroll
      for (int i = 0; i < RI; ++i) {
        const int m = m0 + ty + 8 * i;
        if (m >= M) continue;
        const float* p = pk + (long)t * plane + (long)m * C;
        if (pair && c + 1 < C) {
          const float2 v = *(const float2*)p;
          s0[t][i] += v.x;
          s1[t][i] += v.y;
        } else {
          if (c < C) s0[t][i] += p[0];
          if (c + 1 < C) s1[t][i] += p[1];
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    if (t >= KW) break;
#pragma unroll
    for (int i = 0; i < RI; ++i) {
      tile[t][ty + 8 * i][2 * tx] = s0[t][i];
      tile[t][ty + 8 * i][2 * tx + 1] = s1[t][i];
    }
  }
  __syncthreads();
  const int run = RB * KW;
  for (int e = threadIdx.x; e < 64 * run; e += 256) {
    const int cc = e / run, j = e - cc * run, mr = j / KW, tt = j - mr * KW;
    if (c0 + cc >= C || m0 + mr >= M) continue;
    const long o = ((long)(c0 + cc) * M + m0) * KW + j;
    float v = tile[KW - 1 - tt][mr][cc];
    if (scale) v *= *scale;
    dw[o] = accumulate ? dw[o] + v : v;
  }
}
template <int RB>
static __global__ __launch_bounds__(256) void wgrad_reduce_iok_kernel(const float* partial, float* dw, int splits, int KW,
                                                                      int M, int C, int accumulate, const float* scale) {
  __shared__ float tile[3][RB][65];
  wgrad_reduce_iok_body<RB>(tile, partial, dw, splits, KW, M, C, accumulate, scale, blockIdx.x);
}

// Split plan of the NLC weight-gradient kernels (bf16, bf16x3, f16mx): the contraction runs over `total_rows` (a multiple
// of 64; nseg * rows for a multi-segment launch) and is cut into about 256 / tiles ranges (one workgroup per CU), at most 64.
static inline int wgrad_split_plan(int total_rows, int tiles, int* chunks_per_split) {
  const int nchunks = total_rows / 64;
  int want = (256 + tiles - 1) / tiles;
  if (want < 1) want = 1;
  if (want > nchunks) want = nchunks;
  if (want > 64) want = 64;
  const int cps = (nchunks + want - 1) / want;
  *chunks_per_split = cps;
  return (nchunks + cps - 1) / cps;
}

// Largest split count any launch over 1..maxseg segments of `rows` rows can use.  ceil(n / ceil(n / want)) is NOT monotone
// in n (round-2 advisor finding: sizing for maxseg * rows alone under-sized the 3-segment launches), so take the maximum.
static inline int wgrad_split_bound(int rows, int tiles, int maxseg) {
  int best = 1, cps;
  for (int s = 1; s <= maxseg; ++s) {
    const int k = wgrad_split_plan(s * rows, tiles, &cps);
    if (k > best) best = k;
  }
  return best;
}

static inline void wgrad_reduce_launch(const float* partial, float* dw, int splits, int KW, int M, int C, int w_layout,
                                       int accumulate, hipStream_t s, const float* scale = nullptr) {
  if (w_layout == ALVQ_W_IOK) {
    const int grid = ((M + 15) / 16) * ((C + 63) / 64);
    hipLaunchKernelGGL(wgrad_reduce_iok_kernel<16>, dim3(grid), dim3(256), 0, s, partial, dw, splits, KW, M, C, accumulate, scale);
    return;
  }
  const long total = (long)KW * M * C;
  int rgrid = (int)((total + 255) / 256);
  if (rgrid > 2048) rgrid = 2048;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rgrid), dim3(256), 0, s, partial, dw, splits, KW, M, C, w_layout, accumulate, scale);
}

}  // namespace alvq
