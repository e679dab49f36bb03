// Location head: LocationModule.fc_1 = Linear(L*K -> M) applied to the flattened one-hot codes of a
// spectrogram (vq_vae/location_model/location_model.py:10,21; fed by scripts/train_location.py:69-77 with
// encodings.reshape(B, 201, 1024)).  On one-hot rows the dense product is a sum of L selected weight columns,
//
//     out[b][m] = bias[m] + sum_l W[m][l*K + idx[b][l]]                (W is nn.Linear's (M, L*K) row-major weight)
//
// i.e. an embedding bag over the int indices the quantiser already produced: B*L*M gathered floats (13 MB at
// B=16, L=201, M=1024) instead of a (B x 205 824) x (205 824 x 1024) product that streams the 843 MB weight and a
// 13 MB operand that is 99.9 % zeros.  The backward is the matching scatter-add into the touched columns.
//
// Everything is HBM/L2-latency bound (scattered 4-byte accesses inside 823 KB weight rows); one wave owns one weight
// row, so there is no cross-wave conflict and every sum runs in a fixed order (bitwise reproducible).
#include "alvq_common.h"

namespace alvq {

// enc (rows, K) fp32 -> idx[row] = position of the single 1.0; *flag |= 1 if some row is not exactly one-hot
__global__ __launch_bounds__(256) void onehot_to_index_kernel(const float* enc, int32_t* idx, int* flag, long rows, int K) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* p = enc + row * K;
  int pos = -1, bad = 0, cnt = 0;
  for (int k = lane; k < K; k += 64) {
    const float v = p[k];
    if (v != 0.f) {
      ++cnt;
      pos = k;
      if (v != 1.f) bad = 1;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    cnt += __shfl_xor(cnt, o, 64);
    bad |= __shfl_xor(bad, o, 64);
    pos = max(pos, __shfl_xor(pos, o, 64));
  }
  if (lane == 0) {
    idx[row] = pos < 0 ? 0 : pos;
    if (cnt != 1 || bad) atomicOr(flag, 1);
  }
}

// (n) int64 indices -> int32, flagging any value outside [0, K) (a plain cast would wrap 2^32 + 5 to 5)
__global__ __launch_bounds__(256) void indices_to_i32_kernel(const int64_t* idx, int32_t* out, int* flag, long n, int K) {
  int bad = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int64_t v = idx[i];
    const bool ok = v >= 0 && v < K;
    out[i] = ok ? (int32_t)v : -1;
    bad |= !ok;
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// Stage the (B, L) index table in LDS; an index outside [0, K) becomes -1 (its term is skipped by both kernels: no
// out-of-bounds read of W, no out-of-bounds += into dW) and raises the sticky device flag the caller checks, as
// torch's embedding_bag raises on such input.
__device__ __forceinline__ void stage_indices(const int32_t* idx, int32_t* sidx, int n, int K, int* bad_index) {
  int bad = 0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int32_t v = idx[i];
    const bool ok = v >= 0 && v < K;
    sidx[i] = ok ? v : -1;
    bad |= !ok;
  }
  if (bad_index && blockIdx.x == 0 && __any(bad) && (threadIdx.x & 63) == 0) atomicOr(bad_index, 1);
  __syncthreads();
}

// one wave per weight row m; the (B, L) index table is staged in LDS once per workgroup (4 rows)
__global__ __launch_bounds__(256) void embedding_bag_fwd_kernel(const float* W, const float* bias, const int32_t* idx, float* out,
                                                                int B, int L, int K, int M, int* bad_index) {
  extern __shared__ int32_t sidx[];
  stage_indices(idx, sidx, B * L, K, bad_index);
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const float* w = W + (long)m * L * K;
  const float bm = bias ? bias[m] : 0.f;
  for (int b = 0; b < B; ++b) {
    float s = 0.f;
    for (int l = lane; l < L; l += 64) {
      const int k = sidx[b * L + l];
      s += k >= 0 ? w[(long)l * K + k] : 0.f;
    }
    s = wave_sum(s);
    if (lane == 0) out[(long)b * M + m] = s + bm;
  }
}

// dW[m][l*K + idx[b][l]] += dz[b][m] (dW zero-filled or accumulated into by the caller's choice); lane = l, b sequential
__global__ __launch_bounds__(256) void embedding_bag_bwd_kernel(const float* dz, const int32_t* idx, float* dW, float* dbias,
                                                                int B, int L, int K, int M, int accumulate_bias, int* bad_index) {
  extern __shared__ int32_t sidx[];
  stage_indices(idx, sidx, B * L, K, bad_index);
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  float* w = dW + (long)m * L * K;
  float sb = 0.f;
  for (int b = 0; b < B; ++b) {
    const float g = dz[(long)b * M + m];
    sb += g;
    for (int l = lane; l < L; l += 64) {     // same (l, column) from two samples: same lane, in order
      const int k = sidx[b * L + l];
      if (k >= 0) w[(long)l * K + k] += g;
    }
  }
  if (dbias && lane == 0) dbias[m] = accumulate_bias ? dbias[m] + sb : sb;
}

}  // namespace alvq

using namespace alvq;

extern "C" int alvq_onehot_to_index_f32(const float* encodings, int32_t* idx, int* not_onehot, int64_t rows, int K, void* stream) {
  ALVQ_REQUIRE(encodings && idx && not_onehot, ALVQ_EINVAL, "alvq_onehot_to_index_f32: null pointer");
  ALVQ_REQUIRE(rows > 0 && K > 0, ALVQ_EINVAL, "alvq_onehot_to_index_f32: bad dims");
  ALVQ_REQUIRE((rows + 3) / 4 < (1L << 31), ALVQ_EUNSUPPORTED, "alvq_onehot_to_index_f32: too many rows");
  hipLaunchKernelGGL(onehot_to_index_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, encodings, idx,
                     not_onehot, (long)rows, K);
  return check_launch("alvq_onehot_to_index_f32");
}

extern "C" int alvq_indices_to_i32(const int64_t* idx, int32_t* out, int* bad_index, int64_t n, int K, void* stream) {
  ALVQ_REQUIRE(idx && out && bad_index, ALVQ_EINVAL, "alvq_indices_to_i32: null pointer");
  ALVQ_REQUIRE(n > 0 && K > 0, ALVQ_EINVAL, "alvq_indices_to_i32: bad dims");
  long g = (n + 255) / 256;
  if (g > 1024) g = 1024;
  hipLaunchKernelGGL(indices_to_i32_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, idx, out, bad_index, (long)n, K);
  return check_launch("alvq_indices_to_i32");
}

extern "C" int alvq_embedding_bag_fwd_f32(const float* W, const float* bias, const int32_t* idx, float* out, int B, int L, int K,
                                          int M, int* bad_index, void* stream) {
  ALVQ_REQUIRE(W && idx && out, ALVQ_EINVAL, "alvq_embedding_bag_fwd_f32: null pointer");
  ALVQ_REQUIRE(B > 0 && L > 0 && K > 0 && M > 0, ALVQ_EINVAL, "alvq_embedding_bag_fwd_f32: bad dims");
  ALVQ_REQUIRE((long)B * L * 4 <= 64 * 1024, ALVQ_EUNSUPPORTED, "alvq_embedding_bag_fwd_f32: B*L = %ld indices exceed the 64 KB LDS table",
               (long)B * L);
  hipLaunchKernelGGL(embedding_bag_fwd_kernel, dim3((M + 3) / 4), dim3(256), (size_t)B * L * 4, (hipStream_t)stream, W, bias, idx, out,
                     B, L, K, M, bad_index);
  return check_launch("alvq_embedding_bag_fwd_f32");
}

extern "C" int alvq_embedding_bag_bwd_f32(const float* dz, const int32_t* idx, float* dW, float* dbias, int B, int L, int K, int M,
                                          int accumulate_bias, int* bad_index, void* stream) {
  ALVQ_REQUIRE(dz && idx && dW, ALVQ_EINVAL, "alvq_embedding_bag_bwd_f32: null pointer");
  ALVQ_REQUIRE(B > 0 && L > 0 && K > 0 && M > 0, ALVQ_EINVAL, "alvq_embedding_bag_bwd_f32: bad dims");
  ALVQ_REQUIRE((long)B * L * 4 <= 64 * 1024, ALVQ_EUNSUPPORTED, "alvq_embedding_bag_bwd_f32: B*L = %ld indices exceed the 64 KB LDS table",
               (long)B * L);
  hipLaunchKernelGGL(embedding_bag_bwd_kernel, dim3((M + 3) / 4), dim3(256), (size_t)B * L * 4, (hipStream_t)stream, dz, idx, dW, dbias,
                     B, L, K, M, accumulate_bias, bad_index);
  return check_launch("alvq_embedding_bag_bwd_f32");
}
