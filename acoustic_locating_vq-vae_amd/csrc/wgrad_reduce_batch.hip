// Deferred, batched split reduction of the weight gradients of a train step (bf16 / fp16 kernel family): the launches of a
// backward pass leave their split partials (and bias partials) in DISTINCT caller-owned scratch regions
// (alvq_conv1d_wgrad_*(..., accumulate = ALVQ_WGRAD_DEFER)), and ONE launch at the end sums every one of them into its
// gradient.  Same sums in the same order as the per-launch reductions (bitwise identical results): what changes is 16
// small bandwidth-bound launches per step becoming one that fills the chip.  Descriptors travel by value in the kernel
// argument (graph-capturable).
#include "alvq_common.h"
#include "wgrad_reduce.h"

namespace alvq {

constexpr int RB_MAX = 40;       // descriptors per launch
struct ReduceBatch {
  alvq_reduce_desc d[RB_MAX];
  int blk0[RB_MAX];              // first block of each descriptor
  int nblk[RB_MAX];
  int n;
};

__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(ReduceBatch b) {
  __shared__ float tile[3][16][65];
  int di = 0;
  for (int i = 1; i < b.n; ++i)
    if ((int)blockIdx.x >= b.blk0[i]) di = i;
  const alvq_reduce_desc& d = b.d[di];
  const int lb = blockIdx.x - b.blk0[di];
  const int acc = d.accumulate;
  if (d.w_layout == ALVQ_W_IOK && d.stride == (int64_t)d.KW * d.M * d.C)
    wgrad_reduce_iok_body<16>(tile, d.partial, d.dst, d.splits, d.KW, d.M, d.C, acc, d.scale, lb);
  else
    wgrad_reduce_body(d.partial, d.dst, d.splits, d.KW, d.M, d.C, d.w_layout, acc, d.scale, (long)d.stride, lb, b.nblk[di]);
}

}  // namespace alvq

using namespace alvq;

extern "C" int alvq_wgrad_reduce_batch(const alvq_reduce_desc* descs, int n, void* stream) {
  ALVQ_REQUIRE(descs && n > 0, ALVQ_EINVAL, "alvq_wgrad_reduce_batch: no descriptors");
  for (int i = 0; i < n; ++i) {
    const alvq_reduce_desc& s = descs[i];
    ALVQ_REQUIRE(s.partial && s.dst, ALVQ_EINVAL, "alvq_wgrad_reduce_batch: null pointer in descriptor %d", i);
    ALVQ_REQUIRE(s.splits >= 1 && s.splits <= 64 && s.M > 0 && s.C > 0 && (s.KW == 1 || s.KW == 3), ALVQ_EINVAL,
                 "alvq_wgrad_reduce_batch: bad dims in descriptor %d", i);
    ALVQ_REQUIRE(s.w_layout == ALVQ_W_OIK || s.w_layout == ALVQ_W_IOK, ALVQ_EINVAL, "alvq_wgrad_reduce_batch: w_layout in descriptor %d", i);
    ALVQ_REQUIRE(s.stride >= (int64_t)s.KW * s.M * s.C, ALVQ_EINVAL, "alvq_wgrad_reduce_batch: stride in descriptor %d", i);
  }
  for (int i0 = 0; i0 < n; i0 += RB_MAX) {
    ReduceBatch b{};
    b.n = n - i0 < RB_MAX ? n - i0 : RB_MAX;
    int blocks = 0;
    for (int i = 0; i < b.n; ++i) {
      const alvq_reduce_desc& s = descs[i0 + i];
      b.d[i] = s;
      const long total = (long)s.KW * s.M * s.C;
      int nb;
      if (s.w_layout == ALVQ_W_IOK && s.stride == total) nb = ((s.M + 15) / 16) * ((s.C + 63) / 64);
      else {
        nb = (int)((total + 255) / 256);
        if (nb > 2048) nb = 2048;
      }
      b.blk0[i] = blocks;
      b.nblk[i] = nb;
      blocks += nb;
    }
    hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, b);
    int rc = check_launch("alvq_wgrad_reduce_batch");
    if (rc) return rc;
  }
  return ALVQ_OK;
}
