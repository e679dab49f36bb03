// Batched weight packing: every fp32 conv weight a train step needs, in every layout it is needed, is re-packed
// into its K-contiguous bf16 image (wp[tap][Mp][Cp], see alvq_pack_weight_bf16) by ONE launch.  The descriptors
// travel by value in the kernel argument, so the launch is graph-capturable.  A workgroup owns a 32 (m) x 64 (c)
// tile of one weight for all taps: it reads the source with lanes along the source's contiguous axis (OIK: (c, tap)
// runs of one m; IOK: (m, tap) runs of one c), turns the tile through LDS and writes 16-byte runs along c.
#include "alvq_common.h"
#include "bf16_common.h"
#include "f16mx_common.h"

namespace alvq {

constexpr int PB_MAX = 32;       // descriptors per launch
struct PackDesc {
  const float* w;
  u16* wp;
  int M, C, KW, layout, Mp, Cp, blk0, ctiles;
};
struct PackBatch {
  PackDesc d[PB_MAX];
  int n;
  long plane[PB_MAX];   // PLANES >= 2: element offset of the second image (bf16x3: lo; f16mx: Q)
};

template <int PLANES>
__global__ __launch_bounds__(256) void pack_weights_batch_kernel(PackBatch b) {
  __shared__ float tile[3][32][65];
  if (PLANES == 3) fx_saturating_conversions();
  int di = 0;
  for (int i = 1; i < b.n; ++i)
    if ((int)blockIdx.x >= b.d[i].blk0) di = i;
  const PackDesc& d = b.d[di];
  const int lb = blockIdx.x - d.blk0;
  const int m0 = (lb / d.ctiles) * 32, c0 = (lb % d.ctiles) * 64;
  const int KW = d.KW, tid = threadIdx.x;
  if (d.layout == ALVQ_W_OIK) {
    const int run = 64 * KW;
#pragma unroll 4      // 8 KW iterations: four loads in flight instead of one round trip each
    for (int e = tid; e < 32 * run; e += 256) {
      const int mr = e / run, j = e - mr * run, c = j / KW, t = j - c * KW;
      const int m = m0 + mr;
      tile[t][mr][c] = (m < d.M && c0 + c < d.C) ? d.w[((long)m * d.C + c0) * KW + j] : 0.f;
    }
  } else {
    const int run = 32 * KW;
#pragma unroll 4
    for (int e = tid; e < 64 * run; e += 256) {
      const int cc = e / run, j = e - cc * run, mr = j / KW, k = j - mr * KW;
      const int c = c0 + cc;
      tile[KW - 1 - k][mr][cc] = (c < d.C && m0 + mr < d.M) ? d.w[((long)c * d.M + m0) * KW + j] : 0.f;
    }
  }
  __syncthreads();
  const int mr = tid >> 3, cg = (tid & 7) * 8;
  for (int t = 0; t < KW; ++t) {
    const long o = ((long)t * d.Mp + m0 + mr) * d.Cp + c0 + cg;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = tile[t][mr][cg + e];
    if (PLANES == 3) {   // f16mx: H image (fp16) + Q image ([hi8 x 32 | lo8 x 32] per 32 channels), weight-class scale
      unsigned h[4], qh[2], ql[2];
      fx_split<8>(v, fx_pow2(FX_E_W), fx_pow2(FX_E_W - FX_LO_SHIFT), h, qh, ql);
      *(u32x4*)(d.wp + o) = u32x4{h[0], h[1], h[2], h[3]};
      unsigned char* q = (unsigned char*)(d.wp + b.plane[di]) + ((long)t * d.Mp + m0 + mr) * d.Cp * 2 + fx_q_off(c0 + cg);
      *(u32x2*)q = u32x2{qh[0], qh[1]};
      *(u32x2*)(q + 32) = u32x2{ql[0], ql[1]};
      continue;
    }
    u32x4 hi;
#pragma unroll
    for (int e = 0; e < 4; ++e) hi[e] = f2bf_pk(v[2 * e], v[2 * e + 1]);
    *(u32x4*)(d.wp + o) = hi;
    if (PLANES == 2) {
      u32x4 lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float h0 = __uint_as_float(hi[e] << 16), h1 = __uint_as_float(hi[e] & 0xffff0000u);
        lo[e] = f2bf_pk(v[2 * e] - h0, v[2 * e + 1] - h1);
      }
      *(u32x4*)(d.wp + b.plane[di] + o) = lo;
    }
  }
}

}  // namespace alvq

using namespace alvq;

static inline int pad_to(int x, int q) { return (x + q - 1) / q * q; }

extern "C" int alvq_pack_weights_bf16_batch(const alvq_pack_desc* descs, int n, int planes, void* stream) {
  ALVQ_REQUIRE(descs && n > 0, ALVQ_EINVAL, "alvq_pack_weights_bf16_batch: no descriptors");
  ALVQ_REQUIRE(planes >= 1 && planes <= 3, ALVQ_EINVAL, "alvq_pack_weights_bf16_batch: planes=%d (1, 2 or 3)", planes);
  for (int i = 0; i < n; ++i) {
    const alvq_pack_desc& s = descs[i];
    ALVQ_REQUIRE(s.w && s.wp, ALVQ_EINVAL, "alvq_pack_weights_bf16_batch: null pointer in descriptor %d", i);
    ALVQ_REQUIRE(s.M > 0 && s.C > 0 && (s.KW == 1 || s.KW == 3), ALVQ_EINVAL, "alvq_pack_weights_bf16_batch: bad dims in descriptor %d", i);
    ALVQ_REQUIRE(s.w_layout == ALVQ_W_OIK || s.w_layout == ALVQ_W_IOK, ALVQ_EINVAL, "alvq_pack_weights_bf16_batch: w_layout in descriptor %d", i);
  }
  for (int i0 = 0; i0 < n; i0 += PB_MAX) {
    PackBatch b{};
    b.n = n - i0 < PB_MAX ? n - i0 : PB_MAX;
    int blocks = 0;
    for (int i = 0; i < b.n; ++i) {
      const alvq_pack_desc& s = descs[i0 + i];
      PackDesc& d = b.d[i];
      d.w = s.w; d.wp = (u16*)s.wp; d.M = s.M; d.C = s.C; d.KW = s.KW; d.layout = s.w_layout;
      d.Mp = pad_to(s.M, WP_ROWS); d.Cp = pad_to(s.C, TB_K);
      d.blk0 = blocks; d.ctiles = d.Cp / 64;
      b.plane[i] = (long)s.KW * d.Mp * d.Cp;
      blocks += (d.Mp / 32) * d.ctiles;
    }
    if (planes == 1) hipLaunchKernelGGL(pack_weights_batch_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, b);
    else if (planes == 2) hipLaunchKernelGGL(pack_weights_batch_kernel<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, b);
    else hipLaunchKernelGGL(pack_weights_batch_kernel<3>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, b);
    int rc = check_launch("alvq_pack_weights_bf16_batch");
    if (rc) return rc;
  }
  return ALVQ_OK;
}
