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

// Eight consecutive channels (c .. c+7 of one packed row; `row_off` = element offset of the row in an image, `plane` =
// element offset of the second image) in the format PLANES selects: 1 bf16; 2 bf16 hi + lo; 3 f16mx H + Q.
template <int PLANES>
__device__ __forceinline__ void pack_store8(u16* wp, long plane, long row_off, int c, const float (&v)[8]) {
  const long o = row_off + c;
  if (PLANES == 3) {   // f16mx: H image (fp16) + Q image ([hi8 x 32 | lo8 x 32] per 32 channels), weight-class scale
    unsigned h[4], qh[2], ql[2];
    fx_split<8>(v, fx_pow2(FX_E_W), fx_pow2(FX_E_W - FX_LO_SHIFT), h, qh, ql);
    *(u32x4*)(wp + o) = u32x4{h[0], h[1], h[2], h[3]};
    unsigned char* q = (unsigned char*)(wp + plane) + row_off * 2 + fx_q_off(c);
    *(u32x2*)q = u32x2{qh[0], qh[1]};
    *(u32x2*)(q + 32) = u32x2{ql[0], ql[1]};
    return;
  }
  u32x4 hi;
#pragma unroll
  for (int e = 0; e < 4; ++e) hi[e] = f2bf_pk(v[2 * e], v[2 * e + 1]);
  *(u32x4*)(wp + o) = hi;
  if (PLANES == 2) {
    u32x4 lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float h0 = __uint_as_float(hi[e] << 16), h1 = __uint_as_float(hi[e] & 0xffff0000u);
      lo[e] = f2bf_pk(v[2 * e] - h0, v[2 * e + 1] - h1);
    }
    *(u32x4*)(wp + plane + o) = lo;
  }
}

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
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = tile[t][mr][cg + e];
    pack_store8<PLANES>(d.wp, b.plane[di], ((long)t * d.Mp + m0 + mr) * d.Cp, c0 + cg, v);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Adam + packing in one pass (round 3; round-2 verdict item 4): the optimiser has every new weight in registers, so it
// emits the packed image(s) the next step's convolutions read instead of a second kernel re-reading what it just wrote.
// A workgroup owns a 32 (dim0) x 64 (dim1) x KW tile of one conv weight in its NATIVE layout: it reads w, g, m, v (lanes
// along the contiguous (dim1, tap) axis), applies torch.optim.Adam's arithmetic -- the expression of adam_dev_kernel,
// bit for bit -- writes w, m, v back, keeps the new w tile in LDS and writes it out as the image read OIK
// (M = dim0, C = dim1) and / or the image read IOK (C = dim0, M = dim1, taps flipped), 16-byte runs along c.  Image
// padding outside the weight's 32 x 64-rounded extent is not touched (it holds the zeros of the image's first full pack).
constexpr int AP_MAX = 24;
struct AdamPackDesc {
  float* w; const float* g; float* m; float* v;
  u16* oik; u16* iok;               // packed images or null
  int dim0, dim1, KW, blk0, tiles1;
  int oik_Mp, oik_Cp, iok_Mp, iok_Cp;
};
struct AdamPackBatch {
  AdamPackDesc d[AP_MAX];
  int n;
  const float* sc;                  // {lr / bias_correction1, sqrt(bias_correction2), grad_scale} (alvq_adam_advance_f32)
  const float* skip;                // nullable; non-zero: the step saturated an fp16-range format -- nothing is updated
  float beta1, beta2, eps;
};

template <int PLANES>
__global__ __launch_bounds__(256) void adam_pack_batch_kernel(AdamPackBatch b) {
  __shared__ float tile[3][32][65];
  if (b.skip && *b.skip != 0.f) return;      // skipped step: weights, moments and packed images stay as they are
  if (PLANES == 3) fx_saturating_conversions();
  int di = 0;
  for (int i = 1; i < b.n; ++i)
    if ((int)blockIdx.x >= b.d[i].blk0) di = i;
  const AdamPackDesc& d = b.d[di];
  const int lb = blockIdx.x - d.blk0;
  const int r0 = (lb / d.tiles1) * 32, q0 = (lb % d.tiles1) * 64;       // tile origin along dim0 / dim1
  const int KW = d.KW, tid = threadIdx.x;
  const float lr_bc1 = b.sc[0], bc2_sqrt = b.sc[1], gscale = b.sc[2];
  const float beta1 = b.beta1, beta2 = b.beta2, eps = b.eps;
  const int run = 64 * KW;
  // a row segment of the tile is 64 * KW contiguous floats: 16-byte loads / stores when every segment starts on a 16-byte
  // boundary and the tile is full along dim1 (round 4: the scalar form ran at 3.8 TB/s); the arithmetic per element is the same
  if (q0 + 64 <= d.dim1 && ((d.dim1 * KW) & 3) == 0) {
#pragma unroll 2
    for (int e4 = tid; e4 < 8 * run; e4 += 256) {
      const int e = 4 * e4, rr = e / run, j = e - rr * run;
      f32x4 wn = {0.f, 0.f, 0.f, 0.f};
      if (r0 + rr < d.dim0) {
        const long i = ((long)(r0 + rr) * d.dim1 + q0) * KW + j;
        const f32x4 g4 = *(const f32x4*)(d.g + i), m4 = *(const f32x4*)(d.m + i), v4 = *(const f32x4*)(d.v + i), w4 = *(const f32x4*)(d.w + i);
        f32x4 mo, vo;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float gr = g4[q] * gscale;
          const float mm = m4[q] + (gr - m4[q]) * (1.f - beta1);
          const float vv = v4[q] * beta2 + (1.f - beta2) * gr * gr;
          const float denom = sqrtf(vv) / bc2_sqrt + eps;
          mo[q] = mm;
          vo[q] = vv;
          wn[q] = w4[q] - lr_bc1 * (mm / denom);
        }
        *(f32x4*)(d.m + i) = mo;
        *(f32x4*)(d.v + i) = vo;
        *(f32x4*)(d.w + i) = wn;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int jj = j + q, qq = jj / KW, t = jj - qq * KW;
        tile[t][rr][qq] = wn[q];
      }
    }
  } else {
#pragma unroll 2
    for (int e = tid; e < 32 * run; e += 256) {
      const int rr = e / run, j = e - rr * run, qq = j / KW, t = j - qq * KW;
      float wn = 0.f;
      if (r0 + rr < d.dim0 && q0 + qq < d.dim1) {
        const long i = ((long)(r0 + rr) * d.dim1 + q0) * KW + j;
        const float gr = d.g[i] * gscale;
        const float m0 = d.m[i], v0 = d.v[i];
        const float mm = m0 + (gr - m0) * (1.f - beta1);
        const float vv = v0 * beta2 + (1.f - beta2) * gr * gr;
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        d.m[i] = mm;
        d.v[i] = vv;
        wn = d.w[i] - lr_bc1 * (mm / denom);
        d.w[i] = wn;
      }
      tile[t][rr][qq] = wn;
    }
  }
  __syncthreads();
  if (d.oik) {                     // image[t][m = dim0][c = dim1]
    const int mr = tid >> 3, cg = (tid & 7) * 8;
    const long plane = (long)KW * d.oik_Mp * d.oik_Cp;
    for (int t = 0; t < KW; ++t) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = tile[t][mr][cg + e];
      pack_store8<PLANES>(d.oik, plane, ((long)t * d.oik_Mp + r0 + mr) * d.oik_Cp, q0 + cg, v);
    }
  }
  if (d.iok) {                     // image[KW-1-k][m = dim1][c = dim0]
    const int mr = tid >> 2, cg = (tid & 3) * 8;
    const long plane = (long)KW * d.iok_Mp * d.iok_Cp;
    for (int t = 0; t < KW; ++t) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = tile[KW - 1 - t][cg + e][mr];
      pack_store8<PLANES>(d.iok, plane, ((long)t * d.iok_Mp + q0 + mr) * d.iok_Cp, r0 + cg, v);
    }
  }
}

// Adam over the segments [lo, hi) of a flat buffer (the parameters the fused kernel above does not own: biases, the
// codebook): same arithmetic, one launch; a block owns 1024 consecutive elements of one segment.
constexpr int AS_MAX = 48;
struct AdamSegs {
  long lo[AS_MAX], hi[AS_MAX];
  int blk0[AS_MAX];
  int n;
};
__global__ __launch_bounds__(256) void adam_segments_kernel(float* p, const float* g, float* m, float* v, AdamSegs s, const float* sc,
                                                            float beta1, float beta2, float eps, const float* skip) {
  if (skip && *skip != 0.f) return;
  int si = 0;
  for (int i = 1; i < s.n; ++i)
    if ((int)blockIdx.x >= s.blk0[i]) si = i;
  const float lr_bc1 = sc[0], bc2_sqrt = sc[1], gscale = sc[2];
  const long base = s.lo[si] + (long)(blockIdx.x - s.blk0[si]) * 1024;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long e = base + k * 256 + threadIdx.x;
    if (e >= s.hi[si]) return;
    const float gr = g[e] * gscale;
    const float mm = m[e] + (gr - m[e]) * (1.f - beta1);
    const float vv = v[e] * beta2 + (1.f - beta2) * gr * gr;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    m[e] = mm;
    v[e] = vv;
    p[e] = p[e] - lr_bc1 * (mm / denom);
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

extern "C" int alvq_adam_pack_batch(const alvq_adam_pack_desc* descs, int n, int planes, const float* scalars, float beta1,
                                    float beta2, float eps, const float* skip, void* stream) {
  ALVQ_REQUIRE(descs && n > 0 && scalars, ALVQ_EINVAL, "alvq_adam_pack_batch: no descriptors / scalars");
  ALVQ_REQUIRE(planes >= 1 && planes <= 3, ALVQ_EINVAL, "alvq_adam_pack_batch: planes=%d (1, 2 or 3)", planes);
  for (int i = 0; i < n; ++i) {
    const alvq_adam_pack_desc& s = descs[i];
    ALVQ_REQUIRE(s.w && s.g && s.m && s.v, ALVQ_EINVAL, "alvq_adam_pack_batch: null pointer in descriptor %d", i);
    ALVQ_REQUIRE(s.dim0 > 0 && s.dim1 > 0 && (s.KW == 1 || s.KW == 3), ALVQ_EINVAL, "alvq_adam_pack_batch: bad dims in descriptor %d", i);
  }
  for (int i0 = 0; i0 < n; i0 += AP_MAX) {
    AdamPackBatch b{};
    b.n = n - i0 < AP_MAX ? n - i0 : AP_MAX;
    b.sc = scalars; b.skip = skip; b.beta1 = beta1; b.beta2 = beta2; b.eps = eps;
    int blocks = 0;
    for (int i = 0; i < b.n; ++i) {
      const alvq_adam_pack_desc& s = descs[i0 + i];
      AdamPackDesc& d = b.d[i];
      d.w = s.w; d.g = s.g; d.m = s.m; d.v = s.v; d.oik = (u16*)s.wp_oik; d.iok = (u16*)s.wp_iok;
      d.dim0 = s.dim0; d.dim1 = s.dim1; d.KW = s.KW;
      d.oik_Mp = pad_to(s.dim0, WP_ROWS); d.oik_Cp = pad_to(s.dim1, TB_K);
      d.iok_Mp = pad_to(s.dim1, WP_ROWS); d.iok_Cp = pad_to(s.dim0, TB_K);
      d.blk0 = blocks; d.tiles1 = (s.dim1 + 63) / 64;
      blocks += ((s.dim0 + 31) / 32) * d.tiles1;
    }
    if (planes == 1) hipLaunchKernelGGL(adam_pack_batch_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, b);
    else if (planes == 2) hipLaunchKernelGGL(adam_pack_batch_kernel<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, b);
    else hipLaunchKernelGGL(adam_pack_batch_kernel<3>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, b);
    int rc = check_launch("alvq_adam_pack_batch");
    if (rc) return rc;
  }
  return ALVQ_OK;
}

extern "C" int alvq_adam_segments_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const int64_t* lo,
                                      const int64_t* hi, int nseg, const float* scalars, float beta1, float beta2, float eps,
                                      const float* skip, void* stream) {
  ALVQ_REQUIRE(param && grad && exp_avg && exp_avg_sq && scalars && lo && hi, ALVQ_EINVAL, "alvq_adam_segments_f32: null pointer");
  ALVQ_REQUIRE(nseg > 0, ALVQ_EINVAL, "alvq_adam_segments_f32: no segments");
  for (int i0 = 0; i0 < nseg; i0 += AS_MAX) {
    AdamSegs s{};
    s.n = nseg - i0 < AS_MAX ? nseg - i0 : AS_MAX;
    int blocks = 0;
    for (int i = 0; i < s.n; ++i) {
      ALVQ_REQUIRE(lo[i0 + i] >= 0 && hi[i0 + i] > lo[i0 + i], ALVQ_EINVAL, "alvq_adam_segments_f32: bad segment %d", i0 + i);
      s.lo[i] = lo[i0 + i]; s.hi[i] = hi[i0 + i];
      s.blk0[i] = blocks;
      blocks += (int)((hi[i0 + i] - lo[i0 + i] + 1023) / 1024);
    }
    hipLaunchKernelGGL(adam_segments_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, s, scalars,
                       beta1, beta2, eps, skip);
    int rc = check_launch("alvq_adam_segments_f32");
    if (rc) return rc;
  }
  return ALVQ_OK;
}
