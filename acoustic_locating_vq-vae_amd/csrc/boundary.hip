// Input boundary for sources whose CHANNEL axis is already the contiguous one (round 4; SURVEY 8(b) "element strides of x").
//
// scripts/train_rir.py:42-49 standardises the RIR spectrogram (B, F, T) over dim 1 and hands the model
// `x.permute(0, 2, 1)`: time frames are channels (C = T), frequency is the convolution axis (L = F).  The NLC compute
// layout is act[row(b, l)][c] with channels contiguous -- which is exactly the memory order of the UNPERMUTED tensor:
// act[row(b, f)][t] = x[b][f][t].  Round 3 nevertheless materialised the permute (alvq_transpose_f32) and transposed back
// (alvq_ncl_to_nlc_*): standardise -> transpose -> convert, three HBM passes over the batch on a 1.8 ms step.  Here:
//
//   alvq_rows_to_nlc(x (B, L, C) fp32 contiguous, fmt, standardise, take_abs)
//       y[row(b, l)][c] = convert(s(x)[b][l][c]),   s = identity, or the per-(b, c) standardisation over l
//       ((v - mean_l) / (std_l + 1e-8), unbiased std: train_rir.py:43-44) in the arithmetic -- summation order included -- of
//       alvq_standardise_f32, so the fused result is bit-identical to standardise -> transpose -> convert.
//
// One workgroup owns (sample b, 64 channels): with `standardise` it stages the L x 64 tile in LDS once (L <= 240: 61 KB),
// reduces along l with lanes along c (coalesced 256-byte rows), and converts out of LDS; every thread then writes 8
// consecutive channels of a row (16-byte stores per plane).  Gap rows, row 0, the rows past the batch and channels >= C are
// written as zeros, like the transposing conversions do.  HBM-bound: reads x once, writes the planes once.
#include "alvq_common.h"
#include "bf16_common.h"
#include "f16mx_common.h"

namespace alvq {

constexpr int RN_FMT_BF16 = 1, RN_FMT_BF16X3 = 2, RN_FMT_F16MX = 3;
constexpr int RN_MAX_L_STD = 240;

// eight consecutive channels of one row, in the activation-class form of format FMT
template <int FMT>
__device__ __forceinline__ void rn_store8(u16* y, long plane, long row, int Cp, int c, const float (&v)[8], int& range) {
  const long o = row * Cp + c;
  if (FMT == RN_FMT_F16MX) {
    unsigned h[4], qh[2], ql[2];
#pragma unroll
    for (int e = 0; e < 8; ++e) range |= (fabsf(v[e]) >= 65504.f ? 1 : 0) | (v[e] != v[e] ? 2 : 0);
    fx_split<8>(v, fx_pow2(FX_E_ACT), fx_pow2(FX_E_ACT - FX_LO_SHIFT), h, qh, ql);
    *(u32x4*)(y + o) = u32x4{h[0], h[1], h[2], h[3]};
    unsigned char* q = (unsigned char*)(y + plane) + row * Cp * 2 + fx_q_off(c);
    *(u32x2*)q = u32x2{qh[0], qh[1]};
    *(u32x2*)(q + 32) = u32x2{ql[0], ql[1]};
    return;
  }
  u32x4 hi;
#pragma unroll
  for (int e = 0; e < 4; ++e) hi[e] = f2bf_pk(v[2 * e], v[2 * e + 1]);
  *(u32x4*)(y + o) = hi;
  if (FMT == RN_FMT_BF16X3) {
    u32x4 lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float h0 = __uint_as_float(hi[e] << 16), h1 = __uint_as_float(hi[e] & 0xffff0000u);
      lo[e] = f2bf_pk(v[2 * e] - h0, v[2 * e + 1] - h1);
    }
    *(u32x4*)(y + plane + o) = lo;
  }
}

template <int FMT, int STD>
__global__ __launch_bounds__(256) void rows_to_nlc_kernel(const float* x, u16* y, long plane, int B, int C, int L, int Cp,
                                                          int rows_total, int take_abs, int* range_flag) {
  extern __shared__ float lds[];
  if (FMT == RN_FMT_F16MX) fx_saturating_conversions();
  const int ctiles = Cp / 64, tid = threadIdx.x;
  const int rr = tid >> 3, cg = (tid & 7) * 8;
  const float zero8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int range = 0;
  if ((int)blockIdx.x >= B * ctiles) {     // tail workgroups: row 0 and the rows behind the last sample, one 64-channel column each
    const int c0 = ((int)blockIdx.x - B * ctiles) * 64;
    const int first_tail = 1 + B * (L + 1);
    for (int r = rr; r < 1 + rows_total - first_tail; r += 32)
      rn_store8<FMT>(y, plane, r == 0 ? 0 : first_tail + r - 1, Cp, c0 + cg, zero8, range);
    return;
  }
  const int b = blockIdx.x / ctiles, c0 = (blockIdx.x % ctiles) * 64;
  const float* xb = x + (long)b * L * C;
  float* tile = lds;                        // STD: [L][64]
  float* mean = lds + (STD ? L * 64 : 0);   // [64]
  float* den = mean + 64;                   // [64]
  float* red = den + 64;                    // [4][64]
  if (STD) {
    const int ct = tid & 63, lg = tid >> 6;
    const bool cok = c0 + ct < C;
    float s = 0.f;
    for (int l = lg; l < L; l += 4) {       // the order of standardise_regs_kernel: four interleaved partial sums
      float v = cok ? xb[(long)l * C + c0 + ct] : 0.f;
      v = take_abs ? fabsf(v) : v;
      tile[l * 64 + ct] = v;
      s += v;
    }
    red[lg * 64 + ct] = s;
    __syncthreads();
    const float m = ((red[ct] + red[64 + ct]) + (red[128 + ct] + red[192 + ct])) / (float)L;
    __syncthreads();
    float q = 0.f;
    for (int l = lg; l < L; l += 4) {
      const float d = tile[l * 64 + ct] - m;
      q += d * d;
    }
    red[lg * 64 + ct] = q;
    __syncthreads();
    if (lg == 0) {
      const float var = ((red[ct] + red[64 + ct]) + (red[128 + ct] + red[192 + ct])) / (float)(L - 1);
      mean[ct] = m;
      den[ct] = sqrtf(var) + 1e-8f;
    }
    __syncthreads();
  }
  for (int l = rr; l <= L; l += 32) {       // l == L: the shared zero row behind the sample
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = c0 + cg + e;
      float t = 0.f;
      if (l < L && c < C) {
        if (STD) t = (tile[l * 64 + cg + e] - mean[cg + e]) / den[cg + e];
        else {
          t = xb[(long)l * C + c];
          t = take_abs ? fabsf(t) : t;
        }
      }
      v[e] = t;
    }
    rn_store8<FMT>(y, plane, 1 + (long)b * (L + 1) + l, Cp, c0 + cg, v, range);
  }
  if (FMT == RN_FMT_F16MX && __any(range)) {
    for (int o = 32; o > 0; o >>= 1) range |= __shfl_xor(range, o, 64);
    if ((tid & 63) == 0 && range_flag) atomicOr(range_flag, range);
  }
}

template <int FMT>
static int rows_to_nlc_launch(const float* x, void* y, int B, int C, int L, int standardise, int take_abs, hipStream_t s) {
  const int Cp = (C + 63) / 64 * 64, rows = (int)alvq_nlc_rows(B, L);
  const long plane = ((long)rows + 2L * alvq_nlc_guard_rows()) * Cp;
  const int blocks = (B + 1) * (Cp / 64);
  int* flag = FMT == RN_FMT_F16MX ? fx_range_flag_ptr() : nullptr;
  if (standardise)
    hipLaunchKernelGGL((rows_to_nlc_kernel<FMT, 1>), dim3(blocks), dim3(256), (size_t)(L * 64 + 6 * 64) * sizeof(float), s, x, (u16*)y,
                       plane, B, C, L, Cp, rows, take_abs, flag);
  else
    hipLaunchKernelGGL((rows_to_nlc_kernel<FMT, 0>), dim3(blocks), dim3(256), (size_t)(6 * 64) * sizeof(float), s, x, (u16*)y, plane, B,
                       C, L, Cp, rows, take_abs, flag);
  return check_launch("alvq_rows_to_nlc");
}

}  // namespace alvq

using namespace alvq;

extern "C" int alvq_rows_to_nlc_max_std_rows(void) { return RN_MAX_L_STD; }

extern "C" int alvq_rows_to_nlc(const float* x, void* y, int B, int C, int L, int fmt, int standardise, int take_abs, void* stream) {
  ALVQ_REQUIRE(x && y, ALVQ_EINVAL, "alvq_rows_to_nlc: null pointer");
  ALVQ_REQUIRE(B > 0 && C > 0 && L > 0, ALVQ_EINVAL, "alvq_rows_to_nlc: bad dims");
  ALVQ_REQUIRE(fmt >= RN_FMT_BF16 && fmt <= RN_FMT_F16MX, ALVQ_EINVAL, "alvq_rows_to_nlc: fmt=%d (1 bf16, 2 bf16x3, 3 f16mx)", fmt);
  ALVQ_REQUIRE(!standardise || (L >= 2 && L <= RN_MAX_L_STD), ALVQ_EUNSUPPORTED,
               "alvq_rows_to_nlc: the fused standardisation holds an L x 64 tile in LDS, L = %d is outside 2..%d", L, RN_MAX_L_STD);
  hipStream_t s = (hipStream_t)stream;
  if (fmt == RN_FMT_BF16) return rows_to_nlc_launch<RN_FMT_BF16>(x, y, B, C, L, standardise, take_abs, s);
  if (fmt == RN_FMT_BF16X3) return rows_to_nlc_launch<RN_FMT_BF16X3>(x, y, B, C, L, standardise, take_abs, s);
  return rows_to_nlc_launch<RN_FMT_F16MX>(x, y, B, C, L, standardise, take_abs, s);
}
