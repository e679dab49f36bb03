// Shared pieces of the bf16 path: bf16<->f32 helpers, LDS-DMA wrapper, the NLC-padded layout rules and the
// argument block of the convolution kernels.
#pragma once
#include "alvq_common.h"

namespace alvq {

typedef __bf16 bf16;
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

constexpr int TB_R = 128;      // rows (positions) per workgroup
constexpr int TB_M = 128;      // output channels per workgroup
constexpr int TB_K = 64;       // channels per chunk (128-byte LDS rows)
constexpr int XROWS = TB_R + 8;
constexpr int XBYTES = XROWS * 128;   // 17408
constexpr int WBYTES = TB_M * 128;    // 16384
constexpr int CS = TB_M + 4;          // fp32 C-tile row stride (floats)
constexpr int LDS_BYTES = 2 * XBYTES + 2 * WBYTES;  // 67584 == TB_R * CS * 4
static_assert(LDS_BYTES >= TB_R * CS * 4, "C tile must fit");
constexpr int GUARD_ROWS = 8;

__device__ __forceinline__ float bf2f(u16 v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ u16 f2bf(float f) {  // round-to-nearest-even; NaN stays NaN
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u16)((u >> 16) | 0x40);
  return (u16)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)gsrc,
                                   (void __attribute__((address_space(3)))*)lds_dst, 16, 0, 0);
}

// row validity in the padded matrix: data rows are 1 + b*(L+1) + l with l < L
__device__ __forceinline__ bool row_valid(int row, int Lp1, int nrows_data, int* b, int* l) {
  const int v = row - 1;
  const int bb = v / Lp1, ll = v - bb * Lp1;
  *b = bb;
  *l = ll;
  return row >= 1 && v < nrows_data && ll < Lp1 - 1;
}

struct ConvBArgs {
  const u16* x;     // [rows][Cp], points at row 0
  const u16* wp;    // [KW][Mp128][Cp]
  const float* bias;
  const u16* skip1;
  const u16* skip2;
  const u16* mask;
  const u16* post;
  u16* y;
  u16* y2;
  float* y_ncl;     // OUT==1: (B, M, L) fp32
  int B, L, Cp, M, Mop, Mp128;   // Mop: output row stride (M rounded to 64); Mp128: packed-weight rows per tap
  int relu;
  int rtiles, mtiles;
  // sign bits of activations, one byte per 8 consecutive channels of a row ([rows][Mop/8]): a ReLU'd output can leave
  // its mask behind (bits_out), and a later data-gradient launch reads 1/16 of the bytes instead of the tensor (mask_bits)
  const unsigned char* mask_bits;
  unsigned char* bits_out;
};

constexpr int WP_ROWS = 256;   // packed weights are padded to this many rows per tap (largest m-tile)
constexpr int NLC_ROW_PAD = 256;   // activation matrices are padded to this many rows (largest row tile)

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// two floats -> packed bf16 pair, round-to-nearest-even, by the gfx950 instruction v_cvt_pk_bf16_f32 (same bits as
// f2bf for every finite input)
__device__ __forceinline__ unsigned f2bf_pk(float lo, float hi) {
  const f32x2 f = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_t));
}

// Fused epilogue for 8 consecutive output channels of one row (both conv kernels): v = acc (+bias) (+skip1)
// (+skip2); relu; mask; y = bf16(v); y2 = bf16(v + post).  Gap / tail rows are written as zeros.
__device__ __forceinline__ void epilogue_store8(const ConvBArgs& a, float (&v)[8], const float (&bv)[8], bool ok, long o) {
  u32x4 out = {0u, 0u, 0u, 0u}, out2 = {0u, 0u, 0u, 0u};
  if (ok) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += bv[e];
    if (a.skip1) {
      const u16x8 s = *(const u16x8*)(a.skip1 + o);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += bf2f(s[e]);
    }
    if (a.skip2) {
      const u16x8 s = *(const u16x8*)(a.skip2 + o);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += bf2f(s[e]);
    }
    if (a.relu & 1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    if (a.mask_bits) {
      const unsigned bt = a.mask_bits[o >> 3];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = ((bt >> e) & 1u) ? v[e] : 0.f;
    } else if (a.mask) {
      const u16x8 s = *(const u16x8*)(a.mask + o);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = bf2f(s[e]) > 0.f ? v[e] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) out[e] = f2bf_pk(v[2 * e], v[2 * e + 1]);
    if (a.y2) {
      const u16x8 s = *(const u16x8*)(a.post + o);
#pragma unroll
      for (int e = 0; e < 4; ++e) out2[e] = f2bf_pk(v[2 * e] + bf2f(s[2 * e]), v[2 * e + 1] + bf2f(s[2 * e + 1]));
    }
  }
  *(u32x4*)(a.y + o) = out;
  if (a.y2) *(u32x4*)(a.y2 + o) = out2;
  if (a.bits_out) {   // bit e = (stored y[e] > 0); zero for gap / tail rows
    // a bf16 in the upper half of a word IS the fp32 pattern of its value: positive and non-zero <=> that word > 0 as
    // an integer.  t = max(word, 0) also disposes of -0; the sign of 0 - t is the bit, shifted in from the right by
    // v_alignbit_b32 -- four VALU instructions per element instead of the dozen of the mask-and-compare form.
    unsigned bt = 0;
#pragma unroll
    for (int e = 7; e >= 0; --e) {
      const int half = (int)((e & 1) ? (out[e >> 1] & 0xffff0000u) : (out[e >> 1] << 16));
      const int t = half > 0 ? half : 0;
      bt = __builtin_amdgcn_alignbit(bt, 0u - (unsigned)t, 31);
    }
    a.bits_out[o >> 3] = (unsigned char)bt;
  }
}

// defined in conv1d_bf16_v2.hip: the 256x256-tile kernel for wide layers
int conv1d_bf16_v2_launch(const ConvBArgs& a, int KW, hipStream_t stream);
// defined in conv1d_bf16_k3.hip: the same tile for width 3, one activation slab shared by the three taps
int conv1d_bf16_k3_launch(const ConvBArgs& a, hipStream_t stream);
// defined in conv1d_wgrad_bf16_v2.hip: ring-pipelined weight-gradient (+ its fixed-order split reduction)
int64_t conv1d_wgrad_bf16_v2_workspace_bytes(int total_rows, int C, int M, int KW);
// dbias (optional): the bias gradient, fused into the same launch; bias_partial: >= 64 * pad64(M) floats of scratch
int conv1d_wgrad_bf16_v2_launch(const void* const* dy, const void* const* x, int nseg, float* dw, void* workspace,
                                int total_rows, int C, int M, int KW, int w_layout, int accumulate, hipStream_t s,
                                float* dbias = nullptr, float* bias_partial = nullptr);


}  // namespace alvq
