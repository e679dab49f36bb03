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
  // element type of every 16-bit operand and of the NLC output: 0 = bf16 (the throughput mode), 1 = fp16 (the backward
  // pass of the f16mx_hb mode: gradients under a loss scale, the H planes of f16mx activations and packed weights)
  int elem;
  const float* out_scale;   // OUT == 1: device scalar multiplied into the fp32 output (undoes a loss scale), or null
  int* range_flag;          // fp16 outputs: sticky device flag, |= 4 when a stored value reached fp16's limit (or is a NaN)
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

// ---- element type of the 16-bit pipeline (template parameter F16: 0 = bf16, 1 = fp16).  Same geometry, same MFMA shapes
// and rates; what differs is the MFMA opcode and the conversions at the epilogues.
typedef _Float16 h16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2_t __attribute__((ext_vector_type(2)));
template <int F16>
__device__ __forceinline__ float elem2f(u16 v) {
  if (F16) return (float)__builtin_bit_cast(_Float16, v);
  return bf2f(v);
}
// two floats -> packed pair, round-to-nearest-even (fp16: saturating at +-65504 under elem_saturate())
template <int F16>
__device__ __forceinline__ unsigned elem_pk(float lo, float hi) {
  if (F16) {
    const f32x2 f = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, h16x2_t));
  }
  return f2bf_pk(lo, hi);
}
// fp16 conversions saturate instead of producing infinities: MODE.FP16_OVFL (see f16mx_common.h); once per wave, before
// the first conversion
template <int F16>
__device__ __forceinline__ void elem_saturate() {
  if (F16) __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);
}
template <int F16>
__device__ __forceinline__ f32x4 elem_mfma16(const bf16x8_t& a, const bf16x8_t& b, const f32x4& c) {
  if (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8_t, a), __builtin_bit_cast(h16x8_t, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// Saturation watch of fp16 outputs: `run` keeps the packed maximum of |half| over the words a lane stores (two VALU
// instructions per word); fp16_limit_reached() is true if some half is >= 0x7BFF, i.e. 65504 (the saturating conversion's
// ceiling) or a NaN.  One atomic per wave that saw one, at the end of its epilogue.
__device__ __forceinline__ void fp16_watch(unsigned& run, unsigned word) {
  const unsigned a = word & 0x7fff7fffu;
  asm("v_pk_max_u16 %0, %1, %2" : "=v"(run) : "v"(run), "v"(a));
}
__device__ __forceinline__ bool fp16_limit_reached(unsigned run) { return (run & 0xffffu) >= 0x7bffu || (run >> 16) >= 0x7bffu; }
__device__ __forceinline__ void fp16_report(unsigned run, int* flag) {
  if (flag && __any(fp16_limit_reached(run)) && (threadIdx.x & 63) == 0) atomicOr(flag, 4);
}

// defined in conv1d_bf16_v2.hip: the 256x256-tile kernel for wide layers
int conv1d_bf16_v2_launch(const ConvBArgs& a, int KW, hipStream_t stream);
// defined in conv1d_bf16_k3.hip: the same tile for width 3, one activation slab shared by the three taps
int conv1d_bf16_k3_launch(const ConvBArgs& a, hipStream_t stream);
// defined in conv1d_wgrad_bf16_v2.hip: ring-pipelined weight-gradient (+ its fixed-order split reduction)
int64_t conv1d_wgrad_bf16_v2_workspace_bytes(int total_rows, int C, int M, int KW);
int conv1d_wgrad_bf16_v2_splits(int total_rows, int C, int M, int KW, int nseg, bool with_bias);
// dbias (optional): the bias gradient, fused into the same launch; bias_partial: >= 64 * pad64(M) floats of scratch
// elem: 0 bf16, 1 fp16; out_scale (device scalar or null): multiplied into dw / dbias (undoes a loss scale)
int conv1d_wgrad_bf16_v2_launch(const void* const* dy, const void* const* x, int nseg, float* dw, void* workspace,
                                int total_rows, int C, int M, int KW, int w_layout, int accumulate, hipStream_t s,
                                float* dbias = nullptr, float* bias_partial = nullptr, int elem = 0,
                                const float* out_scale = nullptr);


}  // namespace alvq
