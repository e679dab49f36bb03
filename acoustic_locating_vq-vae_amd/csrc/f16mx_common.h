// "f16mx" split format: fp32-grade products at TWO matrix-pipe units per product instead of the three of the
// split-bf16 path (bf16x3).
//
// Every fp32 value v is carried as
//     H   = fp16(v)                                   (11-bit significand; |v - H| <= 2^-11 |v|)
//     hi8 = e4m3(v / S)            lo8 = e4m3((v - H) / (S * 2^-11))
// and a product a*b is evaluated as
//     Ha*Hb                         one fp16 MFMA  (v_mfma_f32_32x32x16_f16; fp16 x fp16 is exact in fp32)
//   + hi8_a*lo8_b + lo8_a*hi8_b     one block-scaled fp8 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, K = 64 = two blocks of
//                                   32: block 0 pairs A.hi8 with B.lo8, block 1 pairs A.lo8 with B.hi8; the E8M0 block
//                                   scales carry S and S*2^-11 -- the MX fp8 path runs at twice the fp16 rate per K).
// The dropped lo*lo term is 2^-22; the cross terms carry a value of relative size 2^-11 with fp8's 2^-4 precision on
// both factors, i.e. ~1.5e-5 rms per product -- the same class as bf16x3 (2e-6), two orders inside the 1e-3 bar.
// Per 32 channels of a 32x32 output block: 2 x 32 + 64 = 128 matrix-pipe cycles against 192 for bf16x3.
//
// Scales are per tensor CLASS, not per block: e4m3 spans 2^-9 .. 448 (15 binades at full precision), what the cross
// terms need is 4 bits of a quantity that is itself 2^-11 of the product, and an element outside the window degrades
// gracefully (hi8 saturated at 448*S or flushed below 2^-10*S: that one product keeps fp16-grade precision).  So
//     activations / scaled gradients:  S = 1        (exponent byte 127)      window 2^-10 .. 448
//     weights:                          S = 2^-8     (exponent byte 119)      window 2^-17 .. 1.75
// Gradients are brought into the window by a power-of-two loss scale chosen on the device from the amax of the
// gradient that enters a backward chain (exact in fp32; undone where the chain leaves the format).
//
// Storage (same bytes as fp32, same geometry as the two planes of bf16x3): plane 0 = H as [rows][Cp] fp16; plane 1 = Q as
// [rows][Cp/32][hi8 x 32 | lo8 x 32], i.e. 64 bytes per row per 32-channel chunk -- exactly a bf16 plane's geometry, so
// the LDS-DMA staging, the 64-byte swizzled LDS rows and the K-tile walk are those of the bf16x3 kernels.
//
// Operand maps (measured, tools/mx_probe2.hip): lane l = (r = l & 31, g = l >> 5).
//   v_mfma_f32_32x32x16_f16:           A[row r][k = 8g + j], B[k = 8g + j][col r], j = 0..7
//   v_mfma_scale_f32_32x32x64_f8f6f4:  byte j of the lane's 32: k = 32 (j >> 4) + 16 g + (j & 15); the scale of
//                                      (row r, block kb) is byte op_sel of the scale VGPR of lane r + 32 kb
//   C/D (both): col = l & 31, row = (q & 3) + 8 (q >> 2) + 4 g for accumulator register q = 0..15
#pragma once
#include "bf16_common.h"

namespace alvq {

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

constexpr int FX_E_ACT = 127;      // E8M0 exponent of S for activations and (loss-scaled) gradients
constexpr int FX_E_W = 119;        // ... for weights
constexpr int FX_LO_SHIFT = 11;    // lo8 is scaled by S * 2^-11

__device__ __forceinline__ float fx_pow2(int e) { return __uint_as_float((unsigned)e << 23); }   // 2^(e-127), 1 <= e <= 254

// Saturation comes from the wave's MODE register, not from clamps: with MODE.FP16_OVFL set, v_cvt_pk_f16_f32 returns
// +-65504 and the fp8 conversions +-448 where they would return infinity / NaN (measured: tools/ovfl_probe.hip; the
// clamps were 48 of the ~400 VALU instructions per 32 x 32 output tile of the convolution epilogue).  Every kernel that
// converts INTO the format calls this once per wave before its first conversion (s_setreg is a scheduling barrier).
__device__ __forceinline__ void fx_saturating_conversions() { __builtin_amdgcn_s_setreg(1 | (23 << 6), 1); }   // hwreg(MODE, 23, 1)

// two floats -> packed fp16 pair (RNE); saturating at +-65504 under fx_saturating_conversions()
__device__ __forceinline__ unsigned fx_f16_pk(float a, float b) {
  const f32x2 f = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, f16x2_t));
}
__device__ __forceinline__ float fx_h2f_lo(unsigned pk) { return (float)__builtin_bit_cast(f16x2_t, pk)[0]; }
__device__ __forceinline__ float fx_h2f_hi(unsigned pk) { return (float)__builtin_bit_cast(f16x2_t, pk)[1]; }

// four floats -> four e4m3 bytes of v / s (RNE; s a power of two, UNIFORM across the wave: it travels in an SGPR),
// saturating at +-448 s under fx_saturating_conversions() (the bare conversion returns NaN above 464).
// v_cvt_scalef32_pk_fp8_f32 divides by the scale inside the conversion and writes ONE 16-bit half of its destination,
// so the two conversions of a word need no initialised destination (the builtin form cost a v_mov per word).
__device__ __forceinline__ unsigned fx_fp8x4(float a, float b, float c, float d, float s) {
  unsigned r;
  asm("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(s));
  asm("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %3 op_sel:[0,0,0,1]" : "+v"(r) : "v"(c), "v"(d), "s"(s));
  return r;
}

// v - (float)H and lo + (float)H for one half (0 = low, 1 = high) of a packed fp16 pair, each ONE v_fma_mix_f32 reading
// the fp16 half in place (the product by +-1 is exact, so the result is the correctly rounded difference / sum, the same
// value v_cvt_f32_f16 + v_sub / v_add produce -- the compiler does not select the mixed form by itself).
template <int HALF>
__device__ __forceinline__ float fx_minus_h(float v, unsigned hpk) {
  float r;
  if (HALF == 0) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpk), "v"(v));
  else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpk), "v"(v));
  return r;
}
template <int HALF>
__device__ __forceinline__ float fx_plus_h(float lo, unsigned hpk) {
  float r;
  if (HALF == 0) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpk), "v"(lo));
  else asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpk), "v"(lo));
  return r;
}

// e4m3 byte -> float (exact); on gfx950 v_cvt_f32_fp8 with byte select
__device__ __forceinline__ float fx_fp8_to_f(unsigned word, int byte) {
  switch (byte) {
    case 0: return __builtin_amdgcn_cvt_f32_fp8((int)word, 0);
    case 1: return __builtin_amdgcn_cvt_f32_fp8((int)word, 1);
    case 2: return __builtin_amdgcn_cvt_f32_fp8((int)word, 2);
    default: return __builtin_amdgcn_cvt_f32_fp8((int)word, 3);
  }
}

// Split N (multiple of 4) consecutive values: h[N/2] packed fp16 pairs, qh[N/4] = e4m3(v / s), ql[N/4] = e4m3((v - H) / s_lo).
template <int N>
__device__ __forceinline__ void fx_split(const float (&v)[N], float s, float s_lo, unsigned (&h)[N / 2], unsigned (&qh)[N / 4],
                                         unsigned (&ql)[N / 4]) {
  float lo[N];
#pragma unroll
  for (int e = 0; e < N / 2; ++e) {
    h[e] = fx_f16_pk(v[2 * e], v[2 * e + 1]);
    lo[2 * e] = fx_minus_h<0>(v[2 * e], h[e]);
    lo[2 * e + 1] = fx_minus_h<1>(v[2 * e + 1], h[e]);
  }
#pragma unroll
  for (int e = 0; e < N / 4; ++e) {
    qh[e] = fx_fp8x4(v[4 * e], v[4 * e + 1], v[4 * e + 2], v[4 * e + 3], s);
    ql[e] = fx_fp8x4(lo[4 * e], lo[4 * e + 1], lo[4 * e + 2], lo[4 * e + 3], s_lo);
  }
}

// value of 2 consecutive channels from a packed fp16 pair and the matching lo8 bytes (word, first byte index b0 = 0 or 2):
// lo8 * S_lo by v_cvt_scalef32_pk_f32_fp8 (two bytes per instruction; the product by a power of two is exact), then
// + H by one v_fma_mix_f32 each -- one rounding, like the fused multiply-add it replaces
__device__ __forceinline__ void fx_join2(unsigned hpk, unsigned qlo, int b0, float s_lo, float& v0, float& v1) {
  const f32x2 lo = b0 ? __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(qlo, s_lo, true) : __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(qlo, s_lo, false);
  v0 = fx_plus_h<0>(lo[0], hpk);
  v1 = fx_plus_h<1>(lo[1], hpk);
}

// bit 2e / 2e+1 = (low / high fp16 half of h[e] is > 0), e = 0..7: per word max(., 0) then min(., 1) on the packed int16
// halves (a positive fp16 is a positive int16; -0, negatives and -NaN are not), shifted into place by v_lshl_or_b32
__device__ __forceinline__ unsigned fx_sign_bits_of_h(const unsigned (&h)[8]) {
  const unsigned ones = 0x00010001u;
  unsigned acc = 0;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    unsigned t;
    asm("v_pk_max_i16 %0, %1, 0" : "=v"(t) : "v"(h[e]));
    asm("v_pk_min_i16 %0, %1, %2" : "=v"(t) : "v"(t), "v"(ones));
    acc |= t << (2 * e);
  }
  return (acc & 0x5555u) | ((acc >> 15) & 0xAAAAu);
}

// byte offset of channel c (multiple of 8) inside a Q-plane row: 64 bytes per 32-channel chunk, hi8 first, lo8 at +32
__device__ __forceinline__ long fx_q_off(int c) { return (long)(c >> 5) * 64 + (c & 31); }

}  // namespace alvq
