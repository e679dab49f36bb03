// Pieces shared by the two 256 x 256-tile bf16 convolution kernels (conv1d_bf16_v2.hip: any filter width, K-tile =
// one tap; conv1d_bf16_k3.hip: width 3 with the activation slab shared by the three taps): tile constants, the
// fragment register set and the fused epilogue.
#pragma once
#include "bf16_common.h"

namespace alvq {

constexpr int V2_M = 256, V2_R = 256, V2_K = 32;
constexpr int V2_HALF = V2_M * V2_K * 2;          // 16384 B: one operand slab
constexpr int V2_CS = V2_M + 4;                   // fp32 C-slab row stride (floats)
constexpr int V2_EPI_LDS = 64 * V2_CS * 4;        // 66560 B of LDS used by the epilogue

struct FragSet {
  bf16x8_t a[8];
  bf16x8_t b[4];
};

// Epilogue of a 256 x 256 tile held as 8 x 4 MFMA fragments per wave (wave w: out-channels (w>>2)*128.., rows
// (w&3)*64..); D[i = m][j = row], so lane (li, kq) of fragment (mi, ni) holds channels mi*16 + kq*4 .. +3 of row
// ni*16 + li.
//   OUT == 0 (bf16 NLC): stored straight from the registers.  v_permlane16_swap between the fragments mi and mi+1
//     first gives every lane 8 consecutive channels (lanes kq = 0, 2: channels (kq/2)*8.. of fragment mi; kq = 1, 3:
//     of fragment mi+1), so loads and stores are 16 bytes per lane, the four lanes of a row cover 64 contiguous
//     bytes and the next fragment pair completes the 128-byte line.  No LDS round trip, no barriers, and the
//     skip / mask / post loads of different fragments are independent (they overlap instead of queueing behind one
//     another).  The previous version transposed through LDS; its ~3000 VALU instructions per thread (software
//     bf16 rounding, per-pass row decoding) made the epilogue 30 k cycles per tile -- 17 % of a width-3 tile's
//     time, 37 % of a width-1 tile's.
//   OUT == 1 (fp32 NCL, bias only; rare at this tile size): four 64-row slabs through an fp32 LDS tile so that lanes
//     run along l.  All waves must have finished reading the operand stages before the call.
// The register-direct bf16 epilogue of one wave's (NMI*16) m x (NNI*16) rows block of MFMA fragments:
// rows r0 + wn0 .., channels m0 + wm0 ..   (NMI even: fragments are swapped in pairs)
template <int NMI, int NNI, int F16 = 0>
__device__ __forceinline__ void wave_epilogue_bf16(const ConvBArgs& a, const f32x4 (&acc)[NMI][NNI], int m0, int r0, int li,
                                                   int kq, int wm0, int wn0);

template <int OUT, int F16 = 0>
__device__ __forceinline__ void tile256_epilogue(const ConvBArgs& a, const f32x4 (&acc)[8][4], unsigned char* lds, int m0,
                                                 int r0, int wave, int tid, int li, int kq, int wm0) {
  if (OUT == 0) {
    wave_epilogue_bf16<8, 4, F16>(a, acc, m0, r0, li, kq, wm0, (wave & 3) * 64);
    return;
  }
  const float oscale = a.out_scale ? *a.out_scale : 1.f;
  // OUT == 1: NCL fp32 (bias only) -- lane = row (coalesced along l), loop over channels
  float* Cs = (float*)lds;
  const int Lp1 = a.L + 1, ndata = a.B * Lp1;
  for (int slab = 0; slab < 4; ++slab) {
    if ((wave & 3) == slab) {
#pragma unroll
      for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const int rl = ni * 16 + li, ml = wm0 + mi * 16 + kq * 4;
          *(f32x4*)(Cs + rl * V2_CS + ml) = acc[mi][ni];
        }
    }
    __syncthreads();
    const int rl = tid & 63, row = r0 + slab * 64 + rl;
    int b, l;
    if (row_valid(row, Lp1, ndata, &b, &l)) {
      for (int ml = tid >> 6; ml < V2_M; ml += 8) {
        const int m = m0 + ml;
        if (m >= a.M) break;
        a.y_ncl[((long)b * a.M + m) * a.L + l] = (Cs[rl * V2_CS + ml] + (a.bias ? a.bias[m] : 0.f)) * oscale;
      }
    }
    __syncthreads();
  }
}

// Written for few VALU instructions per value -- the epilogue of a 256 x 256 tile is 16 of these 8-channel groups per
// lane, and at ~100 instructions each it took twice the time its 2 bytes per element need on the way to HBM (measured by
// compiling it out: 13 us of a width-3 launch's 86 us per round, 10 of a width-1 launch's 41):
//  * a row block's pointers are formed once, the groups add constants to them;
//  * nothing is computed for operands that are absent (no bias -> no bias vector of zeros to add);
//  * gap / tail rows (one 16-row block in thirty holds one) are zeroed by a select on the four packed output words,
//    inside a wave-uniform branch, instead of a divergent branch around the whole group;
//  * the bias of a group that lies inside M is two 16-byte loads.
template <int NMI, int NNI, int F16>
__device__ __forceinline__ void wave_epilogue_bf16(const ConvBArgs& a, const f32x4 (&acc)[NMI][NNI], int m0, int r0, int li,
                                                   int kq, int wm0, int wn0) {
  static_assert(NMI % 2 == 0, "fragments are swapped in pairs");
  elem_saturate<F16>();
  unsigned watch = 0;
  const int Lp1 = a.L + 1, ndata = a.B * Lp1;
  const int mb0 = m0 + wm0 + (kq & 1) * 16 + (kq >> 1) * 8;      // this lane's 8 channels of fragment pair 0
#pragma unroll
  for (int ni = 0; ni < NNI; ++ni) {
    const int row = r0 + wn0 + ni * 16 + li;
    int b, l;
    const bool ok = row_valid(row, Lp1, ndata, &b, &l);
    const bool gaps = !__all(ok);
    const long o0 = (long)row * a.Mop + mb0;                      // element offset of the lane's first group
    // the skip / mask operands of the whole row block are requested before the first group is finished (a load ->
    // wait -> use chain per group is one L2 / HBM round trip each, sixteen per lane)
    u16x8 s1[NMI / 2], s2[NMI / 2], mk[NMI / 2];
#pragma unroll
    for (int mp = 0; mp < NMI; mp += 2) {
      if (m0 + wm0 + mp * 16 >= a.Mop) continue;
      if (a.skip1) s1[mp / 2] = *(const u16x8*)(a.skip1 + o0 + mp * 16);
      if (a.skip2) s2[mp / 2] = *(const u16x8*)(a.skip2 + o0 + mp * 16);
      if (a.mask && !a.mask_bits) mk[mp / 2] = *(const u16x8*)(a.mask + o0 + mp * 16);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mp = 0; mp < NMI; mp += 2) {
      if (m0 + wm0 + mp * 16 >= a.Mop) continue;        // Mop % 64 == 0 and the pair starts on a multiple of 32
      const long o = o0 + mp * 16;
      const int mb = mb0 + mp * 16;
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // odd rows (of 16 lanes) of the first operand <-> even rows of the second
        const u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[mp][ni][e]), __float_as_uint(acc[mp + 1][ni][e]),
                                                         false, false);
        v[e] = __uint_as_float(r[0]);
        v[e + 4] = __uint_as_float(r[1]);
      }
      if (a.bias) {
        if (m0 + wm0 + mp * 16 + 32 <= a.M) {
          const f32x4 b0 = *(const f32x4*)(a.bias + mb), b1 = *(const f32x4*)(a.bias + mb + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] += b0[e];
            v[4 + e] += b1[e];
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += (mb + e < a.M) ? a.bias[mb + e] : 0.f;
        }
      }
      if (a.skip1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += elem2f<F16>(s1[mp / 2][e]);
      }
      if (a.skip2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += elem2f<F16>(s2[mp / 2][e]);
      }
      if (a.relu & 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      if (a.mask_bits) {     // sign-extend bit e to a word and AND
        const int bt = (int)a.mask_bits[o >> 3];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = __uint_as_float(__float_as_uint(v[e]) & (unsigned)((bt << (31 - e)) >> 31));
      } else if (a.mask) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (short)mk[mp / 2][e] > 0 ? v[e] : 0.f;   // a positive bf16 / fp16 is a positive int16
      }
      u32x4 out;
#pragma unroll
      for (int e = 0; e < 4; ++e) out[e] = elem_pk<F16>(v[2 * e], v[2 * e + 1]);
      if (gaps) {
#pragma unroll
        for (int e = 0; e < 4; ++e) out[e] = ok ? out[e] : 0u;
      }
      *(u32x4*)(a.y + o) = out;
      if (F16) {
#pragma unroll
        for (int e = 0; e < 4; ++e) fp16_watch(watch, out[e]);
      }
      if (a.y2) {
        const u16x8 ps = *(const u16x8*)(a.post + o);
        u32x4 out2;
#pragma unroll
        for (int e = 0; e < 4; ++e) out2[e] = elem_pk<F16>(v[2 * e] + elem2f<F16>(ps[2 * e]), v[2 * e + 1] + elem2f<F16>(ps[2 * e + 1]));
        if (gaps) {
#pragma unroll
          for (int e = 0; e < 4; ++e) out2[e] = ok ? out2[e] : 0u;
        }
        *(u32x4*)(a.y2 + o) = out2;
      }
      if (a.bits_out) {   // bit e = (stored y[e] > 0): a bf16 / fp16 in the upper half of a word is positive iff the word is, as an int
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
  }
  if (F16) fp16_report(watch, a.range_flag);
}

}  // namespace alvq
