// f16mx convolution (see f16mx_common.h for the format): forward / data-gradient / ConvTranspose of the wide and narrow
// layers alike, plus the boundary conversions of the format.
//
// Tiling is that of conv1d_bf16x3.hip -- 256 out-channels x 256 rows per workgroup, 8 waves, a wave owns 128 x 64;
// K-tile = (32 channels, one tap) -- with two LDS-DMA rings: the weight slabs (W.H, W.Q; 32 KB) per K-tile and the
// activation slabs (X.H, X.Q; 34 KB) per CHUNK of 32 channels, shared by the taps as in conv1d_bf16_k3.hip.  The wave's
// block is 4 x 2 tiles of 32x32 and a K-tile is TWO phases of 512 matrix-pipe cycles:
//   phase 1  fp16 main term : 2 k-steps x 8 v_mfma_f32_32x32x16_f16          | meanwhile: the Q fragments of this K-tile
//            s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier    <- K-tile t+1 landed; every read of this stage is done
//   phase 2  fp8 cross terms: 8 v_mfma_scale_f32_32x32x64_f8f6f4              | meanwhile: DMA of K-tile t+2 into THIS
//                                                                               stage, H fragments of K-tile t+1
// 96 fragment VGPRs + 128 accumulators, the barrier between phases whose operands are in registers, as in bf16x3.
#include <stdlib.h>

#include "alvq_common.h"
#include "f16mx_common.h"

namespace alvq {

constexpr int FX_M = 256, FX_R = 256, FX_K = 32;
constexpr int FX_SLAB = FX_M * FX_K * 2;          // 16384 B
constexpr int FX_WSTAGE = 2 * FX_SLAB;            // W.H, W.Q of one K-tile
constexpr int FX_XSLAB = 272 * 64;                // 272 rows x 64 B (258 used: 256 + a halo row either side)
constexpr int FX_XSTAGE = 2 * FX_XSLAB;           // X.H, X.Q of one chunk
constexpr int FX_LDS = 2 * FX_WSTAGE + 2 * FX_XSTAGE;   // 135168 B
constexpr int FX_CS = FX_M + 4;
static_assert(64 * FX_CS * 4 <= FX_LDS, "C slab must fit");

struct ConvFxArgs {
  ConvBArgs b;                        // H planes (and everything shared); mask_bits / bits_out unused
  long x_plane, wp_plane, y_plane;    // element (u16) offsets from an H plane to its Q plane
  int ea, eb;                         // E8M0 scale exponents: weights (A operand), activations / gradients (B operand and outputs)
  const float* out_scale;             // OUT == 1: device scalar multiplied into the fp32 output (undoes a loss scale), or null
  int* range_flag;                    // sticky device flag, |= 4 when a stored H reached fp16's limit
  unsigned long long* stamps;         // DBG instantiations, ALVQ_FX_DBG & 256: per workgroup {start, staged, main loop done, end} (100 MHz)
  int dbg;                            // switches of the DBG instantiations (ALVQ_FX_DBG), timing experiments only: 1 no in-loop
                                      // DMA, 2 no in-loop fragment reads, 4 no wait + barrier, 8 no epilogue, 16 / 32 no fp16 /
                                      // no fp8 MFMAs, 256 phase stamps (results intact), 512 stamps + the epilogue's stores removed
};

// 16 consecutive channels of one row, reconstructed from an f16mx tensor: += H + lo8 * S_lo.  ph -> the row's H at the
// first channel, pq -> the 64-byte Q chunk position of that channel (hi8; lo8 at +32).
__device__ __forceinline__ void fx_join16(const u32x4& h0, const u32x4& h1, const u32x4& ql, float s_lo, float (&v)[16]) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float a0, a1, b0, b1;
    fx_join2(h0[e], ql[e >> 1], (e & 1) * 2, s_lo, a0, a1);
    fx_join2(h1[e], ql[2 + (e >> 1)], (e & 1) * 2, s_lo, b0, b1);
    v[2 * e] += a0;
    v[2 * e + 1] += a1;
    v[8 + 2 * e] += b0;
    v[8 + 2 * e + 1] += b1;
  }
}
__device__ __forceinline__ void fx_load_add16(const u16* ph, const unsigned char* pq, float s_lo, float (&v)[16]) {
  const u32x4 h0 = *(const u32x4*)ph, h1 = *(const u32x4*)(ph + 8);
  const u32x4 ql = *(const u32x4*)(pq + 32);
  fx_join16(h0, h1, ql, s_lo, v);
}

// Store 16 consecutive channels of one row (H: 32 bytes, hi8 / lo8: 16 bytes each) and return the packed H words (the
// sign bits of a ReLU'd output are derived from them, fx_sign_bits_of_h).
template <bool NOSTORE = false>
__device__ __forceinline__ void fx_store16(u16* ph, unsigned char* pq, float s_hi, float s_lo, const float (&v)[16], unsigned (&h)[8]) {
  unsigned qh[4], ql[4];
  fx_split<16>(v, s_hi, s_lo, h, qh, ql);
  if (NOSTORE) {   // timing ablation (ALVQ_FX_DBG & 512): the conversions stay, the stores go
#pragma unroll
    for (int e = 0; e < 4; ++e) asm volatile("" ::"v"(h[e]), "v"(h[4 + e]), "v"(qh[e]), "v"(ql[e]));
    return;
  }
  *(u32x4*)ph = u32x4{h[0], h[1], h[2], h[3]};
  *(u32x4*)(ph + 8) = u32x4{h[4], h[5], h[6], h[7]};
  *(u32x4*)pq = u32x4{qh[0], qh[1], qh[2], qh[3]};
  *(u32x4*)(pq + 32) = u32x4{ql[0], ql[1], ql[2], ql[3]};
}

// Operand loads come FIRST and in bulk: one load -> use -> store chain per 32-channel tile made the epilogue a string of
// eight L2 / HBM round trips (17 us per workgroup on an otherwise idle chip, before any bandwidth limit).  The skip and
// mask operands of a whole 32-row block (four tiles, 48 + 32 registers -- the fragment registers are free by now) are
// requested at once, and the second row block's while the first is being converted and stored.  Rows outside the data
// (gap / tail rows: every plane holds them, as zeros) are loaded like any other and zeroed by a select, so the code has
// no divergent branch.
struct FxEpiLoads {
  u32x4 h0[4], h1[4], ql[4];   // skip1: H (2 x 16 B) and lo8 of 16 channels, per 32-channel tile
  u32x4 m0[4], m1[4];          // mask as a tensor: H
  unsigned mb[4];              // mask as bits
};

template <int NIN, int DBG = 0>
__device__ __forceinline__ void wave_epilogue_fx(const ConvFxArgs& ax, const f32x16 (&acc)[4][2], int m0, int r0, int lane, int wm0,
                                                 int wn0) {
  const ConvBArgs& a = ax.b;
  const int j = lane & 31, h = lane >> 5;
  const int Lp1 = a.L + 1, ndata = a.B * Lp1;
  const float s_lo = fx_pow2(ax.eb - FX_LO_SHIFT), s_hi = fx_pow2(ax.eb);
  fx_saturating_conversions();
  const int cb0 = m0 + wm0 + 16 * h;
  // Everything below addresses a row block through pointers to (row, channel cb0) computed once; a tile adds the
  // constants 32 mi channels = 64 mi bytes of H, one Q chunk (64 bytes) per tile, 4 mi bytes of sign bits.
  const long qo0 = fx_q_off(cb0);
  FxEpiLoads ld[NIN];
  unsigned watch = 0;
  // per row block, once (the row -> (b, l) division is ~20 VALU instructions): is this lane's row a data row; does the
  // block hold any row that is not
  bool okr[NIN], gapsr[NIN];
#pragma unroll
  for (int ni = 0; ni < NIN; ++ni) {
    int b, l;
    okr[ni] = row_valid(r0 + wn0 + ni * 32 + j, Lp1, ndata, &b, &l);
    gapsr[ni] = !__all(okr[ni]);
  }
  auto request = [&](int ni) {
    const long ro = (long)(r0 + wn0 + ni * 32 + j) * a.Mop;
    const u16* s1h = a.skip1 + ro + cb0;
    const unsigned char* s1q = (const unsigned char*)(a.skip1 + ax.y_plane) + ro * 2 + qo0 + 32;
    const u16* mkh = a.mask + ro + cb0;
    const unsigned char* mkb = a.mask_bits + ((ro + cb0) >> 3);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      if (m0 + wm0 + mi * 32 >= a.Mop) continue;        // Mop % 64 == 0: a 32-channel tile is inside or outside as a whole
      if (a.skip1) {
        ld[ni].h0[mi] = *(const u32x4*)(s1h + mi * 32);
        ld[ni].h1[mi] = *(const u32x4*)(s1h + mi * 32 + 8);
        ld[ni].ql[mi] = *(const u32x4*)(s1q + mi * 64);
      }
      if (a.mask_bits) {   // one bit per element, left behind by the ReLU'd launch that made the tensor: 1/16 of its H plane
        ld[ni].mb[mi] = *(const unsigned short*)(mkb + mi * 4);
      } else if (a.mask) {
        ld[ni].m0[mi] = *(const u32x4*)(mkh + mi * 32);
        ld[ni].m1[mi] = *(const u32x4*)(mkh + mi * 32 + 8);
      }
    }
  };
  auto tile = [&](int ni, int mi) {
    if (m0 + wm0 + mi * 32 >= a.Mop) return;
    const bool ok = okr[ni], gaps = gapsr[ni];
    const long ro = (long)(r0 + wn0 + ni * 32 + j) * a.Mop;
    const long hoff = ro + cb0 + mi * 32;                 // elements from a plane's start to this lane's 16 channels
    const long qoff = ro * 2 + qo0 + mi * 64;             // bytes into the Q plane
    float v[16];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const u32x2 ra = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[mi][ni][e]), __float_as_uint(acc[mi][ni][8 + e]), false, false);
      const u32x2 rb = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[mi][ni][4 + e]), __float_as_uint(acc[mi][ni][12 + e]), false, false);
      v[e] = __uint_as_float(ra[0]);
      v[4 + e] = __uint_as_float(ra[1]);
      v[8 + e] = __uint_as_float(rb[0]);
      v[12 + e] = __uint_as_float(rb[1]);
    }
    if (a.bias) {
      const int cb = cb0 + mi * 32;
      if (m0 + wm0 + mi * 32 + 32 <= a.M) {              // the whole 32-channel tile is real channels: four 16-byte loads
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 bq = *(const f32x4*)(a.bias + cb + 4 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[4 * q + e] += bq[e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] += (cb + e < a.M) ? a.bias[cb + e] : 0.f;
      }
    }
    if (a.skip1) fx_join16(ld[ni].h0[mi], ld[ni].h1[mi], ld[ni].ql[mi], s_lo, v);
    if (a.skip2) fx_load_add16(a.skip2 + hoff, (const unsigned char*)(a.skip2 + ax.y_plane) + qoff, s_lo, v);
    if (a.relu & 1) {      // one v_max_f32 per value (fmaxf costs a second one: the compiler canonicalises its operand first)
#pragma unroll
      for (int e = 0; e < 16; ++e) asm("v_max_f32 %0, 0, %1" : "=v"(v[e]) : "v"(v[e]));
    }
    if (a.mask_bits) {     // sign-extend bit e to a word (v_bfe_i32) and AND: two VALU instructions per element; written
                           // as asm because the compiler turns the C form into and + compare + select with wait states
      const unsigned bt = ld[ni].mb[mi];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        unsigned m;
        asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(bt), "n"(e));
        v[e] = __uint_as_float(__float_as_uint(v[e]) & m);
      }
    } else if (a.mask) {   // the sign of a split value is the sign of its H plane (fp16 reaches 6e-8; smaller activations are zero)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[2 * e] = fx_h2f_lo(ld[ni].m0[mi][e]) > 0.f ? v[2 * e] : 0.f;
        v[2 * e + 1] = fx_h2f_hi(ld[ni].m0[mi][e]) > 0.f ? v[2 * e + 1] : 0.f;
        v[8 + 2 * e] = fx_h2f_lo(ld[ni].m1[mi][e]) > 0.f ? v[8 + 2 * e] : 0.f;
        v[8 + 2 * e + 1] = fx_h2f_hi(ld[ni].m1[mi][e]) > 0.f ? v[8 + 2 * e + 1] : 0.f;
      }
    }
    if (gaps) {                            // one 32-row block in sixteen holds a gap row: gap / tail rows stay zero
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = ok ? v[e] : 0.f;
    }
    unsigned hw[8];
    if (DBG && (ax.dbg & 512)) fx_store16<true>(a.y + hoff, (unsigned char*)(a.y + ax.y_plane) + qoff, s_hi, s_lo, v, hw);
    else fx_store16(a.y + hoff, (unsigned char*)(a.y + ax.y_plane) + qoff, s_hi, s_lo, v, hw);
#pragma unroll
    for (int e = 0; e < 8; ++e) fp16_watch(watch, hw[e]);
    if (a.bits_out) *(unsigned short*)(a.bits_out + (hoff >> 3)) = (unsigned short)fx_sign_bits_of_h(hw);
    if (a.y2) {
      fx_load_add16(a.post + hoff, (const unsigned char*)(a.post + ax.y_plane) + qoff, s_lo, v);
      if (gaps) {
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = ok ? v[e] : 0.f;
      }
      fx_store16(a.y2 + hoff, (unsigned char*)(a.y2 + ax.y_plane) + qoff, s_hi, s_lo, v, hw);
    }
  };
  request(0);
  __builtin_amdgcn_sched_barrier(0);
  tile(0, 0);
  tile(0, 1);
  if (NIN > 1) {       // two tiles' accumulators and loads have been retired: room for the second row block's requests
    __builtin_amdgcn_sched_barrier(0);
    request(NIN - 1);
    __builtin_amdgcn_sched_barrier(0);
  }
  tile(0, 2);
  tile(0, 3);
  if (NIN > 1) {
    tile(NIN - 1, 0);
    tile(NIN - 1, 1);
    tile(NIN - 1, 2);
    tile(NIN - 1, 3);
  }
  fp16_report(watch, ax.range_flag);
}

// DBG: 1 run-time ablation switches, 2 also no fp16 MFMAs, 3 also no fp8 MFMAs (timing experiments, tools/ablate_f16mx.sh).
// NIN: 32-row blocks per wave.  2 = the 256-row tile; 1 = a 128-row tile (a wave owns 128 x 32) for launches whose 256-row
// grid would leave CUs idle (one m-tile: M <= 256, or few rows) -- twice the workgroups at 5/4 instead of 6/8 fragment
// reads per MFMA.
// MIN: 32-channel blocks per wave.  4 = the 256-channel m-tile; 2 = a 128-channel m-tile (a wave owns 64 x 32, with
// NIN = 1) for layers of at most 128 output channels, whose 256-wide tile would spend half of its MFMAs on padding.
template <int OUT, int KW, int DBG = 0, int NIN = 2, int MIN = 4>
__global__ __launch_bounds__(512, 2) void conv1d_f16mx_kernel(ConvFxArgs ax) {
  constexpr int PAD = (KW - 1) / 2;
  constexpr int RT = 128 * NIN;          // rows per workgroup
  constexpr int MT = 64 * MIN;           // output channels per workgroup
  const int dbg = DBG ? ax.dbg : 0;
  const ConvBArgs& a = ax.b;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave >> 2) * 32 * MIN, wn0 = (wave & 3) * 32 * NIN;

  const int tile = xcd_remap(blockIdx.x, a.mtiles * a.rtiles);
  const int m0 = (tile % a.mtiles) * MT;
  const int r0 = (tile / a.mtiles) * RT;
  const int Cp = a.Cp;

  // ---- DMA (a piece is 16 rows x 64 B; lane i -> row i>>2, slot i&3 <- 16-byte group (i&3) ^ h[(row>>2)&3] of the
  // row's 64-byte chunk; plane 0 = H, plane 1 = Q).  Two rings: the WEIGHT slabs of one K-tile (32 channels, one tap) and
  // the ACTIVATION slabs of one CHUNK (32 channels, all taps): rows r0-PAD .. r0+255+PAD are staged once and tap t reads
  // the slab t rows further down, so a width-3 layer moves 3 x 32 KB of weights + 34 KB of activations per chunk through
  // LDS-DMA instead of 3 x 64 KB -- a third less DMA issue and L2 -> LDS traffic for the same MFMAs.
  const int hsel = (lane >> 4) & 3;
  const int hval = (hsel == 0) ? 0 : (4 - hsel);
  const int srow = lane >> 2, sgrp = (lane & 3) ^ hval;
  const unsigned lane_off = (unsigned)(srow * Cp + sgrp * 8) * 2u;
  const long tap_w = (long)a.Mp128 * Cp * 2;
  const long row16 = (long)Cp * 32;
  const long wpl = ax.wp_plane * 2, xpl = ax.x_plane * 2;
  const unsigned lds0 = (unsigned)(unsigned long)((__attribute__((address_space(3))) unsigned char*)lds);
  auto dma = [&](const char* sbase, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_off), "s"(sbase), "s"(lds_dst)
                 : "memory");
  };
  const char* const wb = (const char*)(a.wp + ((long)m0 + wave * 32) * Cp);
  const char* const xb = (const char*)(a.x + ((long)r0 - PAD + wave * 32) * Cp);
  constexpr unsigned XBASE = 2 * FX_WSTAGE;
  // K-tile t -> weight stage t & 1: 32 rows of W.H and of W.Q per wave = pieces 0..3 (issued one at a time so that each
  // can sit in the shadow of an MFMA)
  auto pieceW = [&](int t, int k) {
    if (wave * 32 >= MT) return;          // 128-channel m-tile: waves 0-3 stage the weight rows
    const int chunk = t / KW, tap = t - chunk * KW;
    const unsigned dst = lds0 + (t & 1) * FX_WSTAGE + wave * 2048 + (k >> 1) * FX_SLAB + (k & 1) * 1024;
    dma(wb + tap * tap_w + chunk * (FX_K * 2) + (k >> 1) * wpl + (k & 1) * row16, dst);
  };
  auto issueW = [&](int t) {
    pieceW(t, 0);
    pieceW(t, 1);
    pieceW(t, 2);
    pieceW(t, 3);
  };
  // one plane of a chunk's activation slab -> activation stage chunk & 1: pieces 0, 1 (+ the halo rows 256, 257)
  auto pieceX = [&](int chunk, int plane, int k) {
    const unsigned dst = lds0 + XBASE + (chunk & 1) * FX_XSTAGE + plane * FX_XSLAB + wave * 2048;
    const char* xs = xb + plane * xpl + chunk * (FX_K * 2);
    if (wave * 32 >= RT) return;          // 128-row tile: waves 0-3 stage the activation rows
    if (k < 2) dma(xs + k * row16, dst + k * 1024);
    else if (KW == 3 && wave == RT / 32 - 1 && srow < 2) dma(xs + 2 * row16, dst + 2048);
  };
  auto issueX = [&](int chunk, int plane) {
    pieceX(chunk, plane, 0);
    pieceX(chunk, plane, 1);
    pieceX(chunk, plane, 2);
  };

  // ---- fragment reads for the 32x32 shapes: lane (r = lane & 31, g = lane >> 5) takes the 16-byte groups g and 2 + g of
  // row r of a 32-row block.  H slab: group g = k-step 0 (channels 8g..8g+7), group 2+g = k-step 1.  Q slab: group g =
  // hi8[16g..16g+15], group 2+g = lo8[16g..16g+15]; the A operand wants (hi8, lo8), the B operand (lo8, hi8), so that
  // block 0 of the scaled MFMA pairs A.hi8 with B.lo8 and block 1 A.lo8 with B.hi8.  The slot swizzle of the staging
  // (slot = group ^ {0,3,2,1}[(row>>2)&3]) makes the un-shifted reads conflict-free for this lane pattern too; the
  // activation reads of tap t use slab row r + t.
  const int r32 = lane & 31, g = lane >> 5;
  int offA[2], offB[KW][2];
  {
    const int hq = (r32 >> 2) & 3, hsw = hq == 0 ? 0 : 4 - hq;
    offA[0] = r32 * 64 + ((g ^ hsw) << 4);
    offA[1] = r32 * 64 + (((2 + g) ^ hsw) << 4);
  }
#pragma unroll
  for (int t = 0; t < KW; ++t) {
    const int rr = r32 + t, hq = (rr >> 2) & 3, hsw = hq == 0 ? 0 : 4 - hq;
    offB[t][0] = rr * 64 + ((g ^ hsw) << 4);
    offB[t][1] = rr * 64 + (((2 + g) ^ hsw) << 4);
  }
  const unsigned char* const abase = lds + wm0 * 64;
  const unsigned char* const bbase = lds + XBASE + wn0 * 64;
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  f16x8_t aH[4][2], bH[2][2];
  i32x8 aQ[4], bQ[2];
#define FX_RDH_A(WS, MI, KS) if ((MI) < MIN) aH[MI][KS] = *(const f16x8_t*)(abase + (WS) * FX_WSTAGE + (MI) * 2048 + offA[KS]);
#define FX_RDH_B(XS, TAP, NI, KS) if ((NI) < NIN) bH[NI][KS] = *(const f16x8_t*)(bbase + (XS) * FX_XSTAGE + (NI) * 2048 + offB[TAP][KS]);
#define FX_RDQ(DST, P, FIRST, SECOND)                                                           \
  {                                                                                             \
    const i32x4 q0_ = *(const i32x4*)((P) + (FIRST)), q1_ = *(const i32x4*)((P) + (SECOND));    \
    DST = __builtin_shufflevector(q0_, q1_, 0, 1, 2, 3, 4, 5, 6, 7);                            \
  }
#define FX_RDQ_A(WS, MI) if ((MI) < MIN) FX_RDQ(aQ[MI], abase + (WS) * FX_WSTAGE + FX_SLAB + (MI) * 2048, offA[0], offA[1])
#define FX_RDQ_B(XS, TAP, NI) if ((NI) < NIN) FX_RDQ(bQ[NI], bbase + (XS) * FX_XSTAGE + FX_XSLAB + (NI) * 2048, offB[TAP][1], offB[TAP][0])

  // block scales of the fp8 MFMA: lanes 0-31 supply block 0, lanes 32-63 block 1.  Opaque to the compiler so that it keeps
  // them in registers instead of re-materialising them by VALU moves in front of the inline-asm MFMAs (no hazard padding
  // happens for instructions it cannot see).
  int sa = g ? ax.ea - FX_LO_SHIFT : ax.ea, sb = g ? ax.eb : ax.eb - FX_LO_SHIFT;
  asm volatile("" : "+v"(sa), "+v"(sb));

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jn = 0; jn < 2; ++jn)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][jn][q] = 0.f;
#define FX_H(MI, NI, KS) if (DBG != 2 && (NI) < NIN && (MI) < MIN) asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[MI][NI]) : "v"(aH[MI][KS]), "v"(bH[NI][KS]));
#define FX_Q(MI, NI)                                                                          \
  if ((NI) < NIN && (MI) < MIN && DBG != 3)                                                                        \
    asm("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]"            \
        : "+v"(acc[MI][NI]) : "v"(aQ[MI]), "v"(bQ[NI]), "v"(sa), "v"(sb));
#define FX_SB __builtin_amdgcn_sched_barrier(0);

  const int nch = Cp / FX_K;        // chunks; even (Cp % 64 == 0)
  const int n = nch * KW;           // K-tiles
  const bool early = wave < 4;      // the two waves of a SIMD issue their DMA at different points of phase 2

  unsigned long long st0 = 0, st1 = 0, st2 = 0;
  if (DBG) st0 = __builtin_amdgcn_s_memrealtime();
  // ---- prologue: chunk 0's activation slabs, K-tiles 0 and 1 (and, for width 1, chunk 1's slabs) staged; H fragments of
  // K-tile 0 in registers
  issueX(0, 0);
  issueX(0, 1);
  issueW(0);
  if (n > 1) issueW(1);
  if (KW == 1 && nch > 1) {
    issueX(1, 0);
    issueX(1, 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (DBG) st1 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
  for (int mi = 0; mi < MIN; ++mi) { FX_RDH_A(0, mi, 0) FX_RDH_A(0, mi, 1) }
#pragma unroll
  for (int ni = 0; ni < NIN; ++ni) { FX_RDH_B(0, 0, ni, 0) FX_RDH_B(0, 0, ni, 1) }

  // One K-tile: weights in stage WS, activations in stage XS read at tap TAP; the next K-tile's are (NWS, NXS, NTAP).
  // TW = the K-tile whose weights are staged behind this tile's barrier (into the weight stage this tile occupied), or
  // -1; XC / XP = chunk and plane of the activation slab staged here, or XC = -1.
  // Every LDS read and every DMA piece sits in the shadow of ONE MFMA (at most two 16-byte reads or one DMA piece plus
  // two reads per gap): issued in bursts -- all twelve reads of a phase in a row, by both waves of a SIMD at the same
  // point of the program -- they left the matrix pipe idle for the length of the burst (measured: 17 % of the kernel).
  // Reads are ordered by first use; the early / late halves of the workgroup place their DMA pieces in different gaps.
#define FX_RD(X_) if (!(dbg & 2)) { X_ }
#define FX_DMA_E(X_) if (early && !(dbg & 1)) { X_ }
#define FX_DMA_L(X_) if (!early && !(dbg & 1)) { X_ }
#define FX_TILE(WS, XS, TAP, NWS, NXS, NTAP, TW, XC, XP)                                                                 \
  FX_H(0, 0, 0) FX_SB FX_RD(FX_RDQ_B(XS, TAP, 0)) FX_SB                                                                   \
  FX_H(0, 1, 0) FX_SB FX_RD(FX_RDQ_A(WS, 0)) FX_SB                                                                        \
  FX_H(1, 0, 0) FX_SB FX_RD(FX_RDQ_B(XS, TAP, 1)) FX_SB                                                                   \
  FX_H(1, 1, 0) FX_SB FX_RD(FX_RDQ_A(WS, 1)) FX_SB                                                                        \
  FX_H(2, 0, 0) FX_SB FX_RD(FX_RDQ_A(WS, 2)) FX_SB                                                                        \
  FX_H(2, 1, 0) FX_SB FX_RD(FX_RDQ_A(WS, 3)) FX_SB                                                                        \
  FX_H(3, 0, 0) FX_H(3, 1, 0)                                                                                            \
  FX_H(0, 0, 1) FX_H(0, 1, 1) FX_H(1, 0, 1) FX_H(1, 1, 1) FX_H(2, 0, 1) FX_H(2, 1, 1) FX_H(3, 0, 1) FX_H(3, 1, 1) FX_SB   \
  if (!(dbg & 4)) {                                                                                                      \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                          \
    __builtin_amdgcn_s_barrier();                                                                                        \
  }                                                                                                                      \
  FX_Q(0, 0) FX_SB FX_DMA_E(if ((TW) >= 0) pieceW(TW, 0);) FX_RD(FX_RDH_B(NXS, NTAP, 0, 0) FX_RDH_A(NWS, 0, 0)) FX_SB     \
  FX_Q(0, 1) FX_SB FX_DMA_E(if ((TW) >= 0) pieceW(TW, 1);) FX_RD(FX_RDH_B(NXS, NTAP, 1, 0) FX_RDH_A(NWS, 1, 0)) FX_SB     \
  FX_Q(1, 0) FX_SB FX_DMA_E(if ((TW) >= 0) pieceW(TW, 2);) FX_RD(FX_RDH_A(NWS, 2, 0) FX_RDH_A(NWS, 3, 0)) FX_SB           \
  FX_Q(1, 1) FX_SB FX_DMA_E(if ((TW) >= 0) pieceW(TW, 3);) FX_RD(FX_RDH_B(NXS, NTAP, 0, 1) FX_RDH_B(NXS, NTAP, 1, 1)) FX_SB \
  FX_Q(2, 0) FX_SB FX_DMA_L(if ((TW) >= 0) { pieceW(TW, 0); pieceW(TW, 1); }) FX_RD(FX_RDH_A(NWS, 0, 1) FX_RDH_A(NWS, 1, 1)) FX_SB \
  FX_Q(2, 1) FX_SB FX_DMA_L(if ((TW) >= 0) { pieceW(TW, 2); pieceW(TW, 3); }) FX_RD(FX_RDH_A(NWS, 2, 1) FX_RDH_A(NWS, 3, 1)) FX_SB \
  FX_Q(3, 0) FX_SB if (!(dbg & 1) && (XC) >= 0) { pieceX(XC, XP, 0); pieceX(XC, XP, 1); pieceX(XC, XP, 2); } FX_SB       \
  FX_Q(3, 1) FX_SB if (KW == 1 && !(dbg & 1) && (XC) >= 0) { pieceX(XC, 1, 0); pieceX(XC, 1, 1); } FX_SB

  if (KW == 3) {
    // two chunks (six K-tiles) per iteration so that every stage index is a constant.  The next chunk's activation slabs
    // are staged behind the barriers of this chunk's taps 0 (H plane) and 1 (Q plane): its stage was released by the
    // previous chunk's last barrier, and it is first read behind this chunk's last one.
    for (int c = 0; c < nch; c += 2) {
      const int t = 3 * c;
      const int x1 = c + 1 < nch ? c + 1 : -1, x2 = c + 2 < nch ? c + 2 : -1;
      FX_TILE(0, 0, 0, 1, 0, 1, (t + 2 < n ? t + 2 : -1), x1, 0)
      FX_TILE(1, 0, 1, 0, 0, 2, (t + 3 < n ? t + 3 : -1), x1, 1)
      FX_TILE(0, 0, 2, 1, 1, 0, (t + 4 < n ? t + 4 : -1), -1, 0)
      FX_TILE(1, 1, 0, 0, 1, 1, (t + 5 < n ? t + 5 : -1), x2, 0)
      FX_TILE(0, 1, 1, 1, 1, 2, (t + 6 < n ? t + 6 : -1), x2, 1)
      FX_TILE(1, 1, 2, 0, 0, 0, (t + 7 < n ? t + 7 : -1), -1, 0)
    }
  } else {
    // width 1: a chunk is one K-tile; both planes of chunk t + 2 follow its weights (last two gaps of the tile)
    for (int t = 0; t < n; t += 2) {
      const int a2 = t + 2 < n ? t + 2 : -1, a3 = t + 3 < n ? t + 3 : -1;
      FX_TILE(0, 0, 0, 1, 1, 0, a2, a2, 0)
      FX_TILE(1, 1, 0, 0, 0, 0, a3, a3, 0)
    }
  }
#undef FX_TILE
#undef FX_DMA_L
#undef FX_DMA_E
#undef FX_RD
#undef FX_SB
#undef FX_Q
#undef FX_H
#undef FX_RDQ_B
#undef FX_RDQ_A
#undef FX_RDQ
#undef FX_RDH_B
#undef FX_RDH_A
  // the compiler's hazard recogniser does not see inside the asm MFMAs: cover the MFMA-result -> VALU-read wait states
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

  if (DBG) st2 = __builtin_amdgcn_s_memrealtime();
  if (dbg & 8) {
    if (acc[0][0][0] == 12345.678f) a.y[0] = 1;   // keep the accumulators alive
    return;
  }
  if (OUT == 0) {
    wave_epilogue_fx<NIN, DBG>(ax, acc, m0, r0, lane, wm0, wn0);
    if (DBG && (dbg & (256 | 512))) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the stores have left
      __syncthreads();
      if (tid == 0) {
        unsigned long long* o = ax.stamps + (long)blockIdx.x * 4;
        o[0] = st0; o[1] = st1; o[2] = st2; o[3] = __builtin_amdgcn_s_memrealtime();
      }
    }
    return;
  }
  __syncthreads();   // the C slab overlays the stages: the trailing fragment reads of every wave must be done
  // ---- OUT == 1 (fp32 NCL, bias only, optional output scale): 64-row slabs through an fp32 LDS tile
  float* Cs = (float*)lds;
  const int Lp1 = a.L + 1, ndata = a.B * Lp1;
  const float oscale = ax.out_scale ? *ax.out_scale : 1.f;
  for (int slab = 0; slab < RT / 64; ++slab) {
    if ((wn0 >> 6) == slab) {
#pragma unroll
      for (int mi = 0; mi < MIN; ++mi)
#pragma unroll
        for (int ni = 0; ni < NIN; ++ni)
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const int rl = (wn0 & 63) + ni * 32 + r32, ml = wm0 + mi * 32 + 8 * q4 + 4 * g;
            *(f32x4*)(Cs + rl * FX_CS + ml) =
                f32x4{acc[mi][ni][4 * q4], acc[mi][ni][4 * q4 + 1], acc[mi][ni][4 * q4 + 2], acc[mi][ni][4 * q4 + 3]};
          }
    }
    __syncthreads();
    {
      const int rl = tid & 63, row = r0 + slab * 64 + rl;
      int b, l;
      if (row_valid(row, Lp1, ndata, &b, &l)) {
        for (int ml = tid >> 6; ml < MT; ml += 8) {
          const int m = m0 + ml;
          if (m >= a.M) break;
          a.y_ncl[((long)b * a.M + m) * a.L + l] = (Cs[rl * FX_CS + ml] + (a.bias ? a.bias[m] : 0.f)) * oscale;
        }
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------ boundary conversions
// Sticky range flag of the format (one int per device, in the code object's own data segment -- not an allocation):
// bit 0 = a value of magnitude >= 65504 (fp16's largest finite value; it was stored saturated), bit 1 = a NaN, seen by a
// conversion INTO the format (model inputs, gradients entering a backward chain); bit 2 = a value PRODUCED inside a chain
// (an activation, or a loss-scaled gradient that outgrew the 2^8 headroom) reached the limit or is a NaN -- watched by
// the epilogues of the f16mx and fp16 convolutions.  Read / cleared by alvq_f16mx_range_flag.
// Round 4: the flag has two words.  g_fx_range_flag collects the bits raised since the last GUARDED optimiser advance
// (alvq_adam_advance_f32 with a skip slot: the start of a Trainer step), so that a step's last launch can turn "did THIS
// step saturate" into the skip slot of the flat gradient buffer (alvq_range_flag_to_slot); the advance folds the bits into
// g_fx_range_sticky.  alvq_f16mx_range_flag reports / clears the union.
__device__ int g_fx_range_flag = 0;
__device__ int g_fx_range_sticky = 0;

int* fx_range_flag_ptr() {
  static int* cache[64] = {};
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) d = 0;
  if (!cache[d]) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_fx_range_flag)) == hipSuccess) cache[d] = (int*)p;
  }
  return cache[d];
}

int* fx_range_sticky_ptr() {
  static int* cache[64] = {};
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) d = 0;
  if (!cache[d]) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_fx_range_sticky)) == hipSuccess) cache[d] = (int*)p;
  }
  return cache[d];
}

__global__ void fx_range_flag_kernel(int* out, int reset) {
  *out = g_fx_range_flag | g_fx_range_sticky;
  if (reset) { g_fx_range_flag = 0; g_fx_range_sticky = 0; }
}

// (B,C,L) fp32 -> f16mx NLC planes, optionally multiplied by a device scalar (the loss scale of a backward chain)
__global__ __launch_bounds__(256) void ncl_to_nlc_fx_kernel(const float* x, u16* y, long plane, int B, int C, int L, int Cp,
                                                            int rows_total, int e, const float* scale) {
  __shared__ float tile[32][33];
  fx_saturating_conversions();
  int range = 0;
  const int ct = Cp / 32;
  const int r0 = (blockIdx.x / ct) * 32, c0 = (blockIdx.x % ct) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int Lp1 = L + 1, ndata = B * Lp1;
  const float sc = scale ? *scale : 1.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, row = r0 + tx;
    int b, l;
    const bool ok = row_valid(row, Lp1, ndata, &b, &l) && c < C;
    const float v = ok ? x[((long)b * C + c) * L + l] * sc : 0.f;
    tile[ty + 8 * i][tx] = v;
    range |= (fabsf(v) >= 65504.f ? 1 : 0) | (v != v ? 2 : 0);
  }
  if (__any(range)) {                       // rare: one atomic per wave that saw such a value
    for (int o = 32; o > 0; o >>= 1) range |= __shfl_xor(range, o, 64);
    if ((threadIdx.x & 63) == 0) atomicOr(&g_fx_range_flag, range);
  }
  __syncthreads();
  // thread (row = tid >> 3, 4 channels at (tid & 7) * 4): 8 bytes of H, 4 of hi8, 4 of lo8
  const int rr = threadIdx.x >> 3, cq = (threadIdx.x & 7) * 4;
  const int row = r0 + rr;
  if (row < rows_total) {
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = tile[cq + k][rr];
    unsigned h[2], qh[1], ql[1];
    fx_split<4>(v, fx_pow2(e), fx_pow2(e - FX_LO_SHIFT), h, qh, ql);
    *(u32x2*)(y + (long)row * Cp + c0 + cq) = u32x2{h[0], h[1]};
    unsigned char* q = (unsigned char*)(y + plane) + (long)row * Cp * 2 + fx_q_off(c0) + cq;
    *(unsigned*)q = qh[0];
    *(unsigned*)(q + 32) = ql[0];
  }
}

__global__ __launch_bounds__(256) void nlc_to_ncl_fx_kernel(const u16* x, long plane, float* y, int B, int C, int L, int Cp,
                                                            int rows_total, int e, const float* scale) {
  __shared__ float tile[32][33];
  const int ct = Cp / 32;
  const int r0 = (blockIdx.x / ct) * 32, c0 = (blockIdx.x % ct) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int Lp1 = L + 1, ndata = B * Lp1;
  const float sc = scale ? *scale : 1.f, s_lo = fx_pow2(e - FX_LO_SHIFT);
  const int rr = threadIdx.x >> 3, cq = (threadIdx.x & 7) * 4;
  const int rowq = r0 + rr;
  if (rowq < rows_total) {
    const u32x2 h = *(const u32x2*)(x + (long)rowq * Cp + c0 + cq);
    const unsigned ql = *(const unsigned*)((const unsigned char*)(x + plane) + (long)rowq * Cp * 2 + fx_q_off(c0) + cq + 32);
    float v0, v1, v2, v3;
    fx_join2(h[0], ql, 0, s_lo, v0, v1);
    fx_join2(h[1], ql, 2, s_lo, v2, v3);
    tile[rr][cq] = v0 * sc;
    tile[rr][cq + 1] = v1 * sc;
    tile[rr][cq + 2] = v2 * sc;
    tile[rr][cq + 3] = v3 * sc;
  } else {
    tile[rr][cq] = tile[rr][cq + 1] = tile[rr][cq + 2] = tile[rr][cq + 3] = 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, row = r0 + tx;
    int b, l;
    if (row_valid(row, Lp1, ndata, &b, &l) && c < C) y[((long)b * C + c) * L + l] = tile[tx][ty + 8 * i];
  }
}

// out = t > 0 ? dy : 0 on all three parts (16 channels per thread step; the sign of a value is the sign of its H part)
__global__ __launch_bounds__(256) void relu_mask_fx_kernel(const u16* dy, const u16* t, u16* out, long plane, int rows, int Cp) {
  const long n16 = (long)rows * Cp / 16;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) {
    const long row = (i * 16) / Cp;
    const int c = (int)((i * 16) % Cp);
    const long ro = row * Cp;
    const u32x4 m0 = *(const u32x4*)(t + ro + c), m1 = *(const u32x4*)(t + ro + c + 8);
    u32x4 h0 = *(const u32x4*)(dy + ro + c), h1 = *(const u32x4*)(dy + ro + c + 8);
    const unsigned char* qs = (const unsigned char*)(dy + plane) + ro * 2 + fx_q_off(c);
    u32x4 qh = *(const u32x4*)qs, ql = *(const u32x4*)(qs + 32);
    unsigned keep = 0;   // bit k: channel c + k survives
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      keep |= (fx_h2f_lo(m0[e]) > 0.f ? 1u : 0u) << (2 * e) | (fx_h2f_hi(m0[e]) > 0.f ? 1u : 0u) << (2 * e + 1);
      keep |= (fx_h2f_lo(m1[e]) > 0.f ? 1u : 0u) << (8 + 2 * e) | (fx_h2f_hi(m1[e]) > 0.f ? 1u : 0u) << (9 + 2 * e);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const unsigned k0 = (keep >> (2 * e)) & 3u, k1 = (keep >> (8 + 2 * e)) & 3u;
      h0[e] &= ((k0 & 1u) ? 0xffffu : 0u) | ((k0 & 2u) ? 0xffff0000u : 0u);
      h1[e] &= ((k1 & 1u) ? 0xffffu : 0u) | ((k1 & 2u) ? 0xffff0000u : 0u);
      const unsigned kb = (keep >> (4 * e)) & 15u;
      const unsigned bm = ((kb & 1u) ? 0xffu : 0u) | ((kb & 2u) ? 0xff00u : 0u) | ((kb & 4u) ? 0xff0000u : 0u) | ((kb & 8u) ? 0xff000000u : 0u);
      qh[e] &= bm;
      ql[e] &= bm;
    }
    *(u32x4*)(out + ro + c) = h0;
    *(u32x4*)(out + ro + c + 8) = h1;
    unsigned char* qd = (unsigned char*)(out + plane) + ro * 2 + fx_q_off(c);
    *(u32x4*)qd = qh;
    *(u32x4*)(qd + 32) = ql;
  }
}

// Loss scale of a backward chain, chosen on the device: state = {S, 1/S, amax bits, -}.  Pass 1 folds |x| into state[2]
// (integer max of the float bits: order-independent, exact); pass 2 turns it into the power of two that puts amax at
// 2^8 -- 2^8 of headroom below fp16's 65504 for growth along the chain, 2^22 above its smallest normal -- and rearms.
__global__ __launch_bounds__(256) void grad_amax_kernel(const float* x, long n, float* state) {
  float m = 0.f;
  const long n4 = n >> 2, stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {      // four independent 16-byte loads in flight per thread
    const f32x4 a = ((const f32x4*)x)[i], b = ((const f32x4*)x)[i + stride], c = ((const f32x4*)x)[i + 2 * stride],
                d = ((const f32x4*)x)[i + 3 * stride];
#pragma unroll
    for (int k = 0; k < 4; ++k) m = fmaxf(fmaxf(m, fmaxf(fabsf(a[k]), fabsf(b[k]))), fmaxf(fabsf(c[k]), fabsf(d[k])));
  }
  for (; i < n4; i += stride) {
    const f32x4 v = ((const f32x4*)x)[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
  }
  for (long j = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; j < n; j += stride) m = fmaxf(m, fabsf(x[j]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  // one atomic per workgroup (thousands of same-address atomics serialise at the L2: they, not the 26 MB read, set the
  // 51 us this pass took with one atomic per wave on 2048 workgroups)
  __shared__ float wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
    if (m > 0.f) atomicMax((unsigned*)state + 2, __float_as_uint(m));
  }
}
__global__ void grad_scale_kernel(float* state) {
  const unsigned bits = ((unsigned*)state)[2];
  const float amax = __uint_as_float(bits);
  int e = 127;                                        // S = 1 when the gradient is all zero / not finite
  if (amax > 0.f && amax < 3.0e38f) {
    const int ea = (int)((bits >> 23) & 0xff);        // amax in [2^(ea-127), 2^(ea-126))
    e = 127 + 8 - (ea - 127) - 1;                     // S * amax in [2^7, 2^8)
    e = e < 1 ? 1 : (e > 253 ? 253 : e);
  }
  state[0] = fx_pow2(e);
  state[1] = fx_pow2(254 - e);
  ((unsigned*)state)[2] = 0u;
}

}  // namespace alvq

using namespace alvq;

static inline int pad_to(int x, int q) { return (x + q - 1) / q * q; }
static inline long nlc_plane_elems(int B, int L, int C) {
  return ((long)alvq_nlc_rows(B, L) + 2L * alvq_nlc_guard_rows()) * pad_to(C, 64);
}

extern "C" int alvq_ncl_to_nlc_f16mx(const float* x, void* y, int B, int C, int L, const float* scale, void* stream) {
  ALVQ_REQUIRE(x && y, ALVQ_EINVAL, "alvq_ncl_to_nlc_f16mx: null pointer");
  ALVQ_REQUIRE(B > 0 && C > 0 && L > 0, ALVQ_EINVAL, "alvq_ncl_to_nlc_f16mx: bad dims");
  const int Cp = pad_to(C, 64), rows = (int)alvq_nlc_rows(B, L);
  hipLaunchKernelGGL(ncl_to_nlc_fx_kernel, dim3((rows / 32) * (Cp / 32)), dim3(256), 0, (hipStream_t)stream, x, (u16*)y,
                     nlc_plane_elems(B, L, C), B, C, L, Cp, rows, FX_E_ACT, scale);
  return check_launch("alvq_ncl_to_nlc_f16mx");
}

extern "C" int alvq_nlc_to_ncl_f16mx(const void* x, float* y, int B, int C, int L, const float* scale, void* stream) {
  ALVQ_REQUIRE(x && y, ALVQ_EINVAL, "alvq_nlc_to_ncl_f16mx: null pointer");
  ALVQ_REQUIRE(B > 0 && C > 0 && L > 0, ALVQ_EINVAL, "alvq_nlc_to_ncl_f16mx: bad dims");
  const int Cp = pad_to(C, 64), rows = (int)alvq_nlc_rows(B, L);
  hipLaunchKernelGGL(nlc_to_ncl_fx_kernel, dim3((rows / 32) * (Cp / 32)), dim3(256), 0, (hipStream_t)stream, (const u16*)x,
                     nlc_plane_elems(B, L, C), y, B, C, L, Cp, rows, FX_E_ACT, scale);
  return check_launch("alvq_nlc_to_ncl_f16mx");
}

extern "C" int alvq_f16mx_range_flag(int* out, int reset, void* stream) {
  ALVQ_REQUIRE(out, ALVQ_EINVAL, "alvq_f16mx_range_flag: null pointer");
  hipLaunchKernelGGL(fx_range_flag_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, out, reset);
  return check_launch("alvq_f16mx_range_flag");
}

extern "C" int alvq_relu_mask_f16mx(const void* dy, const void* t, void* out, int B, int C, int L, void* stream) {
  ALVQ_REQUIRE(dy && t && out, ALVQ_EINVAL, "alvq_relu_mask_f16mx: null pointer");
  ALVQ_REQUIRE(B > 0 && C > 0 && L > 0, ALVQ_EINVAL, "alvq_relu_mask_f16mx: bad dims");
  const int Cp = pad_to(C, 64), rows = (int)alvq_nlc_rows(B, L);
  long gq = ((long)rows * Cp / 16 + 255) / 256;
  if (gq > 2048) gq = 2048;
  hipLaunchKernelGGL(relu_mask_fx_kernel, dim3((int)gq), dim3(256), 0, (hipStream_t)stream, (const u16*)dy, (const u16*)t, (u16*)out,
                     nlc_plane_elems(B, L, C), rows, Cp);
  return check_launch("alvq_relu_mask_f16mx");
}

extern "C" int alvq_grad_scale_f32(const float* x, int64_t n, float* state, void* stream) {
  ALVQ_REQUIRE(x && state, ALVQ_EINVAL, "alvq_grad_scale_f32: null pointer");
  ALVQ_REQUIRE(n > 0, ALVQ_EINVAL, "alvq_grad_scale_f32: n <= 0");
  ALVQ_REQUIRE(((uintptr_t)x & 15) == 0, ALVQ_EINVAL, "alvq_grad_scale_f32: x must be 16-byte aligned");
  long gq = (n / 4 + 256 * 4 - 1) / (256 * 4);
  if (gq > 512) gq = 512;
  if (gq < 1) gq = 1;
  hipLaunchKernelGGL(grad_amax_kernel, dim3((int)gq), dim3(256), 0, (hipStream_t)stream, x, (long)n, state);
  hipLaunchKernelGGL(grad_scale_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state);
  return check_launch("alvq_grad_scale_f32");
}

extern "C" int alvq_conv1d_f16mx(const void* x, const void* wp, const float* bias, const void* skip1, const void* skip2,
                                 const void* mask, const void* post, void* y, void* y2, float* y_ncl, int B, int C, int M, int L,
                                 int KW, int relu, const void* mask_bits, void* relu_bits_out, const float* out_scale, void* stream) {
  ALVQ_REQUIRE(x && wp && (y || y_ncl), ALVQ_EINVAL, "alvq_conv1d_f16mx: null x/wp/y");
  ALVQ_REQUIRE(!(y && y_ncl), ALVQ_EINVAL, "alvq_conv1d_f16mx: choose one of y (NLC) and y_ncl (NCL fp32)");
  ALVQ_REQUIRE(B > 0 && C > 0 && M > 0 && L > 0, ALVQ_EINVAL, "alvq_conv1d_f16mx: bad dims");
  ALVQ_REQUIRE(KW == 1 || KW == 3, ALVQ_EUNSUPPORTED, "alvq_conv1d_f16mx: KW=%d (only 1 and 3)", KW);
  ALVQ_REQUIRE((y2 == nullptr) == (post == nullptr), ALVQ_EINVAL, "alvq_conv1d_f16mx: y2 and post go together");
  ALVQ_REQUIRE(!y_ncl || (!skip1 && !skip2 && !mask && !post && !relu), ALVQ_EUNSUPPORTED,
               "alvq_conv1d_f16mx: the NCL fp32 epilogue fuses bias (and the output scale) only");
  ALVQ_REQUIRE((long)B * (L + 1) < (1L << 30), ALVQ_EUNSUPPORTED, "alvq_conv1d_f16mx: problem too large");
  ALVQ_REQUIRE(!(mask && mask_bits), ALVQ_EINVAL, "alvq_conv1d_f16mx: pass the mask as a tensor or as bits, not both");
  ALVQ_REQUIRE(!y_ncl || (!mask_bits && !relu_bits_out), ALVQ_EUNSUPPORTED, "alvq_conv1d_f16mx: sign bits go with the NLC output");
  const long rows = alvq_nlc_rows(B, L);
  ConvFxArgs a{{(const u16*)x, (const u16*)wp, bias, (const u16*)skip1, (const u16*)skip2, (const u16*)mask, (const u16*)post,
                (u16*)y, (u16*)y2, y_ncl, B, L, pad_to(C, 64), M, pad_to(M, 64), pad_to(M, WP_ROWS), relu ? 1 : 0,
                (int)(rows / FX_R), pad_to(M, FX_M) / FX_M,   /* rtiles: see below */
                (const unsigned char*)mask_bits, (unsigned char*)relu_bits_out},
               nlc_plane_elems(B, L, C), (long)alvq_packed_weight_elems(M, C, KW), nlc_plane_elems(B, L, M),
               FX_E_W, FX_E_ACT, out_scale, fx_range_flag_ptr(), nullptr, 0};
#ifdef ALVQ_DEBUG_KERNELS   // the ablation / phase-stamp instantiations exist only in the debug library (build.py --debug-kernels)
  static const int dbg_env = getenv("ALVQ_FX_DBG") ? atoi(getenv("ALVQ_FX_DBG")) : 0;   // timing ablations (results are garbage)
#else
  constexpr int dbg_env = 0;
#endif
  a.dbg = dbg_env;
  hipStream_t s = (hipStream_t)stream;
  static DeviceOnce attr;
  if (attr.need()) {
    (void)hipFuncSetAttribute((const void*)conv1d_f16mx_kernel<0, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, FX_LDS);
    (void)hipFuncSetAttribute((const void*)conv1d_f16mx_kernel<1, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, FX_LDS);
    (void)hipFuncSetAttribute((const void*)conv1d_f16mx_kernel<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, FX_LDS);
    (void)hipFuncSetAttribute((const void*)conv1d_f16mx_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, FX_LDS);
    (void)hipFuncSetAttribute((const void*)conv1d_f16mx_kernel<0, 3, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, FX_LDS);
    (void)hipFuncSetAttribute((const void*)conv1d_f16mx_kernel<1, 3, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, FX_LDS);
    (void)hipFuncSetAttribute((const void*)conv1d_f16mx_kernel<0, 1, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, FX_LDS);
    (void)hipFuncSetAttribute((const void*)conv1d_f16mx_kernel<1, 1, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, FX_LDS);
    (void)hipFuncSetAttribute((const void*)conv1d_f16mx_kernel<1, 3, 0, 1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, FX_LDS);
    (void)hipFuncSetAttribute((const void*)conv1d_f16mx_kernel<1, 1, 0, 1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, FX_LDS);
#ifdef ALVQ_DEBUG_KERNELS
    (void)hipFuncSetAttribute((const void*)conv1d_f16mx_kernel<0, 3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, FX_LDS);
    (void)hipFuncSetAttribute((const void*)conv1d_f16mx_kernel<0, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, FX_LDS);
    (void)hipFuncSetAttribute((const void*)conv1d_f16mx_kernel<0, 3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, FX_LDS);
    (void)hipFuncSetAttribute((const void*)conv1d_f16mx_kernel<0, 3, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, FX_LDS);
#endif
  }
  // fewer than ~3/4 of the CUs covered by 256-row tiles (one m-tile: M <= 256; short batches): 128-row tiles
  // (option "fx_rows" = 128 | 256 forces one of the two -- the tests run every shape through both)
  const int forced = (int)option(OPT_FX_ROWS);
  // fp32-NCL output of at most 128 channels (the pre-VQ convolution and the data gradient that leaves the decoder): a
  // 128-channel m-tile on 128-row tiles -- no MFMA spent on padding channels (option "fx_narrow" = 0 switches it off)
  const bool narrow = !dbg_env && y_ncl && M <= 128 && option(OPT_FX_NARROW) != 0;
  const bool half = narrow || (!dbg_env && (forced == 128 || (forced != 256 && a.b.rtiles * a.b.mtiles < 192)));
  if (half) a.b.rtiles = (int)(rows / 128);
  const dim3 grid(a.b.rtiles * a.b.mtiles), block(512);
#ifdef ALVQ_DEBUG_KERNELS
  if (dbg_env && y) {
    static unsigned long long* stamps = nullptr;
    if ((dbg_env & (256 | 512)) && !stamps) (void)hipMalloc(&stamps, 4096 * 4 * sizeof(unsigned long long));
    a.stamps = stamps;
    if (grid.x > 4096) a.dbg &= ~(256 | 512);
    if (KW == 3 && (dbg_env & 16)) hipLaunchKernelGGL((conv1d_f16mx_kernel<0, 3, 2>), grid, block, FX_LDS, s, a);
    else if (KW == 3 && (dbg_env & 32)) hipLaunchKernelGGL((conv1d_f16mx_kernel<0, 3, 3>), grid, block, FX_LDS, s, a);
    else if (KW == 3) hipLaunchKernelGGL((conv1d_f16mx_kernel<0, 3, 1>), grid, block, FX_LDS, s, a);
    else hipLaunchKernelGGL((conv1d_f16mx_kernel<0, 1, 1>), grid, block, FX_LDS, s, a);
    if (a.dbg & (256 | 512)) {     // phase durations of this launch (means over its workgroups; first / last round by start time)
      (void)hipStreamSynchronize(s);
      static unsigned long long h[4096 * 4];
      (void)hipMemcpy(h, stamps, grid.x * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      unsigned long long t0 = ~0ull, t1 = 0;
      for (unsigned i = 0; i < grid.x; ++i) { if (h[4 * i] < t0) t0 = h[4 * i]; if (h[4 * i + 3] > t1) t1 = h[4 * i + 3]; }
      double pro[2] = {0, 0}, mainl[2] = {0, 0}, epi[2] = {0, 0}, start[2] = {0, 0}; int cnt[2] = {0, 0};
      for (unsigned i = 0; i < grid.x; ++i) {
        const int rnd = (h[4 * i] - t0) * 4 > (t1 - t0) ? 1 : 0;     // started in the first quarter of the launch or later
        pro[rnd] += h[4 * i + 1] - h[4 * i]; mainl[rnd] += h[4 * i + 2] - h[4 * i + 1]; epi[rnd] += h[4 * i + 3] - h[4 * i + 2];
        start[rnd] += h[4 * i] - t0; ++cnt[rnd];
      }
      for (int r = 0; r < 2; ++r)
        if (cnt[r])
          fprintf(stderr, "[fx stamps] KW=%d M=%d C=%d round %d: %d workgroups, start +%.1f us, staging %.2f us, main loop %.2f us, epilogue %.2f us\n",
                  KW, M, C, r, cnt[r], start[r] / cnt[r] / 100.0, pro[r] / cnt[r] / 100.0, mainl[r] / cnt[r] / 100.0, epi[r] / cnt[r] / 100.0);
      fprintf(stderr, "[fx stamps] launch span %.1f us\n", (t1 - t0) / 100.0);
    }
    return check_launch("alvq_conv1d_f16mx(dbg)");
  }
#endif
  if (narrow) {
    if (KW == 3) hipLaunchKernelGGL((conv1d_f16mx_kernel<1, 3, 0, 1, 2>), grid, block, FX_LDS, s, a);
    else hipLaunchKernelGGL((conv1d_f16mx_kernel<1, 1, 0, 1, 2>), grid, block, FX_LDS, s, a);
  } else if (half) {
    if (y) {
      if (KW == 3) hipLaunchKernelGGL((conv1d_f16mx_kernel<0, 3, 0, 1>), grid, block, FX_LDS, s, a);
      else hipLaunchKernelGGL((conv1d_f16mx_kernel<0, 1, 0, 1>), grid, block, FX_LDS, s, a);
    } else {
      if (KW == 3) hipLaunchKernelGGL((conv1d_f16mx_kernel<1, 3, 0, 1>), grid, block, FX_LDS, s, a);
      else hipLaunchKernelGGL((conv1d_f16mx_kernel<1, 1, 0, 1>), grid, block, FX_LDS, s, a);
    }
  } else if (y) {
    if (KW == 3) hipLaunchKernelGGL((conv1d_f16mx_kernel<0, 3>), grid, block, FX_LDS, s, a);
    else hipLaunchKernelGGL((conv1d_f16mx_kernel<0, 1>), grid, block, FX_LDS, s, a);
  } else {
    if (KW == 3) hipLaunchKernelGGL((conv1d_f16mx_kernel<1, 3>), grid, block, FX_LDS, s, a);
    else hipLaunchKernelGGL((conv1d_f16mx_kernel<1, 1>), grid, block, FX_LDS, s, a);
  }
  return check_launch("alvq_conv1d_f16mx");
}
