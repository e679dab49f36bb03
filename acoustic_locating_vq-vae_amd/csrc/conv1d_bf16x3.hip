// Split-bf16 ("bf16x3") convolution and weight-gradient: fp32-grade results on the bf16 matrix cores.
//
// gfx950 has no TF32/xf32 MFMA and its exact-fp32 MFMA runs at 1/16 of the bf16 rate.  Here every fp32 value v is
// carried as two bf16 planes, hi = bf16(v) and lo = bf16(v - hi) (|v - hi - lo| <= 2^-18 |v|), and every product
// is evaluated as  hi*hi + hi*lo + lo*hi  (three bf16 MFMAs, fp32 accumulation; the lo*lo term, <= 2^-18 relative,
// is dropped).  bf16 x bf16 products are exact in fp32, so the result differs from an fp32 computation only by
// ~1e-5 relative -- well inside the 1e-3 parity bar -- at one third of the bf16 MFMA rate, i.e. ~5x the exact-fp32
// MFMA peak.  Storage cost equals fp32 (two bf16 planes per tensor).
//
// Layout: the NLC-padded matrix of include/alvq.h, with the lo plane stored right after the hi plane (guard rows
// included): lo = hi + alvq_nlc_plane_bytes(B, L, C).  Packed weights likewise (lo image after the hi image).
//
// Kernels: same tiling as conv1d_bf16_v2 / conv1d_wgrad_bf16_v2 (256x256 tile, 8 waves; 128x128x3-tap tile for the
// weight-gradient) with a 2-stage LDS-DMA ring (a K-tile now carries 4 slabs and 96 MFMAs per wave, so one
// iteration of look-ahead already gives the DMA ~3000 cycles).
#include <stdlib.h>

#include "alvq_common.h"
#include "bf16_common.h"
#include "wgrad_reduce.h"

namespace alvq {

constexpr int X3_M = 256, X3_R = 256, X3_K = 32;
constexpr int X3_SLAB = X3_M * X3_K * 2;          // 16384 B
constexpr int X3_CS = X3_M + 4;

struct ConvX3Args {
  ConvBArgs b;          // hi planes (and everything shared)
  long x_plane, wp_plane, y_plane;   // element offsets (u16) from a hi pointer to its lo plane
};

__device__ __forceinline__ void split2(float v, u16& hi, u16& lo) {
  hi = f2bf(v);
  lo = f2bf(v - bf2f(hi));
}

// 8 consecutive channels of one row -> packed hi and lo words (hardware bf16 rounding)
__device__ __forceinline__ void split_pack8(const float (&v)[8], u32x4& hi, u32x4& lo) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    hi[e] = f2bf_pk(v[2 * e], v[2 * e + 1]);
    lo[e] = f2bf_pk(v[2 * e] - __uint_as_float(hi[e] << 16), v[2 * e + 1] - __uint_as_float(hi[e] & 0xffff0000u));
  }
}

// Register-direct epilogue of one wave's 128 (m) x 64 (rows) block, split-bf16 flavour of wave_epilogue_bf16
// (conv1d_bf16_tile256.h): v_permlane16_swap gives every lane 8 consecutive channels, 16-byte loads and stores.  Written,
// like that one, for few VALU instructions per value: a row block's offset is formed once, the skip / mask operands of
// the whole row block are requested before the first group is finished, nothing is computed for absent operands, gap
// rows are zeroed by a select on the packed words inside a wave-uniform branch.
template <int NNI>
__device__ __forceinline__ void wave_epilogue_x3(const ConvX3Args& ax, const f32x4 (&acc)[8][NNI], int m0, int r0, int li,
                                                 int kq, int wm0, int wn0) {
  const ConvBArgs& a = ax.b;
  const int Lp1 = a.L + 1, ndata = a.B * Lp1;
  const int mb0 = m0 + wm0 + (kq & 1) * 16 + (kq >> 1) * 8;
  const long pl = ax.y_plane;
#pragma unroll
  for (int ni = 0; ni < NNI; ++ni) {
    const int row = r0 + wn0 + ni * 16 + li;
    int b, l;
    const bool ok = row_valid(row, Lp1, ndata, &b, &l);
    const bool gaps = !__all(ok);
    const long o0 = (long)row * a.Mop + mb0;
    u16x8 s1h[4], s1l[4], mk[4];
#pragma unroll
    for (int mp = 0; mp < 8; mp += 2) {
      if (m0 + wm0 + mp * 16 >= a.Mop) continue;
      if (a.skip1) {
        s1h[mp / 2] = *(const u16x8*)(a.skip1 + o0 + mp * 16);
        s1l[mp / 2] = *(const u16x8*)(a.skip1 + pl + o0 + mp * 16);
      }
      if (a.mask) mk[mp / 2] = *(const u16x8*)(a.mask + o0 + mp * 16);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mp = 0; mp < 8; mp += 2) {
      if (m0 + wm0 + mp * 16 >= a.Mop) continue;
      const long o = o0 + mp * 16;
      const int mb = mb0 + mp * 16;
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
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
        for (int e = 0; e < 8; ++e) v[e] += bf2f(s1h[mp / 2][e]) + bf2f(s1l[mp / 2][e]);
      }
      if (a.skip2) {
        const u16x8 sh = *(const u16x8*)(a.skip2 + o), sl = *(const u16x8*)(a.skip2 + pl + o);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bf2f(sh[e]) + bf2f(sl[e]);
      }
      if (a.relu & 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      if (a.mask) {   // sign of a split value is the sign of its hi plane
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = bf2f(mk[mp / 2][e]) > 0.f ? v[e] : 0.f;
      }
      u32x4 hi, lo;
      split_pack8(v, hi, lo);
      if (gaps) {                                           // gap / tail rows stay zero in both planes
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          hi[e] = ok ? hi[e] : 0u;
          lo[e] = ok ? lo[e] : 0u;
        }
      }
      *(u32x4*)(a.y + o) = hi;
      *(u32x4*)(a.y + pl + o) = lo;
      if (a.y2) {
        const u16x8 ph = *(const u16x8*)(a.post + o), pq = *(const u16x8*)(a.post + pl + o);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bf2f(ph[e]) + bf2f(pq[e]);
        split_pack8(v, hi, lo);
        if (gaps) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            hi[e] = ok ? hi[e] : 0u;
            lo[e] = ok ? lo[e] : 0u;
          }
        }
        *(u32x4*)(a.y2 + o) = hi;
        *(u32x4*)(a.y2 + pl + o) = lo;
      }
    }
  }
}

// Main kernel.  A K-tile = (32 channels, one tap).  Two LDS-DMA rings: the WEIGHT slabs of a K-tile (W hi, W lo; 32 KB, two
// stages) and the ACTIVATION slabs of a CHUNK of 32 channels (X hi, X lo; rows r0-PAD .. r0+255+PAD staged once, 34 KB,
// two stages) -- tap t reads the slab t rows further down, as in conv1d_bf16_k3.hip, so a width-3 layer moves a third
// less through LDS-DMA.  Per wave a K-tile is three phases of 32 MFMAs on the same 8 x 4 accumulators:
//   phase 1  hi*hi : A0 = W hi fragments, BX = X hi        | meanwhile: read X lo -> BY, first half of W lo -> A1
//   phase 2  hi*lo : A0, BY                                 | meanwhile: second half of W lo -> A1
//            s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier      <- K-tile t+1 has landed; every read of this K-tile is done
//   phase 3  lo*hi : A1, BX                                 | meanwhile: DMA of K-tile t+2's weights into THIS weight
//                                                             stage (and, on taps 0 / 1, of the next chunk's activation
//                                                             planes); the hi fragments of K-tile t+1 -> A0 and BY
// so the barrier falls between phases whose operands are already in registers: no fragment read is ever exposed
// behind it, every LDS read has at least half a phase (16 MFMAs) to return, and the X fragment sets swap roles from
// one K-tile to the next (BX <-> BY).  96 fragment VGPRs + 128 accumulators.  MFMAs are tied inline asm (hipcc does
// not tie the builtin's destination to its C operand and then shuffles the accumulators through spare registers it
// does not have here); DMA pieces are inline asm with a scalar base and one 32-bit lane offset.
constexpr int X3_WSTAGE = 2 * X3_SLAB;            // W hi, W lo of one K-tile
constexpr int X3_XSLAB = 272 * 64;                // 272 rows x 64 B (258 used)
constexpr int X3_XSTAGE = 2 * X3_XSLAB;           // X hi, X lo of one chunk
constexpr int X3_LDS2 = 2 * X3_WSTAGE + 2 * X3_XSTAGE;   // 135168 B
static_assert(64 * X3_CS * 4 <= X3_LDS2, "C slab must fit");

// NNI: 16-row fragments per wave.  4 = the 256 x 256 tile (a wave owns 128 channels x 64 rows); 2 (round 4) = a 128-channel
// m-tile x 256 rows: every wave owns all 128 channels x 32 rows, waves 0-3 stage the 128 weight rows.  For layers of at most
// 128 output channels -- the pre-VQ convolution's 256-wide tile spent half of its MFMAs (171 us of the default mode's step) on
// padding channels -- and for problems with too few 256 x 256 tiles to cover the chip (the RIR config's 1024-channel layers:
// 104 tiles on 256 CUs -> 208 workgroups of half the work).  1 = 128 channels x 128 rows (a wave owns 128 x 16; waves 0-3 stage
// the activation rows as well): twice the workgroups again for the launches that still leave CUs idle (the speech pre-VQ
// convolution: 126 -> 251 workgroups).  Same K order per output: results are bit-identical.
template <int OUT, int KW, int NNI = 4>
__global__ __launch_bounds__(512, 2) void conv1d_bf16x3_kernel(ConvX3Args ax) {
  static_assert(NNI == 4 || NNI == 2 || NNI == 1, "4, 2 or 1 row fragments per wave");
  constexpr int PAD = (KW - 1) / 2;
  constexpr int MT = NNI == 4 ? X3_M : 128;
  constexpr int RT = NNI == 1 ? 128 : X3_R;
  const ConvBArgs& a = ax.b;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const int wm0 = NNI == 4 ? (wave >> 2) * 128 : 0, wn0 = NNI == 4 ? (wave & 3) * 64 : wave * 16 * NNI;

  const int tile = xcd_remap(blockIdx.x, a.mtiles * a.rtiles);
  const int m0 = (tile % a.mtiles) * MT;
  const int r0 = (tile / a.mtiles) * RT;
  const int Cp = a.Cp;

  // ---- DMA: a piece is 16 rows x 64 B; lane i -> row i>>2, slot i&3 <- channel group (i&3) ^ h[(row>>2)&3]
  const int hsel = (lane >> 4) & 3;
  const int hval = (hsel == 0) ? 0 : (4 - hsel);
  const int srow = lane >> 2, sgrp = (lane & 3) ^ hval;
  const unsigned lane_off = (unsigned)(srow * Cp + sgrp * 8) * 2u;
  const long tap_w = (long)a.Mp128 * Cp * 2;                   // bytes per tap of packed weights
  const long row16 = (long)Cp * 32;                            // bytes per 16 rows
  const long wpl = ax.wp_plane * 2, xpl = ax.x_plane * 2;      // hi -> lo plane, bytes
  const unsigned lds0 = (unsigned)(unsigned long)((__attribute__((address_space(3))) unsigned char*)lds);
  auto dma = [&](const char* sbase, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_off), "s"(sbase), "s"(lds_dst)
                 : "memory");
  };
  const char* const wb = (const char*)(a.wp + ((long)m0 + wave * 32) * Cp);
  const char* const xb = (const char*)(a.x + ((long)r0 - PAD + wave * 32) * Cp);
  constexpr unsigned XBASE = 2 * X3_WSTAGE;
  auto issueW = [&](int t) {   // K-tile t -> weight stage t & 1: 32 rows of W hi and of W lo per wave
    if (wave * 32 >= MT) return;       // 128-channel m-tile: waves 0-3 stage the weight rows
    const int chunk = t / KW, tap = t - chunk * KW;
    const unsigned dst = lds0 + (t & 1) * X3_WSTAGE + wave * 2048;
    const char* ws = wb + tap * tap_w + chunk * (X3_K * 2);
    dma(ws, dst);
    dma(ws + row16, dst + 1024);
    dma(ws + wpl, dst + X3_SLAB);
    dma(ws + wpl + row16, dst + X3_SLAB + 1024);
  };
  auto issueX = [&](int chunk, int plane) {   // one plane of a chunk's activation slab -> activation stage chunk & 1
    if (wave * 32 >= RT) return;       // 128-row tile: waves 0-3 stage the activation rows
    const unsigned dst = lds0 + XBASE + (chunk & 1) * X3_XSTAGE + plane * X3_XSLAB + wave * 2048;
    const char* xs = xb + plane * xpl + chunk * (X3_K * 2);
    dma(xs, dst);
    dma(xs + row16, dst + 1024);
    if (KW == 3 && wave == RT / 32 - 1 && srow < 2) dma(xs + 2 * row16, dst + 2048);   // halo: slab rows RT, RT + 1
  };

  // ---- fragment reads (plane 0 = hi, 1 = lo); activations of tap t: slab row = local row + t
  const int hl = (li >> 2) & 3;
  const int loffA = li * 64 + ((kq ^ (hl == 0 ? 0 : 4 - hl)) << 4);
  int loffB[KW];
#pragma unroll
  for (int t = 0; t < KW; ++t) {
    const int r = li + t, h = (r >> 2) & 3;
    loffB[t] = r * 64 + ((kq ^ (h == 0 ? 0 : 4 - h)) << 4);
  }
  const unsigned char* const abase = lds + wm0 * 64 + loffA;
  const unsigned char* const bbase = lds + XBASE + wn0 * 64;
  bf16x8_t fa0[8], fa1[8], fb0[NNI], fb1[NNI];
#define X3_RDA(DST, HALF, WS, PLANE)                                                               \
  {                                                                                                \
    const unsigned char* pa_ = abase + (WS) * X3_WSTAGE + (PLANE) * X3_SLAB;                       \
    _Pragma("unroll") for (int mi = (HALF) * 4; mi < (HALF) * 4 + 4; ++mi) DST[mi] = *(const bf16x8_t*)(pa_ + mi * 1024); \
  }
#define X3_RDB(DST, XS, TAP, PLANE)                                                                \
  {                                                                                                \
    const unsigned char* pb_ = bbase + (XS) * X3_XSTAGE + (PLANE) * X3_XSLAB + loffB[TAP];         \
    _Pragma("unroll") for (int ni = 0; ni < NNI; ++ni) DST[ni] = *(const bf16x8_t*)(pb_ + ni * 1024); \
  }

  f32x4 acc[8][NNI];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < NNI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#define X3_MM(A, B, HALF)                                                                          \
  _Pragma("unroll") for (int mi = (HALF) * 4; mi < (HALF) * 4 + 4; ++mi)                           \
  _Pragma("unroll") for (int ni = 0; ni < NNI; ++ni)                                               \
      asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[mi][ni]) : "v"(A[mi]), "v"(B[ni]));
#define X3_SB __builtin_amdgcn_sched_barrier(0);

  const int nch = Cp / X3_K;        // chunks; even (Cp % 64 == 0)
  const int n = nch * KW;           // K-tiles
  const bool early = wave < 4;      // the two waves of a SIMD issue their DMA at different points of phase 3

  // ---- prologue: chunk 0's activation slabs, K-tiles 0 and 1 (and, for width 1, chunk 1's slabs) staged; hi fragments
  // of K-tile 0 in A0 / fb0
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
  X3_RDA(fa0, 0, 0, 0)
  X3_RDA(fa0, 1, 0, 0)
  X3_RDB(fb0, 0, 0, 0)

  // one K-tile: weights in stage WS, activations in stage XS at tap TAP, X hi in BX; BY receives X lo and, in phase 3,
  // the next K-tile's X hi (NWS, NXS, NTAP).  DMA_ = what this K-tile stages right behind its barrier.
#define X3_TILE(WS, XS, TAP, NWS, NXS, NTAP, BX, BY, DMA_)                                         \
  X3_MM(fa0, BX, 0) X3_SB X3_RDB(BY, XS, TAP, 1) X3_SB                                             \
  X3_MM(fa0, BX, 1) X3_SB X3_RDA(fa1, 0, WS, 1) X3_SB                                              \
  X3_MM(fa0, BY, 0) X3_SB X3_RDA(fa1, 1, WS, 1) X3_SB                                              \
  X3_MM(fa0, BY, 1) X3_SB                                                                          \
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                      \
  __builtin_amdgcn_s_barrier();                                                                    \
  if (early) { DMA_ }                                                                              \
  X3_MM(fa1, BX, 0) X3_SB X3_RDA(fa0, 0, NWS, 0) X3_RDB(BY, NXS, NTAP, 0) X3_SB                    \
  if (!early) { DMA_ }                                                                             \
  X3_MM(fa1, BX, 1) X3_SB X3_RDA(fa0, 1, NWS, 0) X3_SB

  if (KW == 3) {
    // two chunks (six K-tiles) per iteration: every stage index and the BX / BY roles are constants
    for (int c = 0; c < nch; c += 2) {
      const int t = 3 * c;
      X3_TILE(0, 0, 0, 1, 0, 1, fb0, fb1, if (t + 2 < n) issueW(t + 2); if (c + 1 < nch) issueX(c + 1, 0);)
      X3_TILE(1, 0, 1, 0, 0, 2, fb1, fb0, if (t + 3 < n) issueW(t + 3); if (c + 1 < nch) issueX(c + 1, 1);)
      X3_TILE(0, 0, 2, 1, 1, 0, fb0, fb1, if (t + 4 < n) issueW(t + 4);)
      X3_TILE(1, 1, 0, 0, 1, 1, fb1, fb0, if (t + 5 < n) issueW(t + 5); if (c + 2 < nch) issueX(c + 2, 0);)
      X3_TILE(0, 1, 1, 1, 1, 2, fb0, fb1, if (t + 6 < n) issueW(t + 6); if (c + 2 < nch) issueX(c + 2, 1);)
      X3_TILE(1, 1, 2, 0, 0, 0, fb1, fb0, if (t + 7 < n) issueW(t + 7);)
    }
  } else {
    for (int t = 0; t < n; t += 2) {
      X3_TILE(0, 0, 0, 1, 1, 0, fb0, fb1, if (t + 2 < n) { issueW(t + 2); issueX(t + 2, 0); issueX(t + 2, 1); })
      X3_TILE(1, 1, 0, 0, 0, 0, fb1, fb0, if (t + 3 < n) { issueW(t + 3); issueX(t + 3, 0); issueX(t + 3, 1); })
    }
  }
#undef X3_TILE
#undef X3_SB
#undef X3_MM
#undef X3_RDB
#undef X3_RDA
  // the compiler's hazard recogniser does not see inside the asm MFMAs: cover the MFMA-result -> VALU-read wait
  // states by hand before the epilogue touches the accumulators
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

  if constexpr (OUT == 0) {   // bf16 hi + lo planes, straight from the accumulators
    wave_epilogue_x3(ax, acc, m0, r0, li, kq, wm0, wn0);
    return;
  }
  __syncthreads();   // the C slab overlays the stages: the trailing fragment reads of every wave must be done
  // ---- OUT == 1 (fp32 NCL, bias only): four 64-row slabs through an fp32 LDS tile
  float* Cs = (float*)lds;
  const int Lp1 = a.L + 1, ndata = a.B * Lp1;
  for (int slab = 0; slab < RT / 64; ++slab) {
    if ((wn0 >> 6) == slab) {          // the waves that own rows of this 64-row slab (four of them; two / four in the narrow tiles)
#pragma unroll
      for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < NNI; ++ni) {
          const int rl = (wn0 & 63) + ni * 16 + li, ml = wm0 + mi * 16 + kq * 4;
          *(f32x4*)(Cs + rl * X3_CS + ml) = acc[mi][ni];
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
          a.y_ncl[((long)b * a.M + m) * a.L + l] = Cs[rl * X3_CS + ml] + (a.bias ? a.bias[m] : 0.f);
        }
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------- weight-gradient
constexpr int WX_MAXSEG = 4;
struct WgradX3Args {
  // Up to WX_MAXSEG (dy, x) pairs of identical shape whose products are summed into ONE dW (the R uses of a shared
  // residual weight): virtual row v = seg * total_rows + r, as in the bf16 and f16mx weight gradients.
  const u16* dy[WX_MAXSEG];
  const u16* x[WX_MAXSEG];
  int nseg;
  float* partial;
  float* bias_partial;   // [splits][Mp] column sums of dY (the bias gradient), or null
  long dy_plane, x_plane;
  int Mp, Cp, M, C;
  int mtiles, ctiles, splits, chunks_per_split, total_rows;
};

typedef unsigned long long u64x;
template <int ROW_BYTES>
__device__ __forceinline__ void tr_issue(unsigned lds_addr, u64x& lo, u64x& hi) {
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(lds_addr));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(lds_addr), "n"(16 * ROW_BYTES));
}
__device__ __forceinline__ bf16x8_t tr_join(u64x lo, u64x hi) {
  typedef u64x u64x2 __attribute__((ext_vector_type(2)));
  const u64x2 v = {lo, hi};
  return __builtin_bit_cast(bf16x8_t, v);
}

template <int KW, int NCF>
__global__ __launch_bounds__(512, 2) void conv1d_wgrad_bf16x3_kernel(WgradX3Args a) {
  constexpr int PAD = (KW - 1) / 2;
  constexpr int MT = 128, CT = 4 * NCF * 16;
  constexpr int YRB = MT * 2, XRB = CT * 2;
  constexpr int XROWS = KW == 1 ? 32 : 36;
  constexpr int YBYTES = 32 * YRB, XBYTES = XROWS * XRB;
  constexpr int STAGE = 2 * YBYTES + 2 * XBYTES;      // dY_hi, dY_lo, X_hi, X_lo
  constexpr int XPIECES = XBYTES / 1024, XROWS_PER_PIECE = 1024 / XRB;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave >> 2) * 64, wc0 = (wave & 3) * NCF * 16;
  const int ntile = a.mtiles * a.ctiles;
  const int id = xcd_remap(blockIdx.x, ntile * a.splits);
  const int split = id / ntile, t_id = id % ntile;
  const int m0 = (t_id / a.ctiles) * MT, c0 = (t_id % a.ctiles) * CT;
  const int vrows = a.nseg * a.total_rows;
  const int rbeg = split * a.chunks_per_split * 64;
  const int rend = min(vrows, rbeg + a.chunks_per_split * 64);
  const int n = (rend - rbeg) / 32;

  const int y_r = lane >> 4, y_s = lane & 15;
  const int x_r = (lane * 16) / XRB, x_s = ((lane * 16) % XRB) >> 4;
  auto src_slot = [](int slot, int row) { return (slot & 16) | (((((slot >> 1) & 7) ^ (row & 7)) << 1) | (slot & 1)); };
  const int last_row = a.total_rows - 1;
  const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)lds;
  // LDS-DMA as inline asm (scalar base + 32-bit lane offset): invisible to the compiler, which would otherwise drain
  // the whole ring (s_waitcnt vmcnt(0)) in front of every fragment read
  auto dma = [&](const char* sbase, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
  };
  int is_seg = rbeg / a.total_rows;          // segment and first row (inside it) of the K-tile the next issue() stages
  int is_row = rbeg - is_seg * a.total_rows;
  auto issue = [&](int stage) {
    const unsigned dst = lds0 + stage * STAGE;
    const char* const dy_hi = (const char*)a.dy[is_seg];
    const char* const dy_lo = (const char*)(a.dy[is_seg] + a.dy_plane);
    const char* const x_hi = (const char*)a.x[is_seg];
    const char* const x_lo = (const char*)(a.x[is_seg] + a.x_plane);
    {
      const int lr = 4 * wave + y_r;
      const int mcol = min(m0 + src_slot(y_s, lr) * 8, a.Mp - 8);
      const unsigned off = (unsigned)(((long)(is_row + lr) * a.Mp + mcol) * 2);
      dma(dy_hi, off, dst + wave * 1024);
      dma(dy_lo, off, dst + YBYTES + wave * 1024);
    }
#pragma unroll
    for (int q = 0; q < (XPIECES + 7) / 8; ++q) {
      const int p = wave + 8 * q;
      if (p < XPIECES) {
        const int lr = p * XROWS_PER_PIECE + x_r;
        int gr = is_row - PAD + lr;
        gr = gr < 0 ? 0 : (gr > last_row ? last_row : gr);
        const int ccol = min(c0 + src_slot(x_s, lr) * 8, a.Cp - 8);
        const unsigned off = (unsigned)(((long)gr * a.Cp + ccol) * 2);
        dma(x_hi, off, dst + 2 * YBYTES + p * 1024);
        dma(x_lo, off, dst + 2 * YBYTES + XBYTES + p * 1024);
      }
    }
    is_row += 32;
    if (is_row == a.total_rows) {
      is_row = 0;
      ++is_seg;
    }
  };

  const int g = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3;
  const int krow = 4 * g + q4;
  int ybase, xbase[KW];
  ybase = krow * YRB + ((krow & 7) << 5) + p4 * 8;
#pragma unroll
  for (int t = 0; t < KW; ++t) xbase[t] = (krow + t) * XRB + (((krow + t) & 7) << 5) + p4 * 8;
  int yseg[4], xseg[NCF], xline[NCF];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) yseg[mi] = ((wm0 >> 4) + mi) << 5;
#pragma unroll
  for (int cf = 0; cf < NCF; ++cf) {
    const int cb = (wc0 >> 4) + cf;
    xseg[cf] = (cb & 7) << 5;
    xline[cf] = (cb >> 3) * 256;
  }

  f32x4 acc[KW][4][NCF];
#pragma unroll
  for (int t = 0; t < KW; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NCF; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // bias gradient = column sums of dY (hi + lo): the workgroups of c-tile 0 multiply their dY fragments by an all-ones
  // operand as well (one wave per 64 m), instead of a separate pass re-reading dY
  const bool do_bias = a.bias_partial != nullptr && c0 == 0 && (wave & 3) == 0;
  f32x4 accb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  typedef short s16x8_t __attribute__((ext_vector_type(8)));
  const s16x8_t ones_raw = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
  bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_raw);
  // Opaque from here on: a known constant would be re-materialised by VALU moves right in front of each use, and the
  // compiler inserts the VALU-write -> MFMA-read wait states only for MFMAs it can see (the ones below are inline asm;
  // observed: wrong bias sums in the KW = 1 instantiation, where register pressure triggers the re-materialisation).
  asm volatile("" : "+v"(ones));

  // Fragment halves as the transposing reads return them.  Same phase structure as the convolution above: a K-tile
  // (32 rows) is hi*hi, hi*lo, [barrier], lo*hi; the barrier sits between phases whose operands are in registers,
  // the lo fragments are read during phase 1, the next K-tile's hi fragments during phase 3 (A hi and the X set that
  // phase 2 has finished with are dead by then), and the two X fragment sets swap roles from one K-tile to the next.
  bf16x8_t ah[4], al[4], b0[KW][NCF], b1[KW][NCF];
  typedef short s16x4_t __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) s16x4_t* lds_tr_ptr;
  // two transposing reads (k rows r..r+3 and r+16..r+19 of one 16-column block) -> one 8-element k fragment; through
  // the builtin the two halves land directly in the halves of the fragment's register tuple (no copies)
#define WX_TR(DST, BYTE_OFF, ROW_BYTES)                                                                              \
  {                                                                                                                  \
    const s16x4_t lo_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(lds + (BYTE_OFF)));                    \
    const s16x4_t hi_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(lds + (BYTE_OFF) + 16 * (ROW_BYTES))); \
    DST = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3, 4, 5, 6, 7));                  \
  }
#define WX_RDA(DST, STAGE, PLANE, MI) WX_TR(DST[MI], (STAGE) * STAGE_B + (PLANE) * YBYTES + (ybase ^ yseg[MI]), YRB)
#define WX_RDB(DST, STAGE, PLANE, TP, CF) \
  WX_TR(DST[TP][CF], (STAGE) * STAGE_B + 2 * YBYTES + (PLANE) * XBYTES + ((xbase[TP] ^ xseg[CF]) + xline[CF]), XRB)
  // MFMAs of one m-fragment against every (tap, c-fragment) of an X set (tied asm: see the convolution kernel)
#define WX_MM(A, B, MI)                                                                                    \
  _Pragma("unroll") for (int tp = 0; tp < KW; ++tp)                                                        \
  _Pragma("unroll") for (int cf = 0; cf < NCF; ++cf)                                                       \
      asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[tp][MI][cf]) : "v"(A[MI]), "v"(B[tp][cf]));
  // the bias MFMAs exist only in the loop the c-tile-0 waves run: a per-use `if (do_bias)` around inline asm makes hipcc
  // carry copies of the accumulators across every branch (31 v_mov_b64 each, in every wave)
#define WX_BIAS_ON(A, MI) asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(accb[MI]) : "v"(A[MI]), "v"(ones));
#define WX_BIAS_OFF(A, MI)
#define WX_SB __builtin_amdgcn_sched_barrier(0);
  constexpr int STAGE_B = STAGE;

  // one K-tile in stage S; BX = X hi fragments (already in registers), BY receives X lo, then the next tile's X hi
#define WX_TILE(S, BX, BY, MORE, WX_BIAS)                                                                         \
  /* phase 1: hi*hi; meanwhile X lo -> BY and dY lo -> al */                                               \
  WX_MM(ah, BX, 0) WX_BIAS(ah, 0) WX_SB                                                                    \
  _Pragma("unroll") for (int tp = 0; tp < KW; ++tp) _Pragma("unroll") for (int cf = 0; cf < NCF; ++cf) WX_RDB(BY, S, 1, tp, cf) \
  WX_SB WX_MM(ah, BX, 1) WX_BIAS(ah, 1) WX_SB                                                              \
  WX_RDA(al, S, 1, 0) WX_RDA(al, S, 1, 1) WX_RDA(al, S, 1, 2) WX_RDA(al, S, 1, 3)                          \
  WX_SB WX_MM(ah, BX, 2) WX_BIAS(ah, 2) WX_MM(ah, BX, 3) WX_BIAS(ah, 3) WX_SB                              \
  /* phase 2: hi*lo */                                                                                     \
  WX_MM(ah, BY, 0) WX_MM(ah, BY, 1) WX_MM(ah, BY, 2) WX_MM(ah, BY, 3) WX_SB                                \
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                              \
  __builtin_amdgcn_s_barrier();                                                                            \
  /* phase 3: lo*hi; meanwhile the DMA of K-tile t+2 into this stage and the next tile's hi fragments */   \
  if (MORE) issue(S);                                                                                      \
  WX_MM(al, BX, 0) WX_BIAS(al, 0) WX_SB                                                                    \
  WX_RDA(ah, (S) ^ 1, 0, 0) WX_RDA(ah, (S) ^ 1, 0, 1) WX_RDA(ah, (S) ^ 1, 0, 2) WX_RDA(ah, (S) ^ 1, 0, 3)  \
  WX_SB WX_MM(al, BX, 1) WX_BIAS(al, 1) WX_SB                                                              \
  _Pragma("unroll") for (int tp = 0; tp < KW; ++tp) _Pragma("unroll") for (int cf = 0; cf < NCF; ++cf) WX_RDB(BY, (S) ^ 1, 0, tp, cf) \
  WX_SB WX_MM(al, BX, 2) WX_BIAS(al, 2) WX_MM(al, BX, 3) WX_BIAS(al, 3) WX_SB

  const bool extra = (XPIECES % 8 != 0) && (wave < XPIECES % 8);   // this wave stages one more X piece per K-tile
  if (n > 0) {
    issue(0);
    if (n > 1) issue(1);
    if (n > 1) {
      if (extra) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * ((XPIECES + 7) / 8) + 2) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (XPIECES / 8) + 2) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    WX_RDA(ah, 0, 0, 0) WX_RDA(ah, 0, 0, 1) WX_RDA(ah, 0, 0, 2) WX_RDA(ah, 0, 0, 3)
#pragma unroll
    for (int tp = 0; tp < KW; ++tp)
#pragma unroll
      for (int cf = 0; cf < NCF; ++cf) WX_RDB(b0, 0, 0, tp, cf)
    if (do_bias) {
      for (int t = 0; t < n; t += 2) {
        WX_TILE(0, b0, b1, t + 2 < n, WX_BIAS_ON)
        WX_TILE(1, b1, b0, t + 3 < n, WX_BIAS_ON)
      }
    } else {
      for (int t = 0; t < n; t += 2) {
        WX_TILE(0, b0, b1, t + 2 < n, WX_BIAS_OFF)
        WX_TILE(1, b1, b0, t + 3 < n, WX_BIAS_OFF)
      }
    }
    // the compiler's hazard recogniser does not see inside the asm MFMAs: cover the MFMA-result -> VALU-read wait
    // states by hand before the accumulators are stored
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  }
#undef WX_TILE
#undef WX_SB
#undef WX_BIAS_ON
#undef WX_BIAS_OFF
#undef WX_MM
#undef WX_TR
#undef WX_RDB
#undef WX_RDA

  const int li = lane & 15, kq = lane >> 4;
  float* out = a.partial + (long)split * KW * a.M * a.C;
#pragma unroll
  for (int t = 0; t < KW; ++t)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int cf = 0; cf < NCF; ++cf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = m0 + wm0 + mi * 16 + kq * 4 + r;
          const int c = c0 + wc0 + cf * 16 + li;
          if (m < a.M && c < a.C) out[((long)t * a.M + m) * a.C + c] = acc[t][mi][cf][r];
        }
  if (do_bias && li == 0) {      // every column j of D holds the same sum; lane li = 0 writes it
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm0 + mi * 16 + kq * 4 + r;
        if (m < a.Mp) a.bias_partial[(long)split * a.Mp + m] = accb[mi][r];
      }
  }
}

// dbias[m] (+)= sum_s bias_partial[s][m], fixed order
static __global__ __launch_bounds__(256) void wgrad_x3_bias_reduce_kernel(const float* bp, float* dbias, int splits, int Mp, int M,
                                                                          int accumulate) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  float s = 0.f;
  int k = 0;
  for (; k + 8 <= splits; k += 8) {      // eight loads in flight (one at a time is a round trip to L2 per split)
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = bp[(long)(k + j) * Mp + m];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
  }
  for (; k < splits; ++k) s += bp[(long)k * Mp + m];
  dbias[m] = accumulate ? dbias[m] + s : s;
}

__global__ __launch_bounds__(256) void ncl_to_nlc_x3_kernel(const float* x, u16* y, long plane, int B, int C, int L, int Cp,
                                                            int rows_total) {
  __shared__ float tile[32][33];
  const int ct = Cp / 32;
  const int r0 = (blockIdx.x / ct) * 32, c0 = (blockIdx.x % ct) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int Lp1 = L + 1, ndata = B * Lp1;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, row = r0 + tx;
    int b, l;
    const bool ok = row_valid(row, Lp1, ndata, &b, &l) && c < C;
    tile[ty + 8 * i][tx] = ok ? x[((long)b * C + c) * L + l] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = r0 + ty + 8 * i, c = c0 + tx;
    if (row < rows_total) {
      u16 hi, lo;
      split2(tile[tx][ty + 8 * i], hi, lo);
      y[(long)row * Cp + c] = hi;
      y[plane + (long)row * Cp + c] = lo;
    }
  }
}

__global__ __launch_bounds__(256) void nlc_to_ncl_x3_kernel(const u16* x, long plane, float* y, int B, int C, int L, int Cp,
                                                            int rows_total) {
  __shared__ float tile[32][33];
  const int ct = Cp / 32;
  const int r0 = (blockIdx.x / ct) * 32, c0 = (blockIdx.x % ct) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int Lp1 = L + 1, ndata = B * Lp1;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = r0 + ty + 8 * i, c = c0 + tx;
    tile[ty + 8 * i][tx] = row < rows_total ? bf2f(x[(long)row * Cp + c]) + bf2f(x[plane + (long)row * Cp + c]) : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, row = r0 + tx;
    int b, l;
    if (row_valid(row, Lp1, ndata, &b, &l) && c < C) y[((long)b * C + c) * L + l] = tile[tx][ty + 8 * i];
  }
}

// out = t > 0 ? dy : 0 on both planes (the sign of a split value is the sign of its hi plane)
__global__ __launch_bounds__(256) void relu_mask_x3_kernel(const u16* dy, const u16* t, u16* out, long plane8, long n8) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n8; e += (long)gridDim.x * 256) {
    const u16x8 dh = ((const u16x8*)dy)[e], dl = ((const u16x8*)dy)[plane8 + e], m = ((const u16x8*)t)[e];
    u16x8 oh, ol;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const bool keep = bf2f(m[i]) > 0.f;
      oh[i] = keep ? dh[i] : (u16)0;
      ol[i] = keep ? dl[i] : (u16)0;
    }
    ((u16x8*)out)[e] = oh;
    ((u16x8*)out)[plane8 + e] = ol;
  }
}

template <int KW, int NCF>
static constexpr int wgrad_x3_lds() {
  return 2 * (2 * 32 * 256 + 2 * (KW == 1 ? 32 : 36) * (4 * NCF * 16 * 2));
}

static int wgrad_x3_tiles(int C, int M, int KW) { return ((M + 127) / 128) * ((C + (KW == 3 ? 128 : 256) - 1) / (KW == 3 ? 128 : 256)); }

constexpr int X3_BIAS_SPLITS = 64;     // upper bound of the split count (wgrad_split_plan)

}  // namespace alvq

using namespace alvq;

static inline int pad_to(int x, int q) { return (x + q - 1) / q * q; }
static inline long nlc_plane_elems(int B, int L, int C) {
  return ((long)alvq_nlc_rows(B, L) + 2L * alvq_nlc_guard_rows()) * pad_to(C, 64);
}

extern "C" int64_t alvq_nlc_plane_bytes(int B, int L, int C) {
  return (B <= 0 || L <= 0 || C <= 0) ? -1 : nlc_plane_elems(B, L, C) * 2;
}

extern "C" int alvq_pack_weight_bf16x3(const float* w, void* wp, int M, int C, int KW, int w_layout, void* stream) {
  const alvq_pack_desc d{w, wp, M, C, KW, w_layout};     // one-descriptor batch, hi + lo images (pack_weights.hip)
  return alvq_pack_weights_bf16_batch(&d, 1, 2, stream);
}

extern "C" int alvq_ncl_to_nlc_bf16x3(const float* x, void* y, int B, int C, int L, void* stream) {
  ALVQ_REQUIRE(x && y, ALVQ_EINVAL, "alvq_ncl_to_nlc_bf16x3: null pointer");
  ALVQ_REQUIRE(B > 0 && C > 0 && L > 0, ALVQ_EINVAL, "alvq_ncl_to_nlc_bf16x3: bad dims");
  const int Cp = pad_to(C, 64), rows = (int)alvq_nlc_rows(B, L);
  hipLaunchKernelGGL(ncl_to_nlc_x3_kernel, dim3((rows / 32) * (Cp / 32)), dim3(256), 0, (hipStream_t)stream, x, (u16*)y,
                     nlc_plane_elems(B, L, C), B, C, L, Cp, rows);
  return check_launch("alvq_ncl_to_nlc_bf16x3");
}

extern "C" int alvq_nlc_to_ncl_bf16x3(const void* x, float* y, int B, int C, int L, void* stream) {
  ALVQ_REQUIRE(x && y, ALVQ_EINVAL, "alvq_nlc_to_ncl_bf16x3: null pointer");
  ALVQ_REQUIRE(B > 0 && C > 0 && L > 0, ALVQ_EINVAL, "alvq_nlc_to_ncl_bf16x3: bad dims");
  const int Cp = pad_to(C, 64), rows = (int)alvq_nlc_rows(B, L);
  hipLaunchKernelGGL(nlc_to_ncl_x3_kernel, dim3((rows / 32) * (Cp / 32)), dim3(256), 0, (hipStream_t)stream, (const u16*)x,
                     nlc_plane_elems(B, L, C), y, B, C, L, Cp, rows);
  return check_launch("alvq_nlc_to_ncl_bf16x3");
}

extern "C" int alvq_relu_mask_bf16x3(const void* dy, const void* t, void* out, int B, int C, int L, void* stream) {
  ALVQ_REQUIRE(dy && t && out, ALVQ_EINVAL, "alvq_relu_mask_bf16x3: null pointer");
  ALVQ_REQUIRE(B > 0 && C > 0 && L > 0, ALVQ_EINVAL, "alvq_relu_mask_bf16x3: bad dims");
  const long n = (long)alvq_nlc_rows(B, L) * pad_to(C, 64);
  long g = (n / 8 + 1023) / 1024;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(relu_mask_x3_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, (const u16*)dy, (const u16*)t,
                     (u16*)out, nlc_plane_elems(B, L, C) / 8, n / 8);
  return check_launch("alvq_relu_mask_bf16x3");
}

extern "C" int alvq_conv1d_bf16x3(const void* x, const void* wp, const float* bias, const void* skip1, const void* skip2,
                                  const void* mask, const void* post, void* y, void* y2, float* y_ncl, int B, int C, int M,
                                  int L, int KW, int relu, void* stream) {
  ALVQ_REQUIRE(x && wp && (y || y_ncl), ALVQ_EINVAL, "alvq_conv1d_bf16x3: null x/wp/y");
  ALVQ_REQUIRE(!(y && y_ncl), ALVQ_EINVAL, "alvq_conv1d_bf16x3: choose one of y (NLC) and y_ncl (NCL fp32)");
  ALVQ_REQUIRE(B > 0 && C > 0 && M > 0 && L > 0, ALVQ_EINVAL, "alvq_conv1d_bf16x3: bad dims");
  ALVQ_REQUIRE(KW == 1 || KW == 3, ALVQ_EUNSUPPORTED, "alvq_conv1d_bf16x3: KW=%d (only 1 and 3)", KW);
  ALVQ_REQUIRE((y2 == nullptr) == (post == nullptr), ALVQ_EINVAL, "alvq_conv1d_bf16x3: y2 and post go together");
  ALVQ_REQUIRE(!y_ncl || (!skip1 && !skip2 && !mask && !post && !relu), ALVQ_EUNSUPPORTED,
               "alvq_conv1d_bf16x3: the NCL fp32 epilogue fuses bias only");
  ALVQ_REQUIRE((long)B * (L + 1) < (1L << 30), ALVQ_EUNSUPPORTED, "alvq_conv1d_bf16x3: problem too large");
  const long rows = alvq_nlc_rows(B, L);
  ConvX3Args a{{(const u16*)x, (const u16*)wp, bias, (const u16*)skip1, (const u16*)skip2, (const u16*)mask, (const u16*)post,
                (u16*)y, (u16*)y2, y_ncl, B, L, pad_to(C, 64), M, pad_to(M, 64), pad_to(M, WP_ROWS), relu ? 1 : 0,
                (int)(rows / X3_R), pad_to(M, X3_M) / X3_M},
               nlc_plane_elems(B, L, C), (long)alvq_packed_weight_elems(M, C, KW), nlc_plane_elems(B, L, M)};
  hipStream_t s = (hipStream_t)stream;
  static DeviceOnce attr;
  if (attr.need()) {
    (void)hipFuncSetAttribute((const void*)conv1d_bf16x3_kernel<0, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS2);
    (void)hipFuncSetAttribute((const void*)conv1d_bf16x3_kernel<1, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS2);
    (void)hipFuncSetAttribute((const void*)conv1d_bf16x3_kernel<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS2);
    (void)hipFuncSetAttribute((const void*)conv1d_bf16x3_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS2);
    (void)hipFuncSetAttribute((const void*)conv1d_bf16x3_kernel<1, 3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS2);
    (void)hipFuncSetAttribute((const void*)conv1d_bf16x3_kernel<1, 1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS2);
    (void)hipFuncSetAttribute((const void*)conv1d_bf16x3_kernel<0, 3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS2);
    (void)hipFuncSetAttribute((const void*)conv1d_bf16x3_kernel<0, 1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS2);
    (void)hipFuncSetAttribute((const void*)conv1d_bf16x3_kernel<1, 3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS2);
    (void)hipFuncSetAttribute((const void*)conv1d_bf16x3_kernel<1, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS2);
    (void)hipFuncSetAttribute((const void*)conv1d_bf16x3_kernel<0, 3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS2);
    (void)hipFuncSetAttribute((const void*)conv1d_bf16x3_kernel<0, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS2);
  }
  // fp32-NCL output of at most 128 channels (the pre-VQ convolution): 128-channel m-tile, no MFMA spent on padding channels
  // (option "fx_narrow" = 0 switches it off here as in the f16mx kernel); results are bit-identical to the 256-wide tile's
  // 128-channel m-tile: outputs of at most 128 channels (no MFMA spent on padding channels), and problems whose 256 x 256
  // tiles would leave CUs idle (option "fx_narrow" = 0 switches it off here as in the f16mx kernel; "fx_rows" = 256 forces
  // the wide tile for the second case); results are bit-identical to the 256-wide tile's
  const int forced = (int)option(OPT_FX_ROWS);
  const bool narrow = option(OPT_FX_NARROW) != 0 && (M <= 128 || (forced != 256 && a.b.rtiles * a.b.mtiles < 192) || forced == 128);
  if (narrow) a.b.mtiles = pad_to(M, 128) / 128;
  // still fewer than ~3/4 of the CUs covered: 128-row tiles as well
  const bool small = narrow && forced != 256 && (a.b.rtiles * a.b.mtiles < 192 || forced == 128);
  if (small) a.b.rtiles = (int)(rows / 128);
  const dim3 grid(a.b.rtiles * a.b.mtiles), block(512);
  if (small) {
    if (y) {
      if (KW == 3) hipLaunchKernelGGL((conv1d_bf16x3_kernel<0, 3, 1>), grid, block, X3_LDS2, s, a);
      else hipLaunchKernelGGL((conv1d_bf16x3_kernel<0, 1, 1>), grid, block, X3_LDS2, s, a);
    } else {
      if (KW == 3) hipLaunchKernelGGL((conv1d_bf16x3_kernel<1, 3, 1>), grid, block, X3_LDS2, s, a);
      else hipLaunchKernelGGL((conv1d_bf16x3_kernel<1, 1, 1>), grid, block, X3_LDS2, s, a);
    }
  } else if (narrow) {
    if (y) {
      if (KW == 3) hipLaunchKernelGGL((conv1d_bf16x3_kernel<0, 3, 2>), grid, block, X3_LDS2, s, a);
      else hipLaunchKernelGGL((conv1d_bf16x3_kernel<0, 1, 2>), grid, block, X3_LDS2, s, a);
    } else {
      if (KW == 3) hipLaunchKernelGGL((conv1d_bf16x3_kernel<1, 3, 2>), grid, block, X3_LDS2, s, a);
      else hipLaunchKernelGGL((conv1d_bf16x3_kernel<1, 1, 2>), grid, block, X3_LDS2, s, a);
    }
  } else if (y) {
    if (KW == 3) hipLaunchKernelGGL((conv1d_bf16x3_kernel<0, 3>), grid, block, X3_LDS2, s, a);
    else hipLaunchKernelGGL((conv1d_bf16x3_kernel<0, 1>), grid, block, X3_LDS2, s, a);
  } else {
    if (KW == 3) hipLaunchKernelGGL((conv1d_bf16x3_kernel<1, 3>), grid, block, X3_LDS2, s, a);
    else hipLaunchKernelGGL((conv1d_bf16x3_kernel<1, 1>), grid, block, X3_LDS2, s, a);
  }
  return check_launch("alvq_conv1d_bf16x3");
}

extern "C" int64_t alvq_conv1d_wgrad_bf16x3_workspace_bytes(int B, int C, int M, int L, int KW) {
  if (B <= 0 || C <= 0 || M <= 0 || L <= 0 || (KW != 1 && KW != 3)) return -1;
  const int splits = wgrad_split_bound((int)alvq_nlc_rows(B, L), wgrad_x3_tiles(C, M, KW), WX_MAXSEG);
  return (int64_t)splits * KW * M * C * 4 + (int64_t)X3_BIAS_SPLITS * pad_to(M, 64) * 4;
}

static int wgrad_x3_launch(const void* const* dy, const void* const* x, int nseg, float* dw, float* dbias, void* workspace, int B,
                           int C, int M, int L, int KW, int w_layout, int accumulate, hipStream_t s) {
  const int rows = (int)alvq_nlc_rows(B, L);
  const int ct = KW == 3 ? 128 : 256;
  WgradX3Args a{};
  for (int i = 0; i < WX_MAXSEG; ++i) {
    a.dy[i] = (const u16*)dy[i < nseg ? i : 0];
    a.x[i] = (const u16*)x[i < nseg ? i : 0];
  }
  a.nseg = nseg;
  a.partial = (float*)workspace;
  a.bias_partial = nullptr;
  a.dy_plane = nlc_plane_elems(B, L, M);
  a.x_plane = nlc_plane_elems(B, L, C);
  a.Mp = pad_to(M, 64); a.Cp = pad_to(C, 64); a.M = M; a.C = C;
  a.mtiles = (M + 127) / 128; a.ctiles = (C + ct - 1) / ct;
  a.total_rows = rows;
  a.splits = wgrad_split_plan(nseg * rows, a.mtiles * a.ctiles, &a.chunks_per_split);
  ALVQ_REQUIRE(a.mtiles * a.ctiles == wgrad_x3_tiles(C, M, KW) && a.splits <= wgrad_split_bound(rows, wgrad_x3_tiles(C, M, KW), WX_MAXSEG),
               ALVQ_EINVAL, "alvq_conv1d_wgrad_bf16x3: %d splits exceed what alvq_conv1d_wgrad_bf16x3_workspace_bytes sizes", a.splits);
  float* bpart = (float*)((char*)workspace + (int64_t)a.splits * KW * M * C * 4);
  if (dbias) a.bias_partial = bpart;
  static DeviceOnce attr;
  if (attr.need()) {
    (void)hipFuncSetAttribute((const void*)conv1d_wgrad_bf16x3_kernel<3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, wgrad_x3_lds<3, 2>());
    (void)hipFuncSetAttribute((const void*)conv1d_wgrad_bf16x3_kernel<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, wgrad_x3_lds<1, 4>());
  }
  const int grid = a.mtiles * a.ctiles * a.splits;
  if (KW == 3) hipLaunchKernelGGL((conv1d_wgrad_bf16x3_kernel<3, 2>), dim3(grid), dim3(512), (wgrad_x3_lds<3, 2>()), s, a);
  else hipLaunchKernelGGL((conv1d_wgrad_bf16x3_kernel<1, 4>), dim3(grid), dim3(512), (wgrad_x3_lds<1, 4>()), s, a);
  int rc = check_launch("alvq_conv1d_wgrad_bf16x3");
  if (rc) return rc;
  wgrad_reduce_launch((const float*)workspace, dw, a.splits, KW, M, C, w_layout, accumulate, s);
  if (dbias)     // single segment only (the shared residual weights have no bias)
    hipLaunchKernelGGL(wgrad_x3_bias_reduce_kernel, dim3((M + 255) / 256), dim3(256), 0, s, (const float*)bpart, dbias, a.splits,
                       a.Mp, M, accumulate);
  return check_launch("alvq_conv1d_wgrad_bf16x3/reduce");
}

extern "C" int alvq_conv1d_wgrad_bf16x3_splits(int B, int C, int M, int L, int KW, int nseg) {
  if (B <= 0 || C <= 0 || M <= 0 || L <= 0 || (KW != 1 && KW != 3) || nseg < 1 || nseg > WX_MAXSEG) return -1;
  int cps;
  return wgrad_split_plan(nseg * (int)alvq_nlc_rows(B, L), wgrad_x3_tiles(C, M, KW), &cps);
}

extern "C" int alvq_conv1d_wgrad_bf16x3(const void* dy, const void* x, float* dw, float* dbias, void* workspace, int B, int C,
                                        int M, int L, int KW, int w_layout, int accumulate, void* stream) {
  ALVQ_REQUIRE(dy && x && dw && workspace, ALVQ_EINVAL, "alvq_conv1d_wgrad_bf16x3: null pointer");
  ALVQ_REQUIRE(B > 0 && C > 0 && M > 0 && L > 0, ALVQ_EINVAL, "alvq_conv1d_wgrad_bf16x3: bad dims");
  ALVQ_REQUIRE(KW == 1 || KW == 3, ALVQ_EUNSUPPORTED, "alvq_conv1d_wgrad_bf16x3: KW=%d (only 1 and 3)", KW);
  ALVQ_REQUIRE(w_layout == ALVQ_W_OIK || w_layout == ALVQ_W_IOK, ALVQ_EINVAL, "alvq_conv1d_wgrad_bf16x3: w_layout");
  return wgrad_x3_launch(&dy, &x, 1, dw, dbias, workspace, B, C, M, L, KW, w_layout, accumulate, (hipStream_t)stream);
}

extern "C" int alvq_conv1d_wgrad_bf16x3_multi(const void* const* dy, const void* const* x, int nseg, float* dw, void* workspace,
                                              int B, int C, int M, int L, int KW, int w_layout, int accumulate, void* stream) {
  ALVQ_REQUIRE(dy && x && dw && workspace, ALVQ_EINVAL, "alvq_conv1d_wgrad_bf16x3_multi: null pointer");
  ALVQ_REQUIRE(nseg >= 1 && nseg <= WX_MAXSEG, ALVQ_EUNSUPPORTED, "alvq_conv1d_wgrad_bf16x3_multi: nseg=%d (1..4)", nseg);
  for (int i = 0; i < nseg; ++i) ALVQ_REQUIRE(dy[i] && x[i], ALVQ_EINVAL, "alvq_conv1d_wgrad_bf16x3_multi: null segment %d", i);
  ALVQ_REQUIRE(B > 0 && C > 0 && M > 0 && L > 0, ALVQ_EINVAL, "alvq_conv1d_wgrad_bf16x3_multi: bad dims");
  ALVQ_REQUIRE(KW == 1 || KW == 3, ALVQ_EUNSUPPORTED, "alvq_conv1d_wgrad_bf16x3_multi: KW=%d (only 1 and 3)", KW);
  ALVQ_REQUIRE(w_layout == ALVQ_W_OIK || w_layout == ALVQ_W_IOK, ALVQ_EINVAL, "alvq_conv1d_wgrad_bf16x3_multi: w_layout");
  return wgrad_x3_launch(dy, x, nseg, dw, nullptr, workspace, B, C, M, L, KW, w_layout, accumulate, (hipStream_t)stream);
}
