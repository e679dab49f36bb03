// f16mx weight gradient (format: f16mx_common.h):
//
//   dW_t[m][c] = sum_rows dY[row][m] * X[row + t - pad][c]        (all taps t of one (m, c) tile per workgroup)
//
// Same decomposition, staging and split reduction as the bf16x3 weight gradient (conv1d_bf16x3.hip): workgroup = 128 m x
// CT c x all taps, 8 waves (2 along m x 4 along c), K-tile = 32 rows = a dY slab pair and an X slab pair (H and Q planes)
// in one of two LDS stages.  The contraction runs over ROWS, the slow axis of both operands, so fragments come from the
// transposing LDS reads: ds_read_b64_tr_b16 for the fp16 planes, ds_read_b64_tr_b8 for the fp8 planes.  A wave's block
// is 2 (m) x NC (c) x KW tiles of 32x32 and a K-tile is two phases -- fp16 main term (2 k-steps of 32x32x16), barrier,
// fp8 cross terms (one block-scaled 32x32x64 per tile: block 0 = dY.hi8 x X.lo8, block 1 = dY.lo8 x X.hi8) -- with the
// other phase's fragments and the next K-tile's DMA in flight meanwhile.  A loss scale carried by dY is divided out
// (exactly: it is a power of two) when the fp32 partials are written.  The bias gradient (column sums of dY) is a
// separate bandwidth-bound pass here: as all-ones products in this kernel it needs two more 32x32 accumulators (32
// VGPRs), which the KW = 3 instantiation does not have.
#include <stdlib.h>

#include "alvq_common.h"
#include "f16mx_common.h"
#include "wgrad_reduce.h"

namespace alvq {

constexpr int WF_MAXSEG = 4;
struct WgradFxArgs {
  // Up to WF_MAXSEG (dy, x) pairs of identical shape whose products are summed into ONE dW: the R uses of a shared
  // residual weight (residual_stack.py:40-41) become a single longer contraction -- one split reduction instead of R.
  // Rows are numbered through all segments: virtual row v = seg * total_rows + r (total_rows % 64 == 0, so neither a
  // 64-row chunk nor a 32-row K-tile straddles two segments).
  const u16* dy[WF_MAXSEG];
  const u16* x[WF_MAXSEG];
  int nseg;
  float* partial;        // [splits][KW][M][C]
  const float* inv_scale;   // device scalar multiplied into every partial (1 / loss scale), or null
  long dy_plane, x_plane;
  int Mp, Cp, M, C;
  int mtiles, ctiles, splits, chunks_per_split, total_rows;
  int e;                 // E8M0 scale exponent of both operands (activations / scaled gradients)
  int dbg;               // ablation switches of the DBG instantiation (ALVQ_FX_DBG, as in conv1d_f16mx.hip): 1, 2, 4, 8
};

// NC / MF = 32-wide tiles per wave along c / along m: (1, 2) for KW = 3 (workgroup tile 128 m x 128 c x 3 taps, 96
// accumulator VGPRs) and (2, 4) for KW = 1 (256 m x 256 c, 128 accumulator VGPRs: the convolution's tile, at which a
// K-tile moves 32 bytes per matrix-pipe cycle per CU through LDS-DMA; the 128 x 256 tile of the first version moved 47).
template <int KW, int NC, int MF, bool DBG = false>
__global__ __launch_bounds__(512, 2) void conv1d_wgrad_f16mx_kernel(WgradFxArgs a) {
  const int dbg = DBG ? a.dbg : 0;
  constexpr int PAD = (KW - 1) / 2;
  constexpr int MT = 2 * MF * 32, CT = 4 * NC * 32;
  constexpr int YRB = MT * 2, XRB = CT * 2;
  constexpr int XROWS = KW == 1 ? 32 : 36;
  constexpr int YBYTES = 32 * YRB, XBYTES = XROWS * XRB;
  constexpr int STAGE = 2 * YBYTES + 2 * XBYTES;      // dY.H, dY.Q, X.H, X.Q
  constexpr int XPIECES = XBYTES / 1024, XROWS_PER_PIECE = 1024 / XRB;
  constexpr int YPIECES = YBYTES / 1024, YROWS_PER_PIECE = 1024 / YRB;     // 8 or 16 pieces per plane: 1 or 2 per wave
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave >> 2) * (MF * 32), wc0 = (wave & 3) * NC * 32;
  const int ntile = a.mtiles * a.ctiles;
  const int id = xcd_remap(blockIdx.x, ntile * a.splits);
  const int split = id / ntile, t_id = id % ntile;
  const int m0 = (t_id / a.ctiles) * MT, c0 = (t_id % a.ctiles) * CT;
  const int vrows = a.nseg * a.total_rows;
  const int rbeg = split * a.chunks_per_split * 64;
  const int rend = min(vrows, rbeg + a.chunks_per_split * 64);
  const int n = (rend - rbeg) / 32;

  // ---- staging: identical to the bf16x3 kernel (the Q plane has a bf16 plane's geometry: 64 bytes per 32 channels)
  const int y_r = (lane * 16) / YRB, y_s = ((lane * 16) % YRB) >> 4;
  const int x_r = (lane * 16) / XRB, x_s = ((lane * 16) % XRB) >> 4;
  auto src_slot = [](int slot, int row) { return (slot & 16) | (((((slot >> 1) & 7) ^ (row & 7)) << 1) | (slot & 1)); };
  const int last_row = a.total_rows - 1;
  const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)lds;
  auto dma = [&](const char* sbase, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
  };
  int is_seg = rbeg / a.total_rows;          // segment and first row (inside it) of the K-tile the next issue() stages
  int is_row = rbeg - is_seg * a.total_rows;
  auto issue = [&](int stage) {
    const unsigned dst = lds0 + stage * STAGE;
    const char* const dy_h = (const char*)a.dy[is_seg];
    const char* const dy_q = (const char*)(a.dy[is_seg] + a.dy_plane);
    const char* const x_h = (const char*)a.x[is_seg];
    const char* const x_q = (const char*)(a.x[is_seg] + a.x_plane);
#pragma unroll
    for (int q = 0; q < YPIECES / 8; ++q) {
      const int p = wave + 8 * q;
      const int lr = p * YROWS_PER_PIECE + y_r;
      const int mcol = min(m0 + src_slot(y_s, lr) * 8, a.Mp - 8);
      const unsigned off = (unsigned)(((long)(is_row + lr) * a.Mp + mcol) * 2);
      dma(dy_h, off, dst + p * 1024);
      dma(dy_q, off, dst + YBYTES + p * 1024);
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
        dma(x_h, off, dst + 2 * YBYTES + p * 1024);
        dma(x_q, off, dst + 2 * YBYTES + XBYTES + p * 1024);
      }
    }
    is_row += 32;
    if (is_row == a.total_rows) {
      is_row = 0;
      ++is_seg;
    }
  };

  // ---- transposed fragment reads.  Lane l: i = l & 15 (lane of its 16-group), blk = (l >> 4) & 1 (which 16-column half
  // of the 32-wide tile), g = l >> 5 (k group of the MFMA).
  //  fp16 (tr_b16: a 16-group reads a 4-row x 16-column block; lane 4q+p supplies the address of block row q, columns
  //    4p..4p+3 -- ANY LDS row may stand for block row q):  k-step s, group g covers the 8 rows 2q + g + 8s (first
  //    read) and 16 + 2q + g + 8s (second read), q = 0..3 -- the same permutation of the K-tile's rows for both operands,
  //    which a contraction does not care about.  A half-wave (one g) holds two 16-column blocks, i.e. two neighbouring
  //    32-byte segments L and L+1, times four rows: with rows of ONE parity the (row & 7) segment swizzle sends the
  //    block-0 lanes to L ^ {g, 2+g, 4+g, 6+g} and the block-1 lanes to the complementary four segments -- 8 distinct
  //    segments, conflict-free for every tap offset (four CONSECUTIVE rows collide two-fold: measured 33 % of the
  //    LDS cycles in this kernel).
  //  fp8 (tr_b8: an 8-row x 16-column block; lane 2q+p supplies row q, bytes 8p..8p+7): group g covers rows 16g..16g+15
  //    in two reads, once in the hi8 segment and once in the lo8 segment of the tile's 64-byte chunk.
  const int i16 = lane & 15, blk = (lane >> 4) & 1, g = lane >> 5;
  const int q4 = i16 >> 2, p4 = i16 & 3;
  const int krow = 2 * q4 + g;                       // + 8 s for k-step s, + 16 for the second read
  const int qrow = 16 * g + (i16 >> 1);              // + 8 for the second read
  const int qbyte = 16 * blk + 8 * (i16 & 1);
  typedef short s16x4_t __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) s16x4_t* lds_tr_ptr;
  typedef int i32x2 __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(3))) i32x2* lds_tr8_ptr;
  // Byte offset of (row, 32-byte segment seg) in a slab with ROWB-byte rows: row * ROWB + (seg >> 3) * 256 +
  // (((seg & 7) ^ (row & 7)) << 5).  The segment of fragment f of a wave is (wave part) + 2 f + (lane part) with the
  // three parts in disjoint bits, so the XOR factors: offset = LANE BASE ^ (f << 6) (^ 32 for the lo8 segment) -- one
  // base register per operand and tap instead of one address register per fragment.
  const int sA = wm0 >> 4, sB = wc0 >> 4;            // first 16-column block (= 32-byte H segment) of the wave
  const int aHb = krow * YRB + p4 * 8 + (sA >> 3) * 256 + ((((sA & 7) ^ blk) ^ (krow & 7)) << 5);
  const int aQb = qrow * YRB + qbyte + (sA >> 3) * 256 + (((sA & 7) ^ (qrow & 7)) << 5);   // Q chunk c = segments 2c, 2c+1
  int bHb[KW], bQb[KW];
#pragma unroll
  for (int t = 0; t < KW; ++t) {
    bHb[t] = (krow + t) * XRB + p4 * 8 + (sB >> 3) * 256 + ((((sB & 7) ^ blk) ^ ((krow + t) & 7)) << 5);
    bQb[t] = (qrow + t) * XRB + qbyte + (sB >> 3) * 256 + (((sB & 7) ^ ((qrow + t) & 7)) << 5);
  }

  f16x8_t aH[MF][2], bH[KW][NC][2];
  i32x8 aQ[MF], bQ[KW][NC];
#define WF_TR16(DST, OFF, ROWB)                                                                                  \
  {                                                                                                              \
    const s16x4_t lo_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(lds + (OFF)));                    \
    const s16x4_t hi_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(lds + (OFF) + 16 * (ROWB)));      \
    DST = __builtin_bit_cast(f16x8_t, __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3, 4, 5, 6, 7));                \
  }
  // FIRST / SECOND: byte offsets of the two 16-row column pieces that form blocks 0 and 1 of the scaled MFMA's operand
#define WF_TR8(DST, FIRST, SECOND, ROWB)                                                                         \
  {                                                                                                              \
    const i32x2 a0_ = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_tr8_ptr)(lds + (FIRST)));                     \
    const i32x2 a1_ = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_tr8_ptr)(lds + (FIRST) + 8 * (ROWB)));       \
    const i32x2 b0_ = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_tr8_ptr)(lds + (SECOND)));                    \
    const i32x2 b1_ = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_tr8_ptr)(lds + (SECOND) + 8 * (ROWB)));      \
    DST = i32x8{a0_[0], a0_[1], a1_[0], a1_[1], b0_[0], b0_[1], b1_[0], b1_[1]};                                \
  }
  // A operand (dY), tile MI; B operand (X), tap TP, tile CF: rows shifted by TP; B's blocks are (lo8, hi8) so that block 0
  // pairs dY.hi8 with X.lo8
#define WF_RDH_A(STAGE_, MI, KS) WF_TR16(aH[MI][KS], (STAGE_) * STAGE + (aHb ^ ((MI) << 6)) + 8 * (KS) * YRB, YRB)
#define WF_RDQ_A(STAGE_, MI) \
  WF_TR8(aQ[MI], (STAGE_) * STAGE + YBYTES + (aQb ^ ((MI) << 6)), (STAGE_) * STAGE + YBYTES + (aQb ^ ((MI) << 6) ^ 32), YRB)
#define WF_RDH_B(STAGE_, TP, CF, KS) \
  WF_TR16(bH[TP][CF][KS], (STAGE_) * STAGE + 2 * YBYTES + (bHb[TP] ^ ((CF) << 6)) + 8 * (KS) * XRB, XRB)
#define WF_RDQ_B(STAGE_, TP, CF)                                                                 \
  WF_TR8(bQ[TP][CF], (STAGE_) * STAGE + 2 * YBYTES + XBYTES + (bQb[TP] ^ ((CF) << 6) ^ 32),      \
         (STAGE_) * STAGE + 2 * YBYTES + XBYTES + (bQb[TP] ^ ((CF) << 6)), XRB)

  // block scales: both operands are activation-class; A blocks = (hi8, lo8), B blocks = (lo8, hi8)
  int sa = g ? a.e - FX_LO_SHIFT : a.e, sb = g ? a.e : a.e - FX_LO_SHIFT;
  asm volatile("" : "+v"(sa), "+v"(sb));   // opaque: see the convolution kernel

  f32x16 acc[KW][MF][NC];
#pragma unroll
  for (int t = 0; t < KW; ++t)
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
      for (int jn = 0; jn < NC; ++jn)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][i][jn][q] = 0.f;
#define WF_MMH(MI, KS)                                                                                           \
  _Pragma("unroll") for (int tp = 0; tp < KW; ++tp) _Pragma("unroll") for (int cf = 0; cf < NC; ++cf)            \
      asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[tp][MI][cf]) : "v"(aH[MI][KS]), "v"(bH[tp][cf][KS]));
#define WF_MMQ(MI)                                                                                               \
  _Pragma("unroll") for (int tp = 0; tp < KW; ++tp) _Pragma("unroll") for (int cf = 0; cf < NC; ++cf)            \
      asm("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]"                           \
          : "+v"(acc[tp][MI][cf]) : "v"(aQ[MI]), "v"(bQ[tp][cf]), "v"(sa), "v"(sb));
#define WF_SB __builtin_amdgcn_sched_barrier(0);
#define WF_ALL_B(M_) _Pragma("unroll") for (int tp = 0; tp < KW; ++tp) _Pragma("unroll") for (int cf = 0; cf < NC; ++cf) { M_ }

  // one K-tile in stage S
#define WF_ALL_A(M_) _Pragma("unroll") for (int mi = 0; mi < MF; ++mi) { M_ }
#define WF_TILE(S, MORE)                                                                                           \
  /* phase 1: fp16 main term; meanwhile this K-tile's Q fragments */                                            \
  WF_MMH(0, 0) WF_SB                                                                                             \
  if (!(dbg & 2)) { WF_ALL_B(WF_RDQ_B(S, tp, cf)) } WF_SB                                                        \
  WF_MMH(1, 0) WF_SB                                                                                             \
  if (!(dbg & 2)) { WF_ALL_A(WF_RDQ_A(S, mi)) } WF_SB                                                            \
  _Pragma("unroll") for (int mi = 2; mi < MF; ++mi) { WF_MMH(mi, 0) }                                            \
  WF_ALL_A(WF_MMH(mi, 1)) WF_SB                                                                                  \
  if (!(dbg & 4)) {                                                                                              \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                  \
    __builtin_amdgcn_s_barrier();                                                                                \
  }                                                                                                              \
  /* phase 2: fp8 cross terms; meanwhile the DMA of K-tile t+2 into this stage and the next tile's H fragments */ \
  if ((MORE) && !(dbg & 1)) issue(S);                                                                            \
  WF_MMQ(0) WF_SB                                                                                                \
  if (!(dbg & 2)) { WF_ALL_A(WF_RDH_A((S) ^ 1, mi, 0)) WF_ALL_B(WF_RDH_B((S) ^ 1, tp, cf, 0)) } WF_SB            \
  WF_MMQ(1) WF_SB                                                                                                \
  if (!(dbg & 2)) { WF_ALL_A(WF_RDH_A((S) ^ 1, mi, 1)) WF_ALL_B(WF_RDH_B((S) ^ 1, tp, cf, 1)) } WF_SB            \
  _Pragma("unroll") for (int mi = 2; mi < MF; ++mi) { WF_MMQ(mi) }                                               \
  WF_SB

  const bool extra = (XPIECES % 8 != 0) && (wave < XPIECES % 8);   // this wave stages one more X piece per K-tile
  if (n > 0) {
    issue(0);
    if (n > 1) issue(1);
    if (n > 1) {   // K-tile 0 landed; K-tile 1's pieces (this wave's count) may stay in flight
      if (extra) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * ((XPIECES + 7) / 8) + 2 * (YPIECES / 8)) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (XPIECES / 8) + 2 * (YPIECES / 8)) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    WF_ALL_A(WF_RDH_A(0, mi, 0) WF_RDH_A(0, mi, 1))
    WF_ALL_B(WF_RDH_B(0, tp, cf, 0) WF_RDH_B(0, tp, cf, 1))
    for (int t = 0; t < n; t += 2) {
      WF_TILE(0, t + 2 < n)
      WF_TILE(1, t + 3 < n)
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // MFMA-result -> VALU-read wait states (asm MFMAs are invisible)
  }
#undef WF_TILE
#undef WF_ALL_A
#undef WF_ALL_B
#undef WF_SB
#undef WF_MMQ
#undef WF_MMH
#undef WF_RDQ_B
#undef WF_RDH_B
#undef WF_RDQ_A
#undef WF_RDH_A
#undef WF_TR8
#undef WF_TR16

  // ---- partial[split][t][m][c] = acc / loss scale (fp32); D[i = m][j = c]: lane (j = lane & 31, g), register q holds
  // m = (q & 3) + 8 (q >> 2) + 4 g
  if (dbg & 8) {
    if (acc[0][0][0][0] == 12345.678f) a.partial[0] = 1.f;
    return;
  }
  const float inv = a.inv_scale ? *a.inv_scale : 1.f;
  const int jc = lane & 31;
  float* out = a.partial + (long)split * KW * a.M * a.C;
#pragma unroll
  for (int t = 0; t < KW; ++t)
#pragma unroll
    for (int mi = 0; mi < MF; ++mi)
#pragma unroll
      for (int cf = 0; cf < NC; ++cf)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int m = m0 + wm0 + mi * 32 + (q & 3) + 8 * (q >> 2) + 4 * g;
          const int c = c0 + wc0 + cf * 32 + jc;
          if (m < a.M && c < a.C) out[((long)t * a.M + m) * a.C + c] = acc[t][mi][cf][q] * inv;
        }
}

// ---------------------------------------------------------------------------------------------------- bias gradient
// partial[split][m] = sum over the split's rows of dY[row][m] = H + lo8 * S_lo: 64 channels x one row range per
// workgroup; thread = (8 channels, row lane of 32), rows strided by 32, then a fixed-order reduction over the row lanes.
__global__ __launch_bounds__(256) void bias_grad_fx_partial_kernel(const u16* dy, long plane, float* partial, int rows, int Mp,
                                                                   int rows_per_split, int e) {
  const int cg = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int c = blockIdx.x * 64 + cg * 8;
  const int rb = blockIdx.y * rows_per_split, re = min(rows, rb + rows_per_split);
  const float s_lo = fx_pow2(e - FX_LO_SHIFT);
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int r = rb + rl; r < re; r += 32) {
    const u32x4 h = *(const u32x4*)(dy + (long)r * Mp + c);
    const u32x2 ql = *(const u32x2*)((const unsigned char*)(dy + plane) + (long)r * Mp * 2 + fx_q_off(c) + 32);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float v0, v1;
      fx_join2(h[k], ql[k >> 1], (k & 1) * 2, s_lo, v0, v1);
      s[2 * k] += v0;
      s[2 * k + 1] += v1;
    }
  }
  __shared__ float red[32][65];
#pragma unroll
  for (int k = 0; k < 8; ++k) red[rl][cg * 8 + k] = s[k];
  __syncthreads();
  if (threadIdx.x < 64) {
    float t = 0.f;
    for (int k = 0; k < 32; ++k) t += red[k][threadIdx.x];
    partial[(long)blockIdx.y * Mp + blockIdx.x * 64 + threadIdx.x] = t;
  }
}

// dbias[m] (+)= inv_scale * sum_s partial[s][m]: 32 channels x 8 split phases per workgroup (coalesced along m, 8-way
// parallel along the splits), fixed order.  The split count is a compile-time constant so that a thread's 16 loads are
// all in flight at once (as a run-time loop they were 16 round trips to L2 one after the other: 31 us for 512 KB).
constexpr int FX_BIAS_SPLITS = 128;
static __global__ __launch_bounds__(256) void wgrad_fx_bias_reduce_kernel(const float* bp, float* dbias, int Mp, int M,
                                                                          int accumulate, const float* inv_scale) {
  const int mi = threadIdx.x & 31, ph = threadIdx.x >> 5;
  const int m = blockIdx.x * 32 + mi;
  float v[FX_BIAS_SPLITS / 8];
#pragma unroll
  for (int k = 0; k < FX_BIAS_SPLITS / 8; ++k) v[k] = m < M ? bp[(long)(ph + 8 * k) * Mp + m] : 0.f;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < FX_BIAS_SPLITS / 8; ++k) s += v[k];
  __shared__ float red[8][32];
  red[ph][mi] = s;
  __syncthreads();
  if (ph == 0 && m < M) {
    float t = ((red[0][mi] + red[1][mi]) + (red[2][mi] + red[3][mi])) + ((red[4][mi] + red[5][mi]) + (red[6][mi] + red[7][mi]));
    t *= inv_scale ? *inv_scale : 1.f;
    dbias[m] = accumulate ? dbias[m] + t : t;
  }
}

template <int KW, int NC, int MF>
static constexpr int wgrad_fx_lds() {
  return 2 * (2 * 32 * (2 * MF * 32 * 2) + 2 * (KW == 1 ? 32 : 36) * (4 * NC * 32 * 2));
}

// tiles of a launch: 128 x 128 x 3 taps for width 3, 256 x 256 for width 1
static int wgrad_fx_tiles(int C, int M, int KW) {
  const int ct = KW == 3 ? 128 : 256, mt = KW == 3 ? 128 : 256;
  return ((M + mt - 1) / mt) * ((C + ct - 1) / ct);
}

}  // namespace alvq

using namespace alvq;

static inline int pad_to(int x, int q) { return (x + q - 1) / q * q; }
static inline long nlc_plane_elems(int B, int L, int C) {
  return ((long)alvq_nlc_rows(B, L) + 2L * alvq_nlc_guard_rows()) * pad_to(C, 64);
}

extern "C" int64_t alvq_conv1d_wgrad_f16mx_workspace_bytes(int B, int C, int M, int L, int KW) {
  if (B <= 0 || C <= 0 || M <= 0 || L <= 0 || (KW != 1 && KW != 3)) return -1;
  const int splits = wgrad_split_bound((int)alvq_nlc_rows(B, L), wgrad_fx_tiles(C, M, KW), WF_MAXSEG);
  return (int64_t)splits * KW * M * C * 4 + (int64_t)FX_BIAS_SPLITS * pad_to(M, 64) * 4;
}

static int wgrad_fx_launch(const void* const* dy, const void* const* x, int nseg, float* dw, float* dbias, void* workspace, int B,
                           int C, int M, int L, int KW, int w_layout, int accumulate, const float* inv_scale, hipStream_t s) {
  const int rows = (int)alvq_nlc_rows(B, L);
  const int ct = KW == 3 ? 128 : 256, mt = KW == 3 ? 128 : 256;
  WgradFxArgs a{};
  for (int i = 0; i < WF_MAXSEG; ++i) {
    a.dy[i] = (const u16*)dy[i < nseg ? i : 0];
    a.x[i] = (const u16*)x[i < nseg ? i : 0];
  }
  a.nseg = nseg;
  a.partial = (float*)workspace;
  a.inv_scale = inv_scale;
  a.dy_plane = nlc_plane_elems(B, L, M);
  a.x_plane = nlc_plane_elems(B, L, C);
  a.Mp = pad_to(M, 64); a.Cp = pad_to(C, 64); a.M = M; a.C = C;
  a.mtiles = (M + mt - 1) / mt; a.ctiles = (C + ct - 1) / ct;
  a.total_rows = rows; a.e = FX_E_ACT;
#ifdef ALVQ_DEBUG_KERNELS   // ablation instantiations: debug library only (build.py --debug-kernels)
  static const int dbg_env = getenv("ALVQ_FX_DBG") ? atoi(getenv("ALVQ_FX_DBG")) : 0;   // timing ablations (results are garbage)
#else
  constexpr int dbg_env = 0;
#endif
  a.dbg = dbg_env;
  a.splits = wgrad_split_plan(nseg * rows, a.mtiles * a.ctiles, &a.chunks_per_split);
  ALVQ_REQUIRE(a.mtiles * a.ctiles == wgrad_fx_tiles(C, M, KW) && a.splits <= wgrad_split_bound(rows, wgrad_fx_tiles(C, M, KW), WF_MAXSEG),
               ALVQ_EINVAL, "alvq_conv1d_wgrad_f16mx: %d splits exceed what alvq_conv1d_wgrad_f16mx_workspace_bytes sizes", a.splits);
  float* bpart = (float*)((char*)workspace + (int64_t)a.splits * KW * M * C * 4);
  static DeviceOnce attr;
  if (attr.need()) {
    (void)hipFuncSetAttribute((const void*)conv1d_wgrad_f16mx_kernel<3, 1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, wgrad_fx_lds<3, 1, 2>());
    (void)hipFuncSetAttribute((const void*)conv1d_wgrad_f16mx_kernel<1, 2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, wgrad_fx_lds<1, 2, 4>());
#ifdef ALVQ_DEBUG_KERNELS
    (void)hipFuncSetAttribute((const void*)conv1d_wgrad_f16mx_kernel<3, 1, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, wgrad_fx_lds<3, 1, 2>());
    (void)hipFuncSetAttribute((const void*)conv1d_wgrad_f16mx_kernel<1, 2, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, wgrad_fx_lds<1, 2, 4>());
#endif
  }
  const int grid = a.mtiles * a.ctiles * a.splits;
#ifdef ALVQ_DEBUG_KERNELS
  if (dbg_env) {
    if (KW == 3) hipLaunchKernelGGL((conv1d_wgrad_f16mx_kernel<3, 1, 2, true>), dim3(grid), dim3(512), (wgrad_fx_lds<3, 1, 2>()), s, a);
    else hipLaunchKernelGGL((conv1d_wgrad_f16mx_kernel<1, 2, 4, true>), dim3(grid), dim3(512), (wgrad_fx_lds<1, 2, 4>()), s, a);
  } else
#endif
  if (KW == 3) hipLaunchKernelGGL((conv1d_wgrad_f16mx_kernel<3, 1, 2>), dim3(grid), dim3(512), (wgrad_fx_lds<3, 1, 2>()), s, a);
  else hipLaunchKernelGGL((conv1d_wgrad_f16mx_kernel<1, 2, 4>), dim3(grid), dim3(512), (wgrad_fx_lds<1, 2, 4>()), s, a);
  int rc = check_launch("alvq_conv1d_wgrad_f16mx");
  if (rc) return rc;
  wgrad_reduce_launch((const float*)workspace, dw, a.splits, KW, M, C, w_layout, accumulate, s);
  if (dbias) {     // single segment only (the shared residual weights have no bias)
    const int rps = (rows + FX_BIAS_SPLITS - 1) / FX_BIAS_SPLITS;
    hipLaunchKernelGGL(bias_grad_fx_partial_kernel, dim3(a.Mp / 64, FX_BIAS_SPLITS), dim3(256), 0, s, (const u16*)dy[0], a.dy_plane, bpart,
                       rows, a.Mp, rps, a.e);
    hipLaunchKernelGGL(wgrad_fx_bias_reduce_kernel, dim3((M + 31) / 32), dim3(256), 0, s, (const float*)bpart, dbias, a.Mp, M,
                       accumulate, inv_scale);
  }
  return check_launch("alvq_conv1d_wgrad_f16mx/reduce");
}

extern "C" int alvq_conv1d_wgrad_f16mx_splits(int B, int C, int M, int L, int KW, int nseg) {
  if (B <= 0 || C <= 0 || M <= 0 || L <= 0 || (KW != 1 && KW != 3) || nseg < 1 || nseg > WF_MAXSEG) return -1;
  int cps;
  return wgrad_split_plan(nseg * (int)alvq_nlc_rows(B, L), wgrad_fx_tiles(C, M, KW), &cps);
}

extern "C" int alvq_conv1d_wgrad_f16mx(const void* dy, const void* x, float* dw, float* dbias, void* workspace, int B, int C, int M,
                                       int L, int KW, int w_layout, int accumulate, const float* inv_scale, void* stream) {
  ALVQ_REQUIRE(dy && x && dw && workspace, ALVQ_EINVAL, "alvq_conv1d_wgrad_f16mx: null pointer");
  ALVQ_REQUIRE(B > 0 && C > 0 && M > 0 && L > 0, ALVQ_EINVAL, "alvq_conv1d_wgrad_f16mx: bad dims");
  ALVQ_REQUIRE(KW == 1 || KW == 3, ALVQ_EUNSUPPORTED, "alvq_conv1d_wgrad_f16mx: KW=%d (only 1 and 3)", KW);
  ALVQ_REQUIRE(w_layout == ALVQ_W_OIK || w_layout == ALVQ_W_IOK, ALVQ_EINVAL, "alvq_conv1d_wgrad_f16mx: w_layout");
  return wgrad_fx_launch(&dy, &x, 1, dw, dbias, workspace, B, C, M, L, KW, w_layout, accumulate, inv_scale, (hipStream_t)stream);
}

extern "C" int alvq_conv1d_wgrad_f16mx_multi(const void* const* dy, const void* const* x, int nseg, float* dw, void* workspace,
                                             int B, int C, int M, int L, int KW, int w_layout, int accumulate,
                                             const float* inv_scale, void* stream) {
  ALVQ_REQUIRE(dy && x && dw && workspace, ALVQ_EINVAL, "alvq_conv1d_wgrad_f16mx_multi: null pointer");
  ALVQ_REQUIRE(nseg >= 1 && nseg <= WF_MAXSEG, ALVQ_EUNSUPPORTED, "alvq_conv1d_wgrad_f16mx_multi: nseg=%d (1..4)", nseg);
  for (int i = 0; i < nseg; ++i) ALVQ_REQUIRE(dy[i] && x[i], ALVQ_EINVAL, "alvq_conv1d_wgrad_f16mx_multi: null segment %d", i);
  ALVQ_REQUIRE(B > 0 && C > 0 && M > 0 && L > 0, ALVQ_EINVAL, "alvq_conv1d_wgrad_f16mx_multi: bad dims");
  ALVQ_REQUIRE(KW == 1 || KW == 3, ALVQ_EUNSUPPORTED, "alvq_conv1d_wgrad_f16mx_multi: KW=%d (only 1 and 3)", KW);
  ALVQ_REQUIRE(w_layout == ALVQ_W_OIK || w_layout == ALVQ_W_IOK, ALVQ_EINVAL, "alvq_conv1d_wgrad_f16mx_multi: w_layout");
  return wgrad_fx_launch(dy, x, nseg, dw, nullptr, workspace, B, C, M, L, KW, w_layout, accumulate, inv_scale, (hipStream_t)stream);
}
