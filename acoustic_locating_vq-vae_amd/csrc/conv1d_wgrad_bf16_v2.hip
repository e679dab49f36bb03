// bf16 weight-gradient, ring-pipelined kernel (v2).
//
//   dW_t[m][c] = sum_rows dY[row][m] * X[row + t - pad][c]        (all taps t of one (m, c) tile per workgroup)
//
// The contraction runs over ROWS, the slow axis of both NLC operands, so both MFMA operands are fetched with the
// transposing LDS read ds_read_b64_tr_b16 (two per 16x32 fragment).  Workgroup = 8 waves (2 along m x 4 along c);
// a wave owns 64 m x (NCF*16) c for every tap: KW=3 -> NCF=2 (c-tile 128, 96 accumulator VGPRs),
// KW=1 -> NCF=4 (c-tile 256, 64 accumulator VGPRs).  K-tile = 32 rows: a dY slab [32][128 m] (8 KB) and an X slab
// [32 + halo][c-tile] staged ONCE and re-read at row offsets 0/1/2 by the taps.  Same 4-stage LDS-DMA ring and
// counted-vmcnt pipeline as conv1d_bf16_v2.hip, with one barrier per PAIR of K-tiles.
//
// Bank conflicts: LDS rows are 256 B (or 512 B) = whole bank lines, so without care the 8 rows a half-wave reads
// would hit the same banks.  The 32-B segment s of row r is stored at segment s ^ (r & 7) of its 256-B line
// (swizzle applied on the DMA source address and on the read address).  The MFMA k index is also permuted --
// lane group g takes rows {4g..4g+3} and {16+4g..16+4g+3} of the K-tile, identically for both operands, which a
// contraction does not care about -- so each half-wave reads 8 CONSECUTIVE rows: 8 distinct segments, conflict free
// for every tap offset.
#include <stdlib.h>

#include "alvq_common.h"
#include "bf16_common.h"
#include "wgrad_reduce.h"

namespace alvq {

typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));

constexpr int WG_MAXSEG = 4;
struct WgradV2Args {
  // Up to WG_MAXSEG (dy, x) pairs of identical shape whose products are summed into ONE dW: the R uses of a shared
  // residual weight (residual_stack.py:40-41) become a single longer contraction -- one split reduction instead of R.
  const u16* dy[WG_MAXSEG];   // [rows][Mp]
  const u16* x[WG_MAXSEG];    // [rows][Cp]
  float* partial;             // [splits][KW][M][C]
  float* bias_partial;        // [splits][Mp] column sums of dY (the bias gradient), or null
  int Mp, Cp, M, C;
  int mtiles, ctiles, splits, chunks_per_split, total_rows, nseg;
};

// Two transposing reads (rows r..r+3 and r+16..r+19 of one 16-column block) -> one 8-element k fragment.
// Inline asm on purpose: with the builtin, hipcc (ROCm 7.2) drains the LDS-DMA ring with s_waitcnt vmcnt(0) in
// front of every fragment read (an in-flight LDS-DMA is a pending LDS write on the VM counter).  The reads are
// therefore invisible to the compiler's lgkmcnt bookkeeping: the caller waits with lgkm_drain() before first use.
typedef unsigned long long u64;
template <int ROW_BYTES>
__device__ __forceinline__ void tr_pair_issue(unsigned lds_addr, u64& lo, u64& hi) {
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(lds_addr));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(lds_addr), "n"(16 * ROW_BYTES));
}
__device__ __forceinline__ bf16x8_t tr_pair_join(u64 lo, u64 hi) {
  typedef u64 u64x2 __attribute__((ext_vector_type(2)));
  const u64x2 v = {lo, hi};
  return __builtin_bit_cast(bf16x8_t, v);
}
__device__ __forceinline__ void lgkm_drain() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);   // keep the consuming MFMAs behind the wait (guide 5.4 rule 18)
}

template <int KW, int NCF, int F16 = 0>
__global__ __launch_bounds__(512, 2) void conv1d_wgrad_bf16_v2_kernel(WgradV2Args a) {
  constexpr int PAD = (KW - 1) / 2;
  constexpr int MT = 128, CT = 4 * NCF * 16;          // tile: 128 m x CT c
  constexpr int YRB = MT * 2;                          // 256 B rows
  constexpr int XRB = CT * 2;                          // 256 or 512 B rows
  constexpr int XROWS = KW == 1 ? 32 : 36;             // halo rows, rounded so the slab is whole 1-KB pieces
  constexpr int YBYTES = 32 * YRB;                     // 8192
  constexpr int XBYTES = XROWS * XRB;                  // 9216 / 16384
  constexpr int STAGE = YBYTES + XBYTES;
  constexpr int XPIECES = XBYTES / 1024;               // 9 / 16
  constexpr int XROWS_PER_PIECE = 1024 / XRB;          // 4 / 2
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave >> 2) * 64, wc0 = (wave & 3) * NCF * 16;

  const int ntile = a.mtiles * a.ctiles;
  const int id = xcd_remap(blockIdx.x, ntile * a.splits);
  const int split = id / ntile, t_id = id % ntile;
  const int m0 = (t_id / a.ctiles) * MT, c0 = (t_id % a.ctiles) * CT;
  // rows are numbered through all segments: virtual row v = seg * total_rows + r (total_rows % 64 == 0, so neither a
  // 64-row chunk nor a 32-row K-tile straddles two segments)
  const int vrows = a.nseg * a.total_rows;
  const int rbeg = split * a.chunks_per_split * 64;
  const int rend = min(vrows, rbeg + a.chunks_per_split * 64);
  const int n = (rend - rbeg) / 32;                    // K-tiles in this split (even; may be 0)

  // ---- DMA: lane i of a 1-KB piece covers bytes [16i, 16i+16): row = 16i / RB, 16-B slot = (16i % RB) / 16.
  // The slot holds logical slot (line, ((slot>>1)&7) ^ (row&7), slot&1).
  const int y_r = lane >> 4, y_s = lane & 15;                        // dY piece: 4 rows x 16 slots
  const int x_r = (lane * 16) / XRB, x_s = ((lane * 16) % XRB) >> 4;  // X piece: 4 x 16 or 2 x 32 slots
  auto src_slot = [](int slot, int row) { return (slot & 16) | (((((slot >> 1) & 7) ^ (row & 7)) << 1) | (slot & 1)); };
  const int last_row = a.total_rows - 1;

  int is_seg = rbeg / a.total_rows;          // segment and first row (inside it) of the K-tile the next issue() stages
  int is_row = rbeg - is_seg * a.total_rows;
  auto issue = [&](int stage) {
    unsigned char* dst = lds + stage * STAGE;
    const u16* const dyp = a.dy[is_seg];
    const u16* const xp = a.x[is_seg];
    {  // dY: piece = wave (rows 4*wave .. +3)
      const int lr = 4 * wave + y_r;
      const int mcol = min(m0 + src_slot(y_s, lr) * 8, a.Mp - 8);   // tiles past Mp re-read the last chunk (discarded)
      glds16(dyp + (long)(is_row + lr) * a.Mp + mcol, dst + wave * 1024);
    }
#pragma unroll
    for (int q = 0; q < (XPIECES + 7) / 8; ++q) {
      const int p = wave + 8 * q;
      if (p < XPIECES) {
        const int lr = p * XROWS_PER_PIECE + x_r;
        int gr = is_row - PAD + lr;                                  // rows outside the matrix -> a zero row
        gr = gr < 0 ? 0 : (gr > last_row ? last_row : gr);
        const int ccol = min(c0 + src_slot(x_s, lr) * 8, a.Cp - 8);
        glds16(xp + (long)gr * a.Cp + ccol, dst + YBYTES + p * 1024);
      }
    }
    is_row += 32;
    if (is_row == a.total_rows) {
      is_row = 0;
      ++is_seg;
    }
  };
  // glds issued per K-tile by THIS wave (for the counted waits)
  const bool extra = (XPIECES % 8 != 0) && (wave < XPIECES % 8);

  // ---- transposed fragment reads.  Lane (g = lane>>4, q = (lane>>2)&3, p = lane&3) supplies the address of block
  // row q, columns 4p..4p+3; block rows of group g: 4g + q (first read) and 16 + 4g + q (second read).
  const int g = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3;
  const int krow = 4 * g + q4;
  int ybase, xbase[KW];
  ybase = krow * YRB + ((krow & 7) << 5) + p4 * 8;
#pragma unroll
  for (int t = 0; t < KW; ++t) xbase[t] = (krow + t) * XRB + (((krow + t) & 7) << 5) + p4 * 8;
  // column-block terms (wave-uniform): block index cb -> line = cb >> 3, segment = cb & 7
  int yseg[4], xseg[NCF], xline[NCF];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) yseg[mi] = ((wm0 >> 4) + mi) << 5;          // MT = 128 -> 8 blocks, one line
#pragma unroll
  for (int cf = 0; cf < NCF; ++cf) {
    const int cb = (wc0 >> 4) + cf;
    xseg[cf] = (cb & 7) << 5;
    xline[cf] = (cb >> 3) * 256;
  }

  struct Raw {       // fragment halves as they come back from the transposing reads
    u64 alo[4], ahi[4];
    u64 blo[KW][NCF], bhi[KW][NCF];
  };
  struct Frags {
    bf16x8_t a[4];
    bf16x8_t b[KW][NCF];
  };
  const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)lds;
  auto rd = [&](Raw& f, int stage) {
    const unsigned ys = lds0 + stage * STAGE;
    const unsigned xs = ys + YBYTES;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) tr_pair_issue<YRB>(ys + (ybase ^ yseg[mi]), f.alo[mi], f.ahi[mi]);
#pragma unroll
    for (int t = 0; t < KW; ++t)
#pragma unroll
      for (int cf = 0; cf < NCF; ++cf)
        tr_pair_issue<XRB>(xs + ((xbase[t] ^ xseg[cf]) + xline[cf]), f.blo[t][cf], f.bhi[t][cf]);
  };
  auto join = [&](Frags& f, const Raw& r) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) f.a[mi] = tr_pair_join(r.alo[mi], r.ahi[mi]);
#pragma unroll
    for (int t = 0; t < KW; ++t)
#pragma unroll
      for (int cf = 0; cf < NCF; ++cf) f.b[t][cf] = tr_pair_join(r.blo[t][cf], r.bhi[t][cf]);
  };

  f32x4 acc[KW][4][NCF];
#pragma unroll
  for (int t = 0; t < KW; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NCF; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // Bias gradient = column sums of dY: the workgroups of c-tile 0 multiply their dY fragments by an all-ones
  // operand as well (one wave per 64 m; 4 extra MFMAs per K-tile), instead of a separate pass re-reading dY.
  const bool do_bias = a.bias_partial != nullptr && c0 == 0 && (wave & 3) == 0;
  f32x4 accb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr short ONE = F16 ? 0x3C00 : 0x3F80;        // 1.0 as fp16 / bf16
  const s16x8_t ones_raw = {ONE, ONE, ONE, ONE, ONE, ONE, ONE, ONE};
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_raw);
  auto mm = [&](const Frags& f, int half) {
#pragma unroll
    for (int mi = half * 2; mi < half * 2 + 2; ++mi)
#pragma unroll
      for (int t = 0; t < KW; ++t)
#pragma unroll
        for (int cf = 0; cf < NCF; ++cf)
          acc[t][mi][cf] = elem_mfma16<F16>(f.a[mi], f.b[t][cf], acc[t][mi][cf]);
    if (do_bias) {
#pragma unroll
      for (int mi = half * 2; mi < half * 2 + 2; ++mi)
        accb[mi] = elem_mfma16<F16>(f.a[mi], ones, accb[mi]);
    }
  };
  auto wait_keep = [&](int tiles_in_flight) {   // leave the DMA of `tiles_in_flight` K-tiles (0..2) outstanding
    if (tiles_in_flight == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (XPIECES % 8 == 0) {
      if (tiles_in_flight == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else if (extra) {
      if (tiles_in_flight == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      if (tiles_in_flight == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
  };

  Raw r0, r1;
  Frags f0, f1;
  // K-tiles are walked in PAIRS that share one barrier (a barrier per K-tile cost 9 % of this kernel: with 24 MFMAs
  // per wave a K-tile is too short to amortise it).  Pair (t, t+1) lives in two of the four stages; at the barrier in
  // the middle of the pair every wave has all of the pair's fragments in registers (the reads of tile t+1 are issued
  // during tile t and drained before it) and has waited for the next pair's DMA, so after it the pair's two stages
  // are free -- the DMA of the pair after next goes straight into them -- and the next pair is visible to everyone.
  if (n > 0) {
    issue(0);
    issue(1);
    if (n > 2) {
      issue(2);
      issue(3);
    }
    wait_keep(n > 2 ? 2 : 0);
    __builtin_amdgcn_s_barrier();
    rd(r0, 0);
    lgkm_drain();
    for (int t = 0; t < n; t += 2) {
      // ---- first K-tile of the pair (its reads sit in r0, complete)
      join(f0, r0);
      mm(f0, 0);
      __builtin_amdgcn_sched_barrier(0);
      rd(r1, (t + 1) & 3);
      __builtin_amdgcn_sched_barrier(0);
      mm(f0, 1);
      if (t + 2 < n) wait_keep(0);         // the next pair (issued two K-tiles ago) has landed
      lgkm_drain();                        // r1 landed (issued 12+ MFMAs ago)
      __builtin_amdgcn_s_barrier();
      // ---- second K-tile
      if (t + 4 < n) {
        issue((t + 4) & 3);
        issue((t + 5) & 3);
      }
      join(f1, r1);
      mm(f1, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (t + 2 < n) rd(r0, (t + 2) & 3);
      __builtin_amdgcn_sched_barrier(0);
      mm(f1, 1);
      lgkm_drain();
    }
  }

  // ---- partial[split][t][m][c] = acc (fp32); D[i = m][j = c]
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

// ------------------------------------------------------------------------------------ v3: 32x32 MFMAs, two fragment sets
// The weight gradient without a fused bias gradient (the shared residual weights, i.e. the two largest launches of a
// step per width) on the structure of the f16mx weight gradient (conv1d_wgrad_f16mx.hip) minus its fp8 half:
// v_mfma_f32_32x32x16_bf16, transposing reads with rows of one parity per half-wave (conflict-free), XOR-factored
// fragment addresses (one base register per operand and tap), tied asm MFMAs, LDS-DMA as asm.
//   KW = 1: (NC, MF) = (2, 4): 256 m x 256 c per workgroup, a wave owns 128 x 64 -- half the LDS-DMA bytes per MFMA of
//           the 128 x 256 tile above, which ran at a matrix-pipe utilisation of 0.33;
//   KW = 3: (NC, MF) = (1, 2): 128 m x 128 c x 3 taps, a wave owns 64 x 32 x 3.
// With no second phase to hide them in, the next K-tile's fragments are read into a SECOND register set while this
// K-tile's MFMAs run, the ring is four stages deep (K-tile t+3 is requested at the top of K-tile t: two K-tiles of lead)
// and there is one barrier per K-tile, after which K-tile t+1 is visible and the stage K-tile t-1 occupied is free.
typedef __attribute__((address_space(3))) s16x4_t* v3_lds_tr_ptr;

template <int KW, int NC, int MF, int F16 = 0>
__global__ __launch_bounds__(512, 2) void conv1d_wgrad_bf16_v3_kernel(WgradV2Args a) {
  constexpr int PAD = (KW - 1) / 2;
  constexpr int MT = 2 * MF * 32, CT = 4 * NC * 32;
  constexpr int YRB = MT * 2, XRB = CT * 2;
  constexpr int XROWS = KW == 1 ? 32 : 36;
  constexpr int YBYTES = 32 * YRB, XBYTES = XROWS * XRB;
  constexpr int STAGE = YBYTES + XBYTES;
  constexpr int XPIECES = XBYTES / 1024, XROWS_PER_PIECE = 1024 / XRB;
  constexpr int YPIECES = YBYTES / 1024, YROWS_PER_PIECE = 1024 / YRB;
  constexpr int PER_WAVE = YPIECES / 8 + XPIECES / 8;        // DMA pieces per K-tile of a wave without the extra one
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

  // ---- staging (as in the kernels above: 1-KB pieces, 32-byte segments swizzled by the row)
  const int y_r = (lane * 16) / YRB, y_s = ((lane * 16) % YRB) >> 4;
  const int x_r = (lane * 16) / XRB, x_s = ((lane * 16) % XRB) >> 4;
  auto src_slot = [](int slot, int row) { return (slot & 16) | (((((slot >> 1) & 7) ^ (row & 7)) << 1) | (slot & 1)); };
  const int last_row = a.total_rows - 1;
  const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)lds;
  auto dma = [&](const char* sbase, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
  };
  const bool extra = (XPIECES % 8 != 0) && (wave < XPIECES % 8);   // this wave stages one more X piece per K-tile
  int is_seg = rbeg / a.total_rows;
  int is_row = rbeg - is_seg * a.total_rows;
  auto issue = [&](int stage) {
    const unsigned dst = lds0 + stage * STAGE;
    const char* const dyp = (const char*)a.dy[is_seg];
    const char* const xp = (const char*)a.x[is_seg];
#pragma unroll
    for (int q = 0; q < YPIECES / 8; ++q) {
      const int p = wave + 8 * q;
      const int lr = p * YROWS_PER_PIECE + y_r;
      const int mcol = min(m0 + src_slot(y_s, lr) * 8, a.Mp - 8);
      dma(dyp, (unsigned)(((long)(is_row + lr) * a.Mp + mcol) * 2), dst + p * 1024);
    }
#pragma unroll
    for (int q = 0; q < (XPIECES + 7) / 8; ++q) {
      const int p = wave + 8 * q;
      if (p < XPIECES) {
        const int lr = p * XROWS_PER_PIECE + x_r;
        int gr = is_row - PAD + lr;
        gr = gr < 0 ? 0 : (gr > last_row ? last_row : gr);
        const int ccol = min(c0 + src_slot(x_s, lr) * 8, a.Cp - 8);
        dma(xp, (unsigned)(((long)gr * a.Cp + ccol) * 2), dst + YBYTES + p * 1024);
      }
    }
    is_row += 32;
    if (is_row == a.total_rows) {
      is_row = 0;
      ++is_seg;
    }
  };
  auto wait_keep = [&](int tiles) {      // leave the DMA of `tiles` K-tiles (0..2) of this wave outstanding
    if (tiles == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (tiles == 1) {
      if (extra) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_WAVE + 1) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_WAVE) : "memory");
    } else {
      if (extra) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER_WAVE + 2) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER_WAVE) : "memory");
    }
  };

  // ---- transposed fragment reads (lane geometry and swizzle of conv1d_wgrad_f16mx.hip)
  const int i16 = lane & 15, blk = (lane >> 4) & 1, g = lane >> 5;
  const int q4 = i16 >> 2, p4 = i16 & 3;
  const int krow = 2 * q4 + g;
  const int sA = wm0 >> 4, sB = wc0 >> 4;
  const int aHb = krow * YRB + p4 * 8 + (sA >> 3) * 256 + ((((sA & 7) ^ blk) ^ (krow & 7)) << 5);
  int bHb[KW];
#pragma unroll
  for (int t = 0; t < KW; ++t) bHb[t] = (krow + t) * XRB + p4 * 8 + (sB >> 3) * 256 + ((((sB & 7) ^ blk) ^ ((krow + t) & 7)) << 5);
  bf16x8_t aF[2][MF][2], bF[2][KW][NC][2];       // [register set][...][k-step]
#define V3_TR16(DST, OFF, ROWB)                                                                                    \
  {                                                                                                                \
    const s16x4_t lo_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v3_lds_tr_ptr)(lds + (OFF)));                    \
    const s16x4_t hi_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v3_lds_tr_ptr)(lds + (OFF) + 16 * (ROWB)));      \
    DST = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3, 4, 5, 6, 7));                 \
  }
#define V3_RD_A(SET, ST, MI, KS) V3_TR16(aF[SET][MI][KS], (ST) * STAGE + (aHb ^ ((MI) << 6)) + 8 * (KS) * YRB, YRB)
#define V3_RD_B(SET, ST, TP, CF, KS) V3_TR16(bF[SET][TP][CF][KS], (ST) * STAGE + YBYTES + (bHb[TP] ^ ((CF) << 6)) + 8 * (KS) * XRB, XRB)
#define V3_MM(SET, MI, TP, CF, KS)                                                                                              \
  if (F16) asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[TP][MI][CF]) : "v"(aF[SET][MI][KS]), "v"(bF[SET][TP][CF][KS])); \
  else asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[TP][MI][CF]) : "v"(aF[SET][MI][KS]), "v"(bF[SET][TP][CF][KS]));
#define V3_SB __builtin_amdgcn_sched_barrier(0);

  f32x16 acc[KW][MF][NC];
#pragma unroll
  for (int t = 0; t < KW; ++t)
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
      for (int jn = 0; jn < NC; ++jn)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][i][jn][q] = 0.f;

  // One K-tile held in register set SET; NS = the stage of the next K-tile, read into the other set meanwhile (two
  // transposing reads in the shadow of one MFMA each, ordered by first use).  Written out per instantiation.
#define V3_TILE_K1(SET, NS, MORE)                                                                                  \
  V3_MM(SET, 0, 0, 0, 0) V3_SB if (MORE) { V3_RD_B((SET) ^ 1, NS, 0, 0, 0) } V3_SB                                 \
  V3_MM(SET, 0, 0, 1, 0) V3_SB if (MORE) { V3_RD_A((SET) ^ 1, NS, 0, 0) } V3_SB                                    \
  V3_MM(SET, 1, 0, 0, 0) V3_SB if (MORE) { V3_RD_B((SET) ^ 1, NS, 0, 1, 0) } V3_SB                                 \
  V3_MM(SET, 1, 0, 1, 0) V3_SB if (MORE) { V3_RD_A((SET) ^ 1, NS, 1, 0) } V3_SB                                    \
  V3_MM(SET, 2, 0, 0, 0) V3_SB if (MORE) { V3_RD_A((SET) ^ 1, NS, 2, 0) } V3_SB                                    \
  V3_MM(SET, 2, 0, 1, 0) V3_SB if (MORE) { V3_RD_A((SET) ^ 1, NS, 3, 0) } V3_SB                                    \
  V3_MM(SET, 3, 0, 0, 0) V3_SB if (MORE) { V3_RD_B((SET) ^ 1, NS, 0, 0, 1) } V3_SB                                 \
  V3_MM(SET, 3, 0, 1, 0) V3_SB if (MORE) { V3_RD_A((SET) ^ 1, NS, 0, 1) } V3_SB                                    \
  V3_MM(SET, 0, 0, 0, 1) V3_SB if (MORE) { V3_RD_B((SET) ^ 1, NS, 0, 1, 1) } V3_SB                                 \
  V3_MM(SET, 0, 0, 1, 1) V3_SB if (MORE) { V3_RD_A((SET) ^ 1, NS, 1, 1) } V3_SB                                    \
  V3_MM(SET, 1, 0, 0, 1) V3_SB if (MORE) { V3_RD_A((SET) ^ 1, NS, 2, 1) } V3_SB                                    \
  V3_MM(SET, 1, 0, 1, 1) V3_SB if (MORE) { V3_RD_A((SET) ^ 1, NS, 3, 1) } V3_SB                                    \
  V3_MM(SET, 2, 0, 0, 1) V3_MM(SET, 2, 0, 1, 1) V3_MM(SET, 3, 0, 0, 1) V3_MM(SET, 3, 0, 1, 1) V3_SB
#define V3_TILE_K3(SET, NS, MORE)                                                                                  \
  V3_MM(SET, 0, 0, 0, 0) V3_SB if (MORE) { V3_RD_B((SET) ^ 1, NS, 0, 0, 0) } V3_SB                                 \
  V3_MM(SET, 0, 1, 0, 0) V3_SB if (MORE) { V3_RD_A((SET) ^ 1, NS, 0, 0) } V3_SB                                    \
  V3_MM(SET, 0, 2, 0, 0) V3_SB if (MORE) { V3_RD_B((SET) ^ 1, NS, 1, 0, 0) } V3_SB                                 \
  V3_MM(SET, 1, 0, 0, 0) V3_SB if (MORE) { V3_RD_B((SET) ^ 1, NS, 2, 0, 0) } V3_SB                                 \
  V3_MM(SET, 1, 1, 0, 0) V3_SB if (MORE) { V3_RD_A((SET) ^ 1, NS, 1, 0) } V3_SB                                    \
  V3_MM(SET, 1, 2, 0, 0) V3_SB if (MORE) { V3_RD_B((SET) ^ 1, NS, 0, 0, 1) } V3_SB                                 \
  V3_MM(SET, 0, 0, 0, 1) V3_SB if (MORE) { V3_RD_A((SET) ^ 1, NS, 0, 1) } V3_SB                                    \
  V3_MM(SET, 0, 1, 0, 1) V3_SB if (MORE) { V3_RD_B((SET) ^ 1, NS, 1, 0, 1) } V3_SB                                 \
  V3_MM(SET, 0, 2, 0, 1) V3_SB if (MORE) { V3_RD_B((SET) ^ 1, NS, 2, 0, 1) } V3_SB                                 \
  V3_MM(SET, 1, 0, 0, 1) V3_SB if (MORE) { V3_RD_A((SET) ^ 1, NS, 1, 1) } V3_SB                                    \
  V3_MM(SET, 1, 1, 0, 1) V3_MM(SET, 1, 2, 0, 1) V3_SB
#define V3_TILE(SET, NS, MORE) if constexpr (KW == 1) { V3_TILE_K1(SET, NS, MORE) } else { V3_TILE_K3(SET, NS, MORE) }

  if (n > 0) {
    issue(0);
    if (n > 1) issue(1);
    if (n > 2) issue(2);
    wait_keep(n > 2 ? 2 : (n > 1 ? 1 : 0));      // K-tile 0 landed
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int mi = 0; mi < MF; ++mi) { V3_RD_A(0, 0, mi, 0) V3_RD_A(0, 0, mi, 1) }
#pragma unroll
    for (int tp = 0; tp < KW; ++tp)
#pragma unroll
      for (int cf = 0; cf < NC; ++cf) { V3_RD_B(0, 0, tp, cf, 0) V3_RD_B(0, 0, tp, cf, 1) }
    // K-tile t: its fragments are in set t & 1.  At the top: wait until K-tile t+1 has landed (K-tile t+2 may stay in
    // flight), barrier (every wave has finished reading K-tile t's stage during K-tile t-1 -- and K-tile t-1's stage
    // before that), request K-tile t+3 into the stage of K-tile t-1.
#define V3_TOP(T)                                                                                                  \
  if ((T) + 1 < n) {                                                                                               \
    wait_keep((T) + 2 < n ? 1 : 0);                                                                                \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                             \
    __builtin_amdgcn_s_barrier();                                                                                  \
    if ((T) + 3 < n) issue(((T) + 3) & 3);                                                                         \
  } else {                                                                                                         \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                             \
  }
    for (int t = 0; t < n; t += 4) {       // n is even; four K-tiles per iteration so that stage indices are constants
      V3_TOP(t)
      V3_TILE(0, 1, t + 1 < n)
      V3_TOP(t + 1)
      V3_TILE(1, 2, t + 2 < n)
      if (t + 2 < n) {
        V3_TOP(t + 2)
        V3_TILE(0, 3, t + 3 < n)
        V3_TOP(t + 3)
        V3_TILE(1, 0, t + 4 < n)
      }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // MFMA-result -> VALU-read wait states (asm MFMAs are invisible)
  }
#undef V3_TOP
#undef V3_TILE
#undef V3_TILE_K3
#undef V3_TILE_K1
#undef V3_SB
#undef V3_MM
#undef V3_RD_B
#undef V3_RD_A
#undef V3_TR16

  // ---- partial[split][t][m][c] (fp32); D[i = m][j = c]: lane (j = lane & 31, g), register q holds m = (q & 3) + 8 (q >> 2) + 4 g
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
          if (m < a.M && c < a.C) out[((long)t * a.M + m) * a.C + c] = acc[t][mi][cf][q];
        }
}

template <int KW, int NC, int MF>
static constexpr int wgrad_v3_lds() {
  return 4 * (32 * (2 * MF * 32 * 2) + (KW == 1 ? 32 : 36) * (4 * NC * 32 * 2));
}

// dbias[m] (+)= sum_s bias_partial[s][m], fixed order
static __global__ __launch_bounds__(256) void wgrad_bias_reduce_kernel(const float* bp, float* dbias, int splits, int Mp, int M,
                                                                       int accumulate, const float* scale) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  float s = 0.f;
  for (int k = 0; k < splits; ++k) s += bp[(long)k * Mp + m];
  if (scale) s *= *scale;
  dbias[m] = accumulate ? dbias[m] + s : s;
}

template <int KW, int NCF>
static constexpr int wgrad_v2_lds() {
  return 4 * (32 * 256 + (KW == 1 ? 32 : 36) * (4 * NCF * 16 * 2));
}

// option "wgrad_v3": which launches without a bias gradient (the shared residual weights) run the v3 kernels --
// 0 none, 1 width 1 only, 3 (default) both.
static int wgrad_v3_select() { return (int)option(OPT_WGRAD_V3); }
static bool wgrad_uses_v3(int KW, bool with_bias) {
  return !with_bias && ((KW == 1 && (wgrad_v3_select() & 1)) || (KW == 3 && (wgrad_v3_select() & 2)));
}
// tiles of a launch: 128 m x {128 c x 3 taps | 256 c}; the v3 width-1 kernel owns 256 m x 256 c
static int wgrad_v2_tiles(int C, int M, int KW, bool v3) {
  const int ct = KW == 3 ? 128 : 256, mt = (v3 && KW == 1) ? 256 : 128;
  return ((M + mt - 1) / mt) * ((C + ct - 1) / ct);
}

int conv1d_wgrad_bf16_v2_splits(int total_rows, int C, int M, int KW, int nseg, bool with_bias) {
  int cps;
  return wgrad_split_plan(nseg * total_rows, wgrad_v2_tiles(C, M, KW, wgrad_uses_v3(KW, with_bias)), &cps);
}

int64_t conv1d_wgrad_bf16_v2_workspace_bytes(int total_rows, int C, int M, int KW) {
  // the larger of the two kernels' bounds: either may serve a launch of this shape (with / without a bias gradient)
  const int s2 = wgrad_split_bound(total_rows, wgrad_v2_tiles(C, M, KW, false), WG_MAXSEG);
  const int s3 = wgrad_split_bound(total_rows, wgrad_v2_tiles(C, M, KW, true), WG_MAXSEG);
  return (int64_t)(s2 > s3 ? s2 : s3) * KW * M * C * 4;
}

int conv1d_wgrad_bf16_v2_launch(const void* const* dy, const void* const* x, int nseg, float* dw, void* workspace,
                                int total_rows, int C, int M, int KW, int w_layout, int accumulate, hipStream_t s,
                                float* dbias, float* bias_partial, int elem, const float* out_scale) {
  const int ct = KW == 3 ? 128 : 256;
  const int Mp = (M + 63) / 64 * 64, Cp = (C + 63) / 64 * 64;
  WgradV2Args a{};
  for (int i = 0; i < WG_MAXSEG; ++i) {
    a.dy[i] = (const u16*)dy[i < nseg ? i : 0];
    a.x[i] = (const u16*)x[i < nseg ? i : 0];
  }
  a.partial = (float*)workspace;
  a.bias_partial = dbias ? bias_partial : nullptr;
  a.Mp = Mp; a.Cp = Cp; a.M = M; a.C = C;
  a.mtiles = (M + 127) / 128; a.ctiles = (C + ct - 1) / ct;
  a.total_rows = total_rows; a.nseg = nseg;
  const bool v3 = wgrad_uses_v3(KW, dbias != nullptr);        // no bias gradient (the shared residual weights): the v3 kernels
  if (v3 && KW == 1) a.mtiles = (M + 255) / 256;
  a.splits = wgrad_split_plan(nseg * total_rows, a.mtiles * a.ctiles, &a.chunks_per_split);
  ALVQ_REQUIRE(a.mtiles * a.ctiles == wgrad_v2_tiles(C, M, KW, v3) &&
                   (int64_t)a.splits * KW * M * C * 4 <= conv1d_wgrad_bf16_v2_workspace_bytes(total_rows, C, M, KW),
               ALVQ_EINVAL, "alvq_conv1d_wgrad_bf16: %d splits exceed what alvq_conv1d_wgrad_bf16_workspace_bytes sizes", a.splits);
  static DeviceOnce attr;
  if (attr.need()) {
    (void)hipFuncSetAttribute((const void*)conv1d_wgrad_bf16_v3_kernel<1, 2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, wgrad_v3_lds<1, 2, 4>());
    (void)hipFuncSetAttribute((const void*)conv1d_wgrad_bf16_v3_kernel<3, 1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, wgrad_v3_lds<3, 1, 2>());
    (void)hipFuncSetAttribute((const void*)conv1d_wgrad_bf16_v2_kernel<3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, wgrad_v2_lds<3, 2>());
    (void)hipFuncSetAttribute((const void*)conv1d_wgrad_bf16_v2_kernel<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, wgrad_v2_lds<1, 4>());
    (void)hipFuncSetAttribute((const void*)conv1d_wgrad_bf16_v3_kernel<1, 2, 4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, wgrad_v3_lds<1, 2, 4>());
    (void)hipFuncSetAttribute((const void*)conv1d_wgrad_bf16_v3_kernel<3, 1, 2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, wgrad_v3_lds<3, 1, 2>());
    (void)hipFuncSetAttribute((const void*)conv1d_wgrad_bf16_v2_kernel<3, 2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, wgrad_v2_lds<3, 2>());
    (void)hipFuncSetAttribute((const void*)conv1d_wgrad_bf16_v2_kernel<1, 4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, wgrad_v2_lds<1, 4>());
  }
  const int grid = a.mtiles * a.ctiles * a.splits;
  if (elem) {
    if (v3 && KW == 1) hipLaunchKernelGGL((conv1d_wgrad_bf16_v3_kernel<1, 2, 4, 1>), dim3(grid), dim3(512), (wgrad_v3_lds<1, 2, 4>()), s, a);
    else if (v3) hipLaunchKernelGGL((conv1d_wgrad_bf16_v3_kernel<3, 1, 2, 1>), dim3(grid), dim3(512), (wgrad_v3_lds<3, 1, 2>()), s, a);
    else if (KW == 3) hipLaunchKernelGGL((conv1d_wgrad_bf16_v2_kernel<3, 2, 1>), dim3(grid), dim3(512), (wgrad_v2_lds<3, 2>()), s, a);
    else hipLaunchKernelGGL((conv1d_wgrad_bf16_v2_kernel<1, 4, 1>), dim3(grid), dim3(512), (wgrad_v2_lds<1, 4>()), s, a);
  } else
  if (v3 && KW == 1) hipLaunchKernelGGL((conv1d_wgrad_bf16_v3_kernel<1, 2, 4>), dim3(grid), dim3(512), (wgrad_v3_lds<1, 2, 4>()), s, a);
  else if (v3) hipLaunchKernelGGL((conv1d_wgrad_bf16_v3_kernel<3, 1, 2>), dim3(grid), dim3(512), (wgrad_v3_lds<3, 1, 2>()), s, a);
  else if (KW == 3) hipLaunchKernelGGL((conv1d_wgrad_bf16_v2_kernel<3, 2>), dim3(grid), dim3(512), (wgrad_v2_lds<3, 2>()), s, a);
  else hipLaunchKernelGGL((conv1d_wgrad_bf16_v2_kernel<1, 4>), dim3(grid), dim3(512), (wgrad_v2_lds<1, 4>()), s, a);
  int rc = check_launch("alvq_conv1d_wgrad_bf16(v2)");
  if (rc) return rc;
  if (accumulate == ALVQ_WGRAD_DEFER) return ALVQ_OK;   // the caller sums the partials later (alvq_wgrad_reduce_batch)
  wgrad_reduce_launch((const float*)workspace, dw, a.splits, KW, M, C, w_layout, accumulate, s, out_scale);
  if (dbias)
    hipLaunchKernelGGL(wgrad_bias_reduce_kernel, dim3((M + 255) / 256), dim3(256), 0, s, (const float*)bias_partial, dbias,
                       a.splits, Mp, M, accumulate, out_scale);
  return check_launch("alvq_conv1d_wgrad_bf16(v2)/reduce");
}

}  // namespace alvq
