// bf16 convolution, wide-layer kernel: 256 out-channels x 256 rows per workgroup, 8 waves (512 threads), one
// workgroup per CU, each wave a 128 x 64 output block = 8 x 4 MFMA 16x16x32 fragments (128 accumulator VGPRs).
//
// The contraction is walked in K-tiles of (32 channels, one tap): a tap is nothing but a row offset into the
// NLC activation matrix, so every K-tile is a plain [256 x 32] x [32 x 256] product whose operands are two
// 16 KB slabs -- Wp[tap][m0..+255][c..c+31] and act[r0+tap-pad..+255][c..c+31] -- fetched by LDS-DMA
// (global_load_lds_dwordx4, 4 per wave per K-tile).  No im2col buffer exists; the three shifted views of the
// activation rows are served by the XCD's L2.
//
// Pipeline: a ring of 4 LDS stages (128 KB).  In iteration t every wave
//   1. issues the DMA for K-tile t+3 into the stage K-tile t-1 just left,
//   2. issues the ds_read_b128 fragment loads of K-tile t+1 into the second fragment register set,
//   3. runs the 32 MFMAs of K-tile t from the first set,
//   4. waits with a COUNTED s_waitcnt vmcnt(4) (K-tile t+3 may stay in flight; t+2 has landed) and meets the
//      other waves at one raw s_barrier.
// So DMA latency has two iterations (>= 1024 MFMA cycles) to hide, LDS read latency one, and the two waves of a
// SIMD alternate on the matrix pipe while the other's loads issue.  LDS rows are 64 B; the 16-B chunk of row r
// holding channel group g sits in slot g ^ h[(r>>2)&3], h = {0,3,2,1} (applied on the DMA source address and on
// the read address), which makes every ds_read_b128 fragment read bank-conflict-free.
#include "alvq_common.h"
#include "bf16_common.h"
#include "conv1d_bf16_tile256.h"

namespace alvq {

constexpr int V2_STAGE = 2 * V2_HALF;             // 32768 B
constexpr int V2_NSTAGE = 4;
constexpr int V2_LDS = V2_NSTAGE * V2_STAGE;      // 131072 B
static_assert(V2_EPI_LDS <= V2_LDS, "C slab must fit");

template <int OUT, int F16 = 0>
__global__ __launch_bounds__(512, 2) void conv1d_bf16_v2_kernel(ConvBArgs a, int KW) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int PAD = (KW - 1) / 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const int wm0 = (wave >> 2) * 128, wn0 = (wave & 3) * 64;

  // tile order: all m-tiles of a row tile are neighbours, and each XCD owns a contiguous run of tiles, so the
  // workgroups resident on an XCD at one time share both operands through its L2 (W: one miss per m-tile per
  // wave of workgroups; activation rows: one miss per row tile instead of one per m-tile)
  const int tile = xcd_remap(blockIdx.x, a.mtiles * a.rtiles);
  const int m0 = (tile % a.mtiles) * V2_M;
  const int r0 = (tile / a.mtiles) * V2_R;
  const int Cp = a.Cp;

  // ---- DMA source addressing: piece p (16 rows x 64 B); lane i -> row 16p + (i>>2), slot i&3, which must hold
  // channel group (i&3) ^ h[(row>>2)&3] = (i&3) ^ h[(i>>4)&3]
  const int hsel = (lane >> 4) & 3;
  const int hval = (hsel == 0) ? 0 : (4 - hsel);              // {0,3,2,1}
  const int srow = lane >> 2, sgrp = (lane & 3) ^ hval;
  const long lane_off = (long)srow * Cp + sgrp * 8;            // elements
  const u16* const wbase = a.wp + ((long)m0 + wave * 32) * Cp + lane_off;              // + tap*Mp*Cp + q*16*Cp + chunk*32
  const u16* const xbase = a.x + ((long)r0 - PAD + wave * 32) * Cp + lane_off;         // + tap*Cp      + q*16*Cp + chunk*32
  const long tap_w = (long)a.Mp128 * Cp;

  int is_chunk = 0, is_tap = 0;   // K-tile the next issue() will stage
  auto issue = [&](int stage) {
    unsigned char* dst = lds + stage * V2_STAGE + wave * 2048;
    const u16* ws = wbase + is_tap * tap_w + is_chunk * V2_K;
    const u16* xs = xbase + (long)is_tap * Cp + is_chunk * V2_K;
    glds16(ws, dst);
    glds16(ws + 16L * Cp, dst + 1024);
    glds16(xs, dst + V2_HALF);
    glds16(xs + 16L * Cp, dst + V2_HALF + 1024);
    if (++is_tap == KW) {
      is_tap = 0;
      ++is_chunk;
    }
  };

  // ---- fragment read addressing (same lane offset for both operands)
  const int loff = li * 64 + ((kq ^ ((((li >> 2) & 3) == 0) ? 0 : (4 - ((li >> 2) & 3)))) << 4);
  const unsigned char* const abase = lds + wm0 * 64 + loff;
  const unsigned char* const bbase = lds + V2_HALF + wn0 * 64 + loff;
  auto rd = [&](FragSet& f, int stage) {
    const unsigned char* pa = abase + stage * V2_STAGE;
    const unsigned char* pb = bbase + stage * V2_STAGE;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) f.a[mi] = *(const bf16x8_t*)(pa + mi * 1024);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) f.b[ni] = *(const bf16x8_t*)(pb + ni * 1024);
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // the 32 MFMAs of a K-tile in two halves, so the next K-tile's fragment reads can be issued between them:
  // they then have 16 MFMAs to land, and the reads a half consumes were issued 16 MFMAs + a barrier earlier
  auto mm = [&](const FragSet& f, int half) {
#pragma unroll
    for (int mi = half * 4; mi < half * 4 + 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
        acc[mi][ni] = elem_mfma16<F16>(f.a[mi], f.b[ni], acc[mi][ni]);
  };

  const int n = (Cp / V2_K) * KW;   // K-tiles; always even (Cp % 64 == 0)
  const bool early = wave < 4;
  FragSet f0, f1;

  // ---- prologue: K-tiles 0..2 in flight, fragments of K-tile 0 in f0, K-tile 1 landed
  issue(0);
  issue(1);
  if (n > 2) {
    issue(2);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  rd(f0, 0);
  if (n > 2) {
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();

  for (int t = 0; t < n; t += 2) {
    // ---- even K-tile t (fragments in f0)
    // stagger: the older four waves issue their DMA before the MFMAs, the younger four (their SIMD partners)
    // between the MFMA halves, so the two waves of a SIMD are not both stuck in DMA issue at the same time
    if (t + 3 < n && early) issue((t + 3) & 3);
    mm(f0, 0);
    __builtin_amdgcn_sched_barrier(0);   // pin: reads go BETWEEN the MFMA halves (hipcc otherwise hoists them
    rd(f1, (t + 1) & 3);                 // above all 32 MFMAs and then waits lgkmcnt(0) in front of the first one)
    __builtin_amdgcn_sched_barrier(0);
    if (t + 3 < n && !early) issue((t + 3) & 3);
    mm(f0, 1);
    if (t + 3 < n) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    // ---- odd K-tile t+1 (fragments in f1)
    if (t + 4 < n && early) issue((t + 4) & 3);
    mm(f1, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (t + 2 < n) rd(f0, (t + 2) & 3);
    __builtin_amdgcn_sched_barrier(0);
    if (t + 4 < n && !early) issue((t + 4) & 3);
    mm(f1, 1);
    if (t + 4 < n) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
  }

  tile256_epilogue<OUT, F16>(a, acc, lds, m0, r0, wave, tid, li, kq, wm0);
}

int conv1d_bf16_v2_launch(const ConvBArgs& a_in, int KW, hipStream_t stream) {
  ConvBArgs a = a_in;
  const long rows = (long)a.rtiles * TB_R;     // caller computed rtiles in 128-row units; rows % 256 == 0
  a.rtiles = (int)(rows / V2_R);
  a.mtiles = (a.M + V2_M - 1) / V2_M;
  static DeviceOnce attr;
  if (attr.need()) {
    (void)hipFuncSetAttribute((const void*)conv1d_bf16_v2_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, V2_LDS);
    (void)hipFuncSetAttribute((const void*)conv1d_bf16_v2_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, V2_LDS);
    (void)hipFuncSetAttribute((const void*)conv1d_bf16_v2_kernel<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, V2_LDS);
    (void)hipFuncSetAttribute((const void*)conv1d_bf16_v2_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, V2_LDS);
  }
  const dim3 grid(a.rtiles * a.mtiles), block(512);
  if (a.elem) {
    if (a.y) hipLaunchKernelGGL((conv1d_bf16_v2_kernel<0, 1>), grid, block, V2_LDS, stream, a, KW);
    else hipLaunchKernelGGL((conv1d_bf16_v2_kernel<1, 1>), grid, block, V2_LDS, stream, a, KW);
  } else if (a.y) hipLaunchKernelGGL((conv1d_bf16_v2_kernel<0>), grid, block, V2_LDS, stream, a, KW);
  else hipLaunchKernelGGL((conv1d_bf16_v2_kernel<1>), grid, block, V2_LDS, stream, a, KW);
  return check_launch("alvq_conv1d_bf16(v2)");
}

}  // namespace alvq
