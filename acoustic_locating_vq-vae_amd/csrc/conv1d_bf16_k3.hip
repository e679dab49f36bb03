// bf16 convolution, width-3 wide-layer kernel: the same 256 out-channels x 256 rows workgroup tile and 8-wave
// fragment layout as conv1d_bf16_v2.hip, but the K loop walks CHUNKS of 32 channels and serves all three taps of a
// chunk from ONE activation slab: rows r0-1 .. r0+256 are staged once (258 x 64 B) and tap t simply reads the slab
// t rows further down.  Per chunk the workgroup therefore moves 3 x 16 KB of weights + 16.1 KB of activations
// through LDS-DMA instead of 3 x (16 + 16) KB -- a third less L2->LDS traffic for the same 96 MFMAs per wave.
//
// Pipeline: two LDS stages of 65 KB.  Chunk c lives in stage c&1 and is processed as three tap phases of 32 MFMAs,
// each phase reading the next phase's fragments between and after its MFMA halves:
//   tap 0:  DMA(part B of chunk c+1);  MFMAs
//   tap 1:  MFMAs;  s_waitcnt vmcnt(0) lgkmcnt(0);  s_barrier     <- chunk c+1 landed, every read of stage c&1 done
//   tap 2:  DMA(part A of chunk c+2, into the stage this chunk is leaving);  MFMAs (fragments of chunk c+1, tap 0
//           are fetched from the other stage meanwhile)
// Part A = the weight slabs of taps 0 and 1 (4 DMA pieces per wave), part B = tap 2's weights and the activation
// slab (4 pieces, wave 7 one more for the two halo rows).  One barrier per 96 MFMAs; every DMA piece has more than
// a full tap phase to land.  LDS rows are 64 B; the weight slabs carry the v2 slot swizzle (slot = group ^
// {0,3,2,1}[(row>>2)&3]), the activation slab one that stays conflict-free when read 1 or 2 rows further down
// (slot = group ^ {0,2,0,2}[(row>>2)&3], derivation at lane_off_x below).
#include <stdlib.h>

#include "alvq_common.h"
#include "bf16_common.h"
#include "conv1d_bf16_tile256.h"

namespace alvq {

constexpr int K3_XSLAB = 17 * 1024;                       // 272 rows x 64 B (258 used)
constexpr int K3_STAGE = 3 * V2_HALF + K3_XSLAB;          // 66560 B
constexpr int K3_LDS = 2 * K3_STAGE;                      // 133120 B
static_assert(V2_EPI_LDS <= K3_LDS, "C slab must fit");

template <int OUT, int F16 = 0>
__global__ __launch_bounds__(512, 2) void conv1d_bf16_k3_kernel(ConvBArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const int wm0 = (wave >> 2) * 128, wn0 = (wave & 3) * 64;

  const int tile = xcd_remap(blockIdx.x, a.mtiles * a.rtiles);      // same tile order as the v2 kernel
  const int m0 = (tile % a.mtiles) * V2_M;
  const int r0 = (tile / a.mtiles) * V2_R;
  const int Cp = a.Cp;

  // ---- DMA source addressing: a piece is 16 rows x 64 B; lane i -> row i>>2, slot i&3 <- channel group
  // (i&3) ^ h[(row>>2)&3]
  const int hsel = (lane >> 4) & 3;
  const int hval = (hsel == 0) ? 0 : (4 - hsel);              // {0,3,2,1}
  const int srow = lane >> 2, sgrp = (lane & 3) ^ hval;
  // One 32-bit per-lane byte offset serves every piece; the rest of a piece's source address is wave-uniform and
  // goes into the instruction's SGPR base.  Written as inline asm: through the builtin, hipcc hoists
  // (lane offset + k * 16 rows) into five loop-invariant 64-bit VGPR pairs, which this kernel has no room for.
  const unsigned lane_off = (unsigned)(srow * Cp + sgrp * 8) * 2u;
  // The ACTIVATION slab uses its own slot swizzle, slot = group ^ {0,2,0,2}[(row>>2)&3]: the taps read it at row
  // offsets 0, 1, 2, and under the weight slabs' {0,3,2,1} the shifted reads collide two-fold in two of every sixteen
  // lanes of a ds_read_b128 bank group (18 % of this kernel's LDS cycles were conflicts).  A 16-lane group takes four
  // row quads with channel groups (a, b, b, a), b = a ^ 1; rows shifted across a quad boundary keep their lane's group
  // but take the next quad's key, so the key f must make both {f0, f3, f1^1, f2^1} and {f0, f1, f2^1, f3^1}
  // permutations of 0..3 -- (0, 2, 0, 2) does, (0, 3, 2, 1) only the first.
  const unsigned lane_off_x = (unsigned)(srow * Cp + ((lane & 3) ^ ((hsel & 1) * 2)) * 8) * 2u;
  const long tap_w = (long)a.Mp128 * Cp;
  const long row16 = (long)Cp * 32;                            // bytes per 16 rows
  const unsigned lds0 = (unsigned)(unsigned long)((__attribute__((address_space(3))) unsigned char*)lds);
  auto dma = [&](const char* sbase, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_off), "s"(sbase), "s"(lds_dst)
                 : "memory");
  };
  auto dma_x = [&](const char* sbase, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_off_x), "s"(sbase), "s"(lds_dst)
                 : "memory");
  };
  // part A: waves 0-3 stage tap 0, waves 4-7 tap 1; 64 weight rows (4 pieces) each
  const char* const wA = (const char*)(a.wp + (wave >> 2) * tap_w + ((long)m0 + (wave & 3) * 64) * Cp);
  const unsigned dA = lds0 + (wave >> 2) * V2_HALF + (wave & 3) * 4096;
  // part B: 32 rows (2 pieces) of tap 2 and 32 rows of the activation slab per wave; slab row 0 = matrix row r0-1
  const char* const wB = (const char*)(a.wp + 2 * tap_w + ((long)m0 + wave * 32) * Cp);
  const char* const xB = (const char*)(a.x + ((long)r0 - 1 + wave * 32) * Cp);
  const unsigned dBw = lds0 + 2 * V2_HALF + wave * 2048, dBx = lds0 + 3 * V2_HALF + wave * 2048;

  auto issueA = [&](int c) {
    const unsigned dst = (c & 1) * K3_STAGE + dA;
    const char* ws = wA + c * (V2_K * 2);
    dma(ws, dst);
    dma(ws + row16, dst + 1024);
    dma(ws + 2 * row16, dst + 2048);
    dma(ws + 3 * row16, dst + 3072);
  };
  auto issueB = [&](int c) {
    const unsigned st = (c & 1) * K3_STAGE;
    const char* ws = wB + c * (V2_K * 2);
    const char* xs = xB + c * (V2_K * 2);
    dma(ws, st + dBw);
    dma(ws + row16, st + dBw + 1024);
    dma_x(xs, st + dBx);
    dma_x(xs + row16, st + dBx + 1024);
    if (wave == 7 && srow < 2) dma_x(xs + 2 * row16, st + dBx + 2048);   // halo: slab rows 256, 257
  };

  // ---- fragment read addressing: weights as in v2; activations per tap (slab row = local row + tap)
  const int hl = (li >> 2) & 3;
  const int loffA = li * 64 + ((kq ^ (hl == 0 ? 0 : 4 - hl)) << 4);
  int loffX[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int r = li + t;
    loffX[t] = r * 64 + ((kq ^ (((r >> 2) & 1) * 2)) << 4);
  }
  const unsigned char* const abase = lds + wm0 * 64 + loffA;
  const unsigned char* const xbase = lds + 3 * V2_HALF + wn0 * 64;
  // Fragment registers: ONE set of weight fragments, refilled in place half by half (the four fragments an MFMA
  // half has consumed are dead, so the next phase's are read straight into them), and two sets of the four
  // activation fragments (both halves use them, so they are double-buffered): 64 VGPRs instead of 96.
  bf16x8_t fa[8], fb[2][4];
  auto rdA = [&](int half, int stage, int tap) {
    const unsigned char* pa = abase + stage * K3_STAGE + tap * V2_HALF;
#pragma unroll
    for (int mi = half * 4; mi < half * 4 + 4; ++mi) fa[mi] = *(const bf16x8_t*)(pa + mi * 1024);
  };
  auto rdB = [&](int q, int stage, int tap) {
    const unsigned char* pb = xbase + stage * K3_STAGE + loffX[tap];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) fb[q][ni] = *(const bf16x8_t*)(pb + ni * 1024);
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto mm = [&](int q, int half) {
#pragma unroll
    for (int mi = half * 4; mi < half * 4 + 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
        // tied inline asm rather than the builtin: hipcc does not tie the builtin's destination to its C operand
        // (the accumulators then wander through the register file and this kernel spills inside the loop)
        if (F16) asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[mi][ni]) : "v"(fa[mi]), "v"(fb[q][ni]));
        else asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[mi][ni]) : "v"(fa[mi]), "v"(fb[q][ni]));
  };

  const int nch = Cp / V2_K;        // chunks; always even (Cp % 64 == 0)
  const bool early = wave < 4;      // the two waves of a SIMD issue their DMA at different points of a phase

  // ---- prologue: chunk 0 complete, part A of chunk 1 in flight, fragments of (chunk 0, tap 0) in registers
  issueA(0);
  issueB(0);
  issueA(1);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  rdA(0, 0, 0);
  rdA(1, 0, 0);
  rdB(0, 0, 0);

  // One phase = the 32 MFMAs of (stage, tap) from fa / fb[Q]; between and after its halves the fragments of the
  // NEXT phase (NS, NT) are read.  PRE / MID: DMA issue slots of the early / late waves.
#define ALVQ_K3_PHASE(Q, NS, NT, DMA)                    \
  if (early) { DMA; }                                    \
  mm(Q, 0);                                              \
  __builtin_amdgcn_sched_barrier(0);                     \
  rdA(0, NS, NT);                                        \
  rdB((Q) ^ 1, NS, NT);                                  \
  __builtin_amdgcn_sched_barrier(0);                     \
  if (!early) { DMA; }                                   \
  mm(Q, 1);                                              \
  __builtin_amdgcn_sched_barrier(0);                     \
  rdA(1, NS, NT);                                        \
  __builtin_amdgcn_sched_barrier(0);

  // One chunk (stage S = c & 1, first phase on fb[Q0]); past the last chunk the reads fetch stale data nobody uses
#define ALVQ_K3_CHUNK(c, S, Q0)                                                   \
  ALVQ_K3_PHASE(Q0, S, 1, if ((c) + 1 < nch) issueB((c) + 1))                     \
  ALVQ_K3_PHASE((Q0) ^ 1, S, 2, (void)0)                                          \
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                     \
  __builtin_amdgcn_s_barrier();                                                   \
  ALVQ_K3_PHASE(Q0, (S) ^ 1, 0, if ((c) + 2 < nch) issueA((c) + 2))

  for (int c = 0; c < nch; c += 2) {
    ALVQ_K3_CHUNK(c, 0, 0)
    ALVQ_K3_CHUNK(c + 1, 1, 1)
  }
#undef ALVQ_K3_CHUNK
#undef ALVQ_K3_PHASE
  // the compiler's hazard recogniser does not see inside the asm MFMAs: cover the MFMA-result -> VALU-read wait
  // states by hand before the epilogue touches the accumulators
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  __syncthreads();   // OUT == 1: the C slab overlays the stages

  tile256_epilogue<OUT, F16>(a, acc, lds, m0, r0, wave, tid, li, kq, wm0);
}

int conv1d_bf16_k3_launch(const ConvBArgs& a_in, hipStream_t stream) {
  ConvBArgs a = a_in;
  const long rows = (long)a.rtiles * TB_R;     // caller computed rtiles in 128-row units; rows % 256 == 0
  a.rtiles = (int)(rows / V2_R);
  a.mtiles = (a.M + V2_M - 1) / V2_M;
  static DeviceOnce attr;
  if (attr.need()) {
    (void)hipFuncSetAttribute((const void*)conv1d_bf16_k3_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, K3_LDS);
    (void)hipFuncSetAttribute((const void*)conv1d_bf16_k3_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, K3_LDS);
    (void)hipFuncSetAttribute((const void*)conv1d_bf16_k3_kernel<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, K3_LDS);
    (void)hipFuncSetAttribute((const void*)conv1d_bf16_k3_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, K3_LDS);
  }
  const dim3 grid(a.rtiles * a.mtiles), block(512);
  if (a.elem) {
    if (a.y) hipLaunchKernelGGL((conv1d_bf16_k3_kernel<0, 1>), grid, block, K3_LDS, stream, a);
    else hipLaunchKernelGGL((conv1d_bf16_k3_kernel<1, 1>), grid, block, K3_LDS, stream, a);
  } else if (a.y) hipLaunchKernelGGL((conv1d_bf16_k3_kernel<0>), grid, block, K3_LDS, stream, a);
  else hipLaunchKernelGGL((conv1d_bf16_k3_kernel<1>), grid, block, K3_LDS, stream, a);
  return check_launch("alvq_conv1d_bf16(k3)");
}

}  // namespace alvq
