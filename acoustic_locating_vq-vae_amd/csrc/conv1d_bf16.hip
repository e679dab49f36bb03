// bf16-MFMA 1-D convolution family for gfx950 (throughput path): forward / data-grad / ConvTranspose on
// v_mfma_f32_16x16x32_bf16 with fp32 accumulation.  This file holds the layout helpers, the 128 x 128-tile kernel for
// narrow layers and the entry points; wide layers go to conv1d_bf16_k3.hip / conv1d_bf16_v2.hip, the weight gradient
// to conv1d_wgrad_bf16_v2.hip, weight packing to pack_weights.hip.
//
// Layout in HBM ("NLC-padded"): an activation is a 2-D bf16 matrix act[row][Cp], channels contiguous,
//   row(b,l) = 1 + b*(L+1) + l,  Cp = C rounded up to 64,
// so the contraction (channel) axis is contiguous for BOTH MFMA operands and a k=3 tap is a plain row shift.
// Row 0 and the row after each sample (l == L) are zero "gap" rows: a halo never reads a neighbouring sample
// and the whole batch is one uniform matrix -- tiles need no per-sample logic.  Every producer writes zeros to
// gap rows, to rows beyond the batch (the buffer is rounded up to 128 rows) and to padded channels.  Eight
// readable guard rows precede row 0 and follow the last row (contents irrelevant: they only feed gap rows).
// Weights are re-packed per step to Wp[tap][Mp128][Cp] bf16 (K-contiguous rows; ConvTranspose / data-grad
// orientation is resolved by the packer), 6.3 MB for a 1024x1024x3 layer.
//
// Workgroup = 128 rows x 128 out-channels, 4 waves x (4x4) MFMA fragments.  The K loop steps over
// (64-channel chunk, tap): the input tile (136 rows x 64 ch, 17 KB) is staged once per chunk and re-read at
// row offsets 0/1/2 by the three taps (im2col-free); the weight tile (128 x 64, 16 KB) is staged per step.  All
// staging is LDS-DMA (global_load_lds_dwordx4, 1 KB per wave-instruction) into a lane-linear image whose 16-B
// slots are XOR-swizzled by (row & 7) on the SOURCE address, which makes every ds_read_b128 fragment read
// conflict-free.  Double-buffered (66 KB) -> two workgroups per CU.
#include <stdlib.h>

#include "alvq_common.h"
#include "bf16_common.h"
#include "conv1d_bf16_tile256.h"

namespace alvq {

template <int KW, int OUT, int F16 = 0>
__global__ __launch_bounds__(256, 2) void conv1d_bf16_kernel(ConvBArgs a) {
  constexpr int PAD = (KW - 1) / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* const Xb = lds;                 // 2 x XBYTES
  unsigned char* const Wb = lds + 2 * XBYTES;    // 2 x WBYTES

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;

  const int tile = xcd_remap(blockIdx.x, a.mtiles * a.rtiles);
  const int m0 = (tile % a.mtiles) * TB_M;
  const int r0 = (tile / a.mtiles) * TB_R;
  const int Cp = a.Cp;

  // ---- staging: lane i of a piece writes LDS row (i>>3), 16-B slot (i&7); it fetches source chunk slot^(row&7)
  const int srow = lane >> 3;
  const int schunk = (lane & 7) ^ srow;
  const u16* xsrc = a.x + ((long)(r0 - PAD + srow)) * Cp + schunk * 8;      // + piece*8 rows + chunk*64
  const u16* wsrc = a.wp + ((long)(m0 + srow)) * Cp + schunk * 8;          // + tap*Mp128 rows + piece*8 rows + chunk*64

  auto stage_x = [&](int buf, int chunk) {
    unsigned char* dst = Xb + buf * XBYTES;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      const int p = wave + 4 * q;
      if (p < XROWS / 8) glds16(xsrc + ((long)p * 8) * Cp + chunk * TB_K, dst + p * 1024);
    }
  };
  auto stage_w = [&](int buf, int tap, int chunk) {
    unsigned char* dst = Wb + buf * WBYTES;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int p = wave * 4 + q;
      glds16(wsrc + ((long)tap * a.Mp128 + p * 8) * Cp + chunk * TB_K, dst + p * 1024);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nchunks = Cp / TB_K;
  const int nsteps = nchunks * KW;
  stage_x(0, 0);
  stage_w(0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int chunk = 0, tap = 0;
  for (int step = 0; step < nsteps; ++step) {
    int ntap = tap + 1, nchunk = chunk;
    if (ntap == KW) {
      ntap = 0;
      nchunk = chunk + 1;
    }
    if (step + 1 < nsteps) {
      stage_w((step + 1) & 1, ntap, nchunk);
      if (ntap == 0) stage_x(nchunk & 1, nchunk);
    }
    const unsigned char* Wc = Wb + (step & 1) * WBYTES;
    const unsigned char* Xc = Xb + (chunk & 1) * XBYTES;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8_t af[4], bfr[4];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const int row = wm0 + mi * 16 + li;
        af[mi] = *(const bf16x8_t*)(Wc + row * 128 + (((s * 4 + kq) ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int row = wn0 + ni * 16 + li + tap;
        bfr[ni] = *(const bf16x8_t*)(Xc + row * 128 + (((s * 4 + kq) ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[mi][ni] = elem_mfma16<F16>(af[mi], bfr[ni], acc[mi][ni]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    tap = ntap;
    chunk = nchunk;
  }

  if (OUT == 0) {   // NLC bf16: straight from the accumulators (conv1d_bf16_tile256.h)
    wave_epilogue_bf16<4, 4, F16>(a, acc, m0, r0, li, kq, wm0, wn0);
    return;
  }
  const float oscale = a.out_scale ? *a.out_scale : 1.f;
  // ---- OUT == 1, step 1: D[i = m][j = row] -> fp32 C tile Cs[row][m] (4 consecutive m per lane = one 16-B write)
  float* Cs = (float*)lds;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int rl = wn0 + ni * 16 + li, ml = wm0 + mi * 16 + kq * 4;
      *(f32x4*)(Cs + rl * CS + ml) = acc[mi][ni];
    }
  __syncthreads();

  const int Lp1 = a.L + 1, ndata = a.B * Lp1;
  {
    // ---- step 2 (NCL fp32, bias only): lane = row (coalesced along l), loop over channels
    const int rl = tid & 127, row = r0 + rl;
    int b, l;
    const bool ok = row_valid(row, Lp1, ndata, &b, &l);
    if (ok) {
      for (int ml = tid >> 7; ml < TB_M; ml += 2) {
        const int m = m0 + ml;
        if (m >= a.M) break;
        a.y_ncl[((long)b * a.M + m) * a.L + l] = (Cs[rl * CS + ml] + (a.bias ? a.bias[m] : 0.f)) * oscale;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------- helpers
// (B,C,L) fp32 -> NLC-padded bf16 [rows_total][Cp] (gap rows, tail rows and padded channels zero).
template <int F16 = 0>
__global__ __launch_bounds__(256) void ncl_to_nlc_kernel(const float* x, u16* y, int B, int C, int L, int Cp, int rows_total,
                                                         const float* scale = nullptr) {
  __shared__ float tile[32][33];
  elem_saturate<F16>();
  const float sc = scale ? *scale : 1.f;
  const int ct = Cp / 32, rt = (rows_total + 31) / 32;
  const int r0 = (blockIdx.x / ct) * 32, c0 = (blockIdx.x % ct) * 32;
  (void)rt;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int Lp1 = L + 1, ndata = B * Lp1;
#pragma unroll
  for (int i = 0; i < 4; ++i) {  // read: lanes along rows (l contiguous in the source)
    const int c = c0 + ty + 8 * i, row = r0 + tx;
    int b, l;
    const bool ok = row_valid(row, Lp1, ndata, &b, &l) && c < C;
    tile[ty + 8 * i][tx] = ok ? x[((long)b * C + c) * L + l] * sc : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {  // write: lanes along channels
    const int row = r0 + ty + 8 * i, c = c0 + tx;
    if (row < rows_total) y[(long)row * Cp + c] = F16 ? (u16)(elem_pk<1>(tile[tx][ty + 8 * i], 0.f) & 0xffffu) : f2bf(tile[tx][ty + 8 * i]);
  }
}

// NLC-padded bf16 -> (B,C,L) fp32 (for module outputs that leave the bf16 pipeline).
template <int F16 = 0>
__global__ __launch_bounds__(256) void nlc_to_ncl_kernel(const u16* x, float* y, int B, int C, int L, int Cp, int rows_total,
                                                         const float* scale = nullptr) {
  __shared__ float tile[32][33];
  const float sc = scale ? *scale : 1.f;
  const int ct = Cp / 32;
  const int r0 = (blockIdx.x / ct) * 32, c0 = (blockIdx.x % ct) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int Lp1 = L + 1, ndata = B * Lp1;
#pragma unroll
  for (int i = 0; i < 4; ++i) {  // read: lanes along channels
    const int row = r0 + ty + 8 * i, c = c0 + tx;
    tile[ty + 8 * i][tx] = row < rows_total ? elem2f<F16>(x[(long)row * Cp + c]) * sc : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {  // write: lanes along rows (l contiguous in the destination)
    const int c = c0 + ty + 8 * i, row = r0 + tx;
    int b, l;
    if (row_valid(row, Lp1, ndata, &b, &l) && c < C) y[((long)b * C + c) * L + l] = tile[tx][ty + 8 * i];
  }
}

// out = mask > 0 ? dy : 0 on NLC bf16 or fp16 buffers (whole padded matrix): a positive bf16 / fp16 is a positive int16
__global__ __launch_bounds__(256) void relu_mask_bf16_kernel(const u16* dy, const u16* t, u16* out, long n8) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n8; e += (long)gridDim.x * 256) {
    const u16x8 d = ((const u16x8*)dy)[e], m = ((const u16x8*)t)[e];
    u16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (short)m[i] > 0 ? d[i] : (u16)0;
    ((u16x8*)out)[e] = o;
  }
}

}  // namespace alvq

using namespace alvq;

static inline int pad_to(int x, int q) { return (x + q - 1) / q * q; }

extern "C" int64_t alvq_nlc_rows(int B, int L) { return (B <= 0 || L <= 0) ? -1 : (int64_t)pad_to(1 + B * (L + 1), NLC_ROW_PAD); }
extern "C" int alvq_nlc_channels(int C) { return C <= 0 ? -1 : pad_to(C, TB_K); }
extern "C" int alvq_nlc_guard_rows(void) { return GUARD_ROWS; }

extern "C" int64_t alvq_packed_weight_elems(int M, int C, int KW) {
  if (M <= 0 || C <= 0 || (KW != 1 && KW != 3)) return -1;
  return (int64_t)KW * pad_to(M, WP_ROWS) * pad_to(C, TB_K);
}

extern "C" int alvq_pack_weight_bf16(const float* w, void* wp, int M, int C, int KW, int w_layout, void* stream) {
  const alvq_pack_desc d{w, wp, M, C, KW, w_layout};     // one-descriptor batch (pack_weights.hip validates)
  return alvq_pack_weights_bf16_batch(&d, 1, 1, stream);
}

extern "C" int alvq_ncl_to_nlc_bf16(const float* x, void* y, int B, int C, int L, void* stream) {
  ALVQ_REQUIRE(x && y, ALVQ_EINVAL, "alvq_ncl_to_nlc_bf16: null pointer");
  ALVQ_REQUIRE(B > 0 && C > 0 && L > 0, ALVQ_EINVAL, "alvq_ncl_to_nlc_bf16: bad dims");
  const int Cp = pad_to(C, TB_K), rows = (int)alvq_nlc_rows(B, L);
  hipLaunchKernelGGL(ncl_to_nlc_kernel<0>, dim3((rows / 32) * (Cp / 32)), dim3(256), 0, (hipStream_t)stream, x, (u16*)y, B, C,
                     L, Cp, rows, (const float*)nullptr);
  return check_launch("alvq_ncl_to_nlc_bf16");
}

extern "C" int alvq_nlc_to_ncl_f32(const void* x, float* y, int B, int C, int L, void* stream) {
  ALVQ_REQUIRE(x && y, ALVQ_EINVAL, "alvq_nlc_to_ncl_f32: null pointer");
  ALVQ_REQUIRE(B > 0 && C > 0 && L > 0, ALVQ_EINVAL, "alvq_nlc_to_ncl_f32: bad dims");
  const int Cp = pad_to(C, TB_K), rows = (int)alvq_nlc_rows(B, L);
  hipLaunchKernelGGL(nlc_to_ncl_kernel<0>, dim3((rows / 32) * (Cp / 32)), dim3(256), 0, (hipStream_t)stream, (const u16*)x, y, B,
                     C, L, Cp, rows, (const float*)nullptr);
  return check_launch("alvq_nlc_to_ncl_f32");
}

extern "C" int alvq_relu_mask_bf16(const void* dy, const void* t, void* out, int64_t n, void* stream) {
  ALVQ_REQUIRE(dy && t && out, ALVQ_EINVAL, "alvq_relu_mask_bf16: null pointer");
  ALVQ_REQUIRE(n > 0 && n % 8 == 0, ALVQ_EINVAL, "alvq_relu_mask_bf16: n must be a positive multiple of 8");
  long g = (n / 8 + 1023) / 1024;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(relu_mask_bf16_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, (const u16*)dy, (const u16*)t,
                     (u16*)out, (long)(n / 8));
  return check_launch("alvq_relu_mask_bf16");
}

// the 16-bit convolution behind alvq_conv1d_bf16 (elem 0) and alvq_conv1d_f16 (elem 1)
static int conv1d_16bit(int elem, const float* out_scale, const void* x, const void* wp, const float* bias, const void* skip1,
                        const void* skip2, const void* mask, const void* post, void* y, void* y2, float* y_ncl, int B, int C, int M,
                        int L, int KW, int relu, const void* mask_bits, void* relu_bits_out, void* stream) {
  const char* const who = elem ? "alvq_conv1d_f16" : "alvq_conv1d_bf16";
  ALVQ_REQUIRE(x && wp && (y || y_ncl), ALVQ_EINVAL, "%s: null x/wp/y", who);
  ALVQ_REQUIRE(!(y && y_ncl), ALVQ_EINVAL, "%s: choose one of y (NLC) and y_ncl (NCL fp32)", who);
  ALVQ_REQUIRE(B > 0 && C > 0 && M > 0 && L > 0, ALVQ_EINVAL, "%s: bad dims", who);
  ALVQ_REQUIRE(KW == 1 || KW == 3, ALVQ_EUNSUPPORTED, "%s: KW=%d (only 1 and 3)", who, KW);
  ALVQ_REQUIRE((y2 == nullptr) == (post == nullptr), ALVQ_EINVAL, "%s: y2 and post go together", who);
  ALVQ_REQUIRE(!y_ncl || (!skip1 && !skip2 && !mask && !post && !relu), ALVQ_EUNSUPPORTED,
               "%s: the NCL fp32 epilogue fuses bias only", who);
  ALVQ_REQUIRE((long)B * (L + 1) < (1L << 30), ALVQ_EUNSUPPORTED, "%s: problem too large", who);
  ConvBArgs a{(const u16*)x, (const u16*)wp, bias, (const u16*)skip1, (const u16*)skip2, (const u16*)mask, (const u16*)post,
              (u16*)y, (u16*)y2, y_ncl, B, L, pad_to(C, TB_K), M, pad_to(M, TB_K), pad_to(M, WP_ROWS), relu,
              (int)(alvq_nlc_rows(B, L) / TB_R), pad_to(M, TB_M) / TB_M, (const unsigned char*)mask_bits,
              (unsigned char*)relu_bits_out};
  ALVQ_REQUIRE(!(mask && mask_bits), ALVQ_EINVAL, "%s: pass the mask as a tensor or as bits, not both", who);
  ALVQ_REQUIRE(!y_ncl || (!mask_bits && !relu_bits_out), ALVQ_EUNSUPPORTED, "%s: sign bits go with the NLC output", who);
  hipStream_t s = (hipStream_t)stream;
  a.relu = relu ? 1 : 0;
  a.elem = elem;
  a.out_scale = out_scale;
  a.range_flag = elem ? fx_range_flag_ptr() : nullptr;
  // Wide layers: 256 x 256 tiles whenever the 256-wide m-tile is (nearly) full -- the width-3 kernel with the shared
  // activation slab, or the generic one; narrow or ragged M (128, 192, 201, 64, 1) stays on 128 x 128 tiles, which
  // waste less there and give more workgroups.  ALVQ_CONV_V2=0 / ALVQ_CONV_K3=0 force the fallbacks (used by
  // tools/ab_kernels.py to cross-check and compare the kernels on one device).
  const long use_v2 = option(OPT_CONV_V2), use_k3 = option(OPT_CONV_K3);
  // ... and only when that gives every CU a workgroup: a small problem (the RIR config: 26 row tiles x 4 m-tiles = 104
  // workgroups of 256 x 256 for 256 CUs) is better served by four times as many 128 x 128 tiles at two per CU.
  // Option "wide_min_tiles" (default 192; the unit tests lower it to drive the wide kernels with small problems).
  const long tiles256 = (alvq_nlc_rows(B, L) / 256) * (pad_to(M, 256) / 256);
  const long min_tiles = option(OPT_WIDE_MIN_TILES);
  if (use_v2 && pad_to(M, 256) - M <= 32 && tiles256 >= min_tiles)
    return (KW == 3 && use_k3) ? conv1d_bf16_k3_launch(a, s) : conv1d_bf16_v2_launch(a, KW, s);
  static DeviceOnce attr;
  if (attr.need()) {
    (void)hipFuncSetAttribute((const void*)conv1d_bf16_kernel<3, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)conv1d_bf16_kernel<3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)conv1d_bf16_kernel<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)conv1d_bf16_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)conv1d_bf16_kernel<3, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)conv1d_bf16_kernel<3, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)conv1d_bf16_kernel<1, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)conv1d_bf16_kernel<1, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  }
  const dim3 grid(a.rtiles * a.mtiles), block(256);
  if (elem) {
    if (KW == 3) {
      if (y) hipLaunchKernelGGL((conv1d_bf16_kernel<3, 0, 1>), grid, block, LDS_BYTES, s, a);
      else hipLaunchKernelGGL((conv1d_bf16_kernel<3, 1, 1>), grid, block, LDS_BYTES, s, a);
    } else {
      if (y) hipLaunchKernelGGL((conv1d_bf16_kernel<1, 0, 1>), grid, block, LDS_BYTES, s, a);
      else hipLaunchKernelGGL((conv1d_bf16_kernel<1, 1, 1>), grid, block, LDS_BYTES, s, a);
    }
  } else if (KW == 3) {
    if (y) hipLaunchKernelGGL((conv1d_bf16_kernel<3, 0>), grid, block, LDS_BYTES, s, a);
    else hipLaunchKernelGGL((conv1d_bf16_kernel<3, 1>), grid, block, LDS_BYTES, s, a);
  } else {
    if (y) hipLaunchKernelGGL((conv1d_bf16_kernel<1, 0>), grid, block, LDS_BYTES, s, a);
    else hipLaunchKernelGGL((conv1d_bf16_kernel<1, 1>), grid, block, LDS_BYTES, s, a);
  }
  return check_launch("alvq_conv1d_bf16");
}

extern "C" int alvq_conv1d_bf16(const void* x, const void* wp, const float* bias, const void* skip1, const void* skip2,
                                const void* mask, const void* post, void* y, void* y2, float* y_ncl, int B, int C, int M,
                                int L, int KW, int relu, const void* mask_bits, void* relu_bits_out, void* stream) {
  return conv1d_16bit(0, nullptr, x, wp, bias, skip1, skip2, mask, post, y, y2, y_ncl, B, C, M, L, KW, relu, mask_bits, relu_bits_out,
                      stream);
}

// ---- fp16 element type (the backward pass of the f16mx_hb mode; include/alvq.h)
extern "C" int alvq_conv1d_f16(const void* x, const void* wp, const float* bias, const void* skip1, const void* skip2,
                               const void* mask, const void* post, void* y, void* y2, float* y_ncl, int B, int C, int M, int L,
                               int KW, int relu, const void* mask_bits, void* relu_bits_out, const float* out_scale, void* stream) {
  return conv1d_16bit(1, out_scale, x, wp, bias, skip1, skip2, mask, post, y, y2, y_ncl, B, C, M, L, KW, relu, mask_bits,
                      relu_bits_out, stream);
}

extern "C" int alvq_ncl_to_nlc_f16(const float* x, void* y, int B, int C, int L, const float* scale, void* stream) {
  ALVQ_REQUIRE(x && y, ALVQ_EINVAL, "alvq_ncl_to_nlc_f16: null pointer");
  ALVQ_REQUIRE(B > 0 && C > 0 && L > 0, ALVQ_EINVAL, "alvq_ncl_to_nlc_f16: bad dims");
  const int Cp = pad_to(C, TB_K), rows = (int)alvq_nlc_rows(B, L);
  hipLaunchKernelGGL(ncl_to_nlc_kernel<1>, dim3((rows / 32) * (Cp / 32)), dim3(256), 0, (hipStream_t)stream, x, (u16*)y, B, C, L, Cp,
                     rows, scale);
  return check_launch("alvq_ncl_to_nlc_f16");
}

extern "C" int alvq_nlc_to_ncl_f16(const void* x, float* y, int B, int C, int L, const float* scale, void* stream) {
  ALVQ_REQUIRE(x && y, ALVQ_EINVAL, "alvq_nlc_to_ncl_f16: null pointer");
  ALVQ_REQUIRE(B > 0 && C > 0 && L > 0, ALVQ_EINVAL, "alvq_nlc_to_ncl_f16: bad dims");
  const int Cp = pad_to(C, TB_K), rows = (int)alvq_nlc_rows(B, L);
  hipLaunchKernelGGL(nlc_to_ncl_kernel<1>, dim3((rows / 32) * (Cp / 32)), dim3(256), 0, (hipStream_t)stream, (const u16*)x, y, B, C, L,
                     Cp, rows, scale);
  return check_launch("alvq_nlc_to_ncl_f16");
}

// workspace = the split partials of the weight gradient, then 64 * pad64(M) floats of bias-gradient partials
static int64_t wgrad_bias_offset(int rows, int C, int M, int KW) { return conv1d_wgrad_bf16_v2_workspace_bytes(rows, C, M, KW); }

extern "C" int64_t alvq_conv1d_wgrad_bf16_workspace_bytes(int B, int C, int M, int L, int KW) {
  if (B <= 0 || C <= 0 || M <= 0 || L <= 0 || (KW != 1 && KW != 3)) return -1;
  return wgrad_bias_offset((int)alvq_nlc_rows(B, L), C, M, KW) + (int64_t)64 * pad_to(M, TB_K) * 4;
}

extern "C" int64_t alvq_conv1d_wgrad_bf16_bias_offset(int B, int C, int M, int L, int KW) {
  if (B <= 0 || C <= 0 || M <= 0 || L <= 0 || (KW != 1 && KW != 3)) return -1;
  return wgrad_bias_offset((int)alvq_nlc_rows(B, L), C, M, KW);
}

extern "C" int alvq_conv1d_wgrad_bf16_splits(int B, int C, int M, int L, int KW, int nseg, int with_bias) {
  if (B <= 0 || C <= 0 || M <= 0 || L <= 0 || (KW != 1 && KW != 3) || nseg < 1 || nseg > 4) return -1;
  return conv1d_wgrad_bf16_v2_splits((int)alvq_nlc_rows(B, L), C, M, KW, nseg, with_bias != 0);
}

extern "C" int alvq_conv1d_wgrad_bf16(const void* dy, const void* x, float* dw, float* dbias, void* workspace, int B, int C,
                                      int M, int L, int KW, int w_layout, int accumulate, void* stream) {
  ALVQ_REQUIRE(dy && x && (dw || accumulate == ALVQ_WGRAD_DEFER) && workspace, ALVQ_EINVAL, "alvq_conv1d_wgrad_bf16: null pointer");
  ALVQ_REQUIRE(B > 0 && C > 0 && M > 0 && L > 0, ALVQ_EINVAL, "alvq_conv1d_wgrad_bf16: bad dims");
  ALVQ_REQUIRE(KW == 1 || KW == 3, ALVQ_EUNSUPPORTED, "alvq_conv1d_wgrad_bf16: KW=%d (only 1 and 3)", KW);
  ALVQ_REQUIRE(w_layout == ALVQ_W_OIK || w_layout == ALVQ_W_IOK, ALVQ_EINVAL, "alvq_conv1d_wgrad_bf16: w_layout");
  const int rows = (int)alvq_nlc_rows(B, L);
  // the bias gradient rides in the same launch (column sums of dY by an all-ones MFMA operand)
  return conv1d_wgrad_bf16_v2_launch(&dy, &x, 1, dw, workspace, rows, C, M, KW, w_layout, accumulate, (hipStream_t)stream,
                                     dbias, (float*)((char*)workspace + wgrad_bias_offset(rows, C, M, KW)));
}

extern "C" int alvq_conv1d_wgrad_f16(const void* dy, const void* x, float* dw, float* dbias, void* workspace, int B, int C, int M,
                                     int L, int KW, int w_layout, int accumulate, const float* inv_scale, void* stream) {
  ALVQ_REQUIRE(dy && x && (dw || accumulate == ALVQ_WGRAD_DEFER) && workspace, ALVQ_EINVAL, "alvq_conv1d_wgrad_f16: null pointer");
  ALVQ_REQUIRE(B > 0 && C > 0 && M > 0 && L > 0, ALVQ_EINVAL, "alvq_conv1d_wgrad_f16: bad dims");
  ALVQ_REQUIRE(KW == 1 || KW == 3, ALVQ_EUNSUPPORTED, "alvq_conv1d_wgrad_f16: KW=%d (only 1 and 3)", KW);
  ALVQ_REQUIRE(w_layout == ALVQ_W_OIK || w_layout == ALVQ_W_IOK, ALVQ_EINVAL, "alvq_conv1d_wgrad_f16: w_layout");
  const int rows = (int)alvq_nlc_rows(B, L);
  return conv1d_wgrad_bf16_v2_launch(&dy, &x, 1, dw, workspace, rows, C, M, KW, w_layout, accumulate, (hipStream_t)stream, dbias,
                                     (float*)((char*)workspace + wgrad_bias_offset(rows, C, M, KW)), 1, inv_scale);
}

extern "C" int alvq_conv1d_wgrad_f16_multi(const void* const* dy, const void* const* x, int nseg, float* dw, void* workspace, int B,
                                           int C, int M, int L, int KW, int w_layout, int accumulate, const float* inv_scale,
                                           void* stream) {
  ALVQ_REQUIRE(dy && x && (dw || accumulate == ALVQ_WGRAD_DEFER) && workspace, ALVQ_EINVAL, "alvq_conv1d_wgrad_f16_multi: null pointer");
  ALVQ_REQUIRE(nseg >= 1 && nseg <= 4, ALVQ_EUNSUPPORTED, "alvq_conv1d_wgrad_f16_multi: nseg=%d (1..4)", nseg);
  for (int i = 0; i < nseg; ++i) ALVQ_REQUIRE(dy[i] && x[i], ALVQ_EINVAL, "alvq_conv1d_wgrad_f16_multi: null segment %d", i);
  ALVQ_REQUIRE(B > 0 && C > 0 && M > 0 && L > 0, ALVQ_EINVAL, "alvq_conv1d_wgrad_f16_multi: bad dims");
  ALVQ_REQUIRE(KW == 1 || KW == 3, ALVQ_EUNSUPPORTED, "alvq_conv1d_wgrad_f16_multi: KW=%d (only 1 and 3)", KW);
  ALVQ_REQUIRE(w_layout == ALVQ_W_OIK || w_layout == ALVQ_W_IOK, ALVQ_EINVAL, "alvq_conv1d_wgrad_f16_multi: w_layout");
  return conv1d_wgrad_bf16_v2_launch(dy, x, nseg, dw, workspace, (int)alvq_nlc_rows(B, L), C, M, KW, w_layout, accumulate,
                                     (hipStream_t)stream, nullptr, nullptr, 1, inv_scale);
}

extern "C" int alvq_conv1d_wgrad_bf16_multi(const void* const* dy, const void* const* x, int nseg, float* dw, void* workspace,
                                            int B, int C, int M, int L, int KW, int w_layout, int accumulate, void* stream) {
  ALVQ_REQUIRE(dy && x && (dw || accumulate == ALVQ_WGRAD_DEFER) && workspace, ALVQ_EINVAL, "alvq_conv1d_wgrad_bf16_multi: null pointer");
  ALVQ_REQUIRE(nseg >= 1 && nseg <= 4, ALVQ_EUNSUPPORTED, "alvq_conv1d_wgrad_bf16_multi: nseg=%d (1..4)", nseg);
  for (int i = 0; i < nseg; ++i) ALVQ_REQUIRE(dy[i] && x[i], ALVQ_EINVAL, "alvq_conv1d_wgrad_bf16_multi: null segment %d", i);
  ALVQ_REQUIRE(B > 0 && C > 0 && M > 0 && L > 0, ALVQ_EINVAL, "alvq_conv1d_wgrad_bf16_multi: bad dims");
  ALVQ_REQUIRE(KW == 1 || KW == 3, ALVQ_EUNSUPPORTED, "alvq_conv1d_wgrad_bf16_multi: KW=%d (only 1 and 3)", KW);
  ALVQ_REQUIRE(w_layout == ALVQ_W_OIK || w_layout == ALVQ_W_IOK, ALVQ_EINVAL, "alvq_conv1d_wgrad_bf16_multi: w_layout");
  return conv1d_wgrad_bf16_v2_launch(dy, x, nseg, dw, workspace, (int)alvq_nlc_rows(B, L), C, M, KW, w_layout, accumulate,
                                     (hipStream_t)stream);
}
