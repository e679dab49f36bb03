// bf16-MFMA 1-D convolution family for gfx950 (throughput path): forward / data-grad / ConvTranspose and
// weight-grad on v_mfma_f32_16x16x32_bf16 with fp32 accumulation.
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
#include "wgrad_reduce.h"

namespace alvq {

template <int KW, int OUT, int DBG = 0>
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
  const int m0 = (a.relu & 2 ? tile / a.rtiles : tile % a.mtiles) * TB_M;
  const int r0 = (a.relu & 2 ? tile % a.rtiles : tile / a.mtiles) * TB_R;
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
    if (step + 1 < nsteps && DBG != 1) {
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
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mi], bfr[ni], acc[mi][ni], 0, 0, 0);
    }
    if (DBG != 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (DBG != 2) __syncthreads(); else __builtin_amdgcn_s_barrier();
    tap = ntap;
    chunk = nchunk;
  }

  // ---- epilogue 1: D[i = m][j = row] -> fp32 C tile Cs[row][m] (4 consecutive m per lane = one 16-B write)
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
  if (OUT == 0) {
    // ---- epilogue 2 (NLC bf16): thread = 8 consecutive channels of one row; 16 rows per pass
    const int tx = tid & 15, ty = tid >> 4;
    const int mbase = m0 + tx * 8;
    if (mbase < a.Mop) {
      float bv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) bv[e] = (a.bias && mbase + e < a.M) ? a.bias[mbase + e] : 0.f;
#pragma unroll
      for (int pass = 0; pass < 8; ++pass) {
        const int rl = pass * 16 + ty, row = r0 + rl;
        int b, l;
        const bool ok = row_valid(row, Lp1, ndata, &b, &l);
        const long o = (long)row * a.Mop + mbase;
        float v[8];
        const f32x4 c0 = *(const f32x4*)(Cs + rl * CS + tx * 8), c1 = *(const f32x4*)(Cs + rl * CS + tx * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = c0[e];
          v[e + 4] = c1[e];
        }
        epilogue_store8(a, v, bv, ok, o);
      }
    }
  } else {
    // ---- epilogue 2 (NCL fp32, bias only): lane = row (coalesced along l), loop over channels
    const int rl = tid & 127, row = r0 + rl;
    int b, l;
    const bool ok = row_valid(row, Lp1, ndata, &b, &l);
    if (ok) {
      for (int ml = tid >> 7; ml < TB_M; ml += 2) {
        const int m = m0 + ml;
        if (m >= a.M) break;
        a.y_ncl[((long)b * a.M + m) * a.L + l] = Cs[rl * CS + ml] + (a.bias ? a.bias[m] : 0.f);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------- weight-grad
// dW_t[m][c] = sum_rows dY[row][m] * X[row + t - PAD][c]; the contraction axis (rows) is the SLOW axis of both
// NLC operands, so fragments are fetched with the transposing LDS read ds_read_b64_tr_b16.
constexpr int WG_M = 128, WG_C = 64, WG_R = 64;   // tile: 128 dy-channels x 64 x-channels, 64 rows per chunk

struct WgradBArgs {
  const u16* dy;   // [rows][Mp]
  const u16* x;    // [rows][Cp]
  float* partial;  // [splits][KW][M][C]
  int Mp, Cp, M, C;
  int mtiles, ctiles, splits, chunks_per_split, total_rows;
};

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// 8 consecutive k (rows) of one column, gathered by two transposing LDS reads (4 rows each).
__device__ __forceinline__ bf16x8_t tr_frag(const unsigned char* p, int row_bytes) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + 4 * row_bytes));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

template <int KW>
__global__ __launch_bounds__(256, 2) void conv1d_wgrad_bf16_kernel(WgradBArgs a) {
  constexpr int PAD = (KW - 1) / 2;
  constexpr int YROW = WG_M * 2 + 16;            // bytes per LDS row of the dy tile (padded, 16-B aligned)
  constexpr int XROW = WG_C * 2 + 16;            // bytes per LDS row of the x tile
  constexpr int XR = WG_R + 8;                   // rows staged for x (halo)
  __shared__ __attribute__((aligned(16))) unsigned char Ys[WG_R * YROW];
  __shared__ __attribute__((aligned(16))) unsigned char Xs[XR * XROW];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave >> 1) * 64, wc0 = (wave & 1) * 32;
  const int ntile = a.mtiles * a.ctiles;
  const int id = xcd_remap(blockIdx.x, ntile * a.splits);
  const int split = id / ntile, t_id = id % ntile;
  const int m0 = (t_id / a.ctiles) * WG_M, c0 = (t_id % a.ctiles) * WG_C;
  const int rbeg = split * a.chunks_per_split * WG_R;
  const int rend = min(a.total_rows, rbeg + a.chunks_per_split * WG_R);

  f32x4 acc[KW][4][2];
#pragma unroll
  for (int t = 0; t < KW; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // staging through registers (16-B loads): dy tile 64 rows x 256 B = 1024 x 16 B -> 4 per thread;
  // x tile 72 rows x 128 B = 576 x 16 B -> 3 per thread (last partial)
  u16x8 yr[4], xr[3];
  auto load_chunk = [&](int rr) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + 256 * i, row = e >> 4, ch = e & 15;
      const int r = rr + row;
      const int mcol = m0 + ch * 8;
      yr[i] = (r < rend && mcol < a.Mp) ? *(const u16x8*)(a.dy + (long)r * a.Mp + mcol) : u16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int e = tid + 256 * i, row = e >> 3, ch = e & 7;
      const int ccol = c0 + ch * 8;
      const int r = rr - PAD + row;   // rows outside the matrix are guard rows: treat as zero (0 * garbage = NaN)
      xr[i] = (row < XR && ccol < a.Cp && r >= 0 && r < a.total_rows) ? *(const u16x8*)(a.x + (long)r * a.Cp + ccol)
                                                                     : u16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + 256 * i, row = e >> 4, ch = e & 15;
      *(u16x8*)(Ys + row * YROW + ch * 16) = yr[i];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int e = tid + 256 * i, row = e >> 3, ch = e & 7;
      if (row < XR) *(u16x8*)(Xs + row * XROW + ch * 16) = xr[i];
    }
  };

  // transposed fragment read: 16-lane group g = lane>>4 handles k rows 8g..8g+7 in two 4-row blocks;
  // lane 4q+p of the group supplies the address of block row q, columns 4p..4p+3
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;

  if (rbeg < rend) {
    load_chunk(rbeg);
    store_chunk();
    __syncthreads();
    for (int rr = rbeg; rr < rend; rr += WG_R) {
      const bool more = rr + WG_R < rend;
      if (more) load_chunk(rr + WG_R);
#pragma unroll
      for (int s = 0; s < WG_R / 32; ++s) {
        bf16x8_t af[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          af[mi] = tr_frag(Ys + (s * 32 + 8 * g + q) * YROW + (wm0 + mi * 16 + 4 * p) * 2, YROW);
        }
#pragma unroll
        for (int t = 0; t < KW; ++t) {
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            const bf16x8_t bfr = tr_frag(Xs + (s * 32 + 8 * g + q + t) * XROW + (wc0 + ni * 16 + 4 * p) * 2, XROW);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
              acc[t][mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mi], bfr, acc[t][mi][ni], 0, 0, 0);
          }
        }
      }
      __syncthreads();
      if (more) {
        store_chunk();
        __syncthreads();
      }
    }
  }

  const int li = lane & 15, kq = lane >> 4;
  float* out = a.partial + (long)split * KW * a.M * a.C;
#pragma unroll
  for (int t = 0; t < KW; ++t)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = m0 + wm0 + mi * 16 + kq * 4 + r;
          const int c = c0 + wc0 + ni * 16 + li;
          if (m < a.M && c < a.C) out[((long)t * a.M + m) * a.C + c] = acc[t][mi][ni][r];
        }
}

// ------------------------------------------------------------------------------------------- helpers
// Wp[t][m][c] = bf16(A_t[m][c]);  OIK: A_t[m][c] = w[m][c][t];  IOK: A_t[m][c] = w[c][m][KW-1-t]; zero padded.
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* w, u16* wp, int M, int C, int KW, int Mp128, int Cp,
                                                          int w_layout) {
  const long total = (long)KW * Mp128 * Cp;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c = (int)(e % Cp);
    const int m = (int)((e / Cp) % Mp128);
    const int t = (int)(e / ((long)Cp * Mp128));
    float v = 0.f;
    if (m < M && c < C) v = w_layout == ALVQ_W_OIK ? w[((long)m * C + c) * KW + t] : w[((long)c * M + m) * KW + (KW - 1 - t)];
    wp[e] = f2bf(v);
  }
}

// (B,C,L) fp32 -> NLC-padded bf16 [rows_total][Cp] (gap rows, tail rows and padded channels zero).
__global__ __launch_bounds__(256) void ncl_to_nlc_kernel(const float* x, u16* y, int B, int C, int L, int Cp, int rows_total) {
  __shared__ float tile[32][33];
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
    tile[ty + 8 * i][tx] = ok ? x[((long)b * C + c) * L + l] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {  // write: lanes along channels
    const int row = r0 + ty + 8 * i, c = c0 + tx;
    if (row < rows_total) y[(long)row * Cp + c] = f2bf(tile[tx][ty + 8 * i]);
  }
}

// NLC-padded bf16 -> (B,C,L) fp32 (for module outputs that leave the bf16 pipeline).
__global__ __launch_bounds__(256) void nlc_to_ncl_kernel(const u16* x, float* y, int B, int C, int L, int Cp, int rows_total) {
  __shared__ float tile[32][33];
  const int ct = Cp / 32;
  const int r0 = (blockIdx.x / ct) * 32, c0 = (blockIdx.x % ct) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int Lp1 = L + 1, ndata = B * Lp1;
#pragma unroll
  for (int i = 0; i < 4; ++i) {  // read: lanes along channels
    const int row = r0 + ty + 8 * i, c = c0 + tx;
    tile[ty + 8 * i][tx] = row < rows_total ? bf2f(x[(long)row * Cp + c]) : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {  // write: lanes along rows (l contiguous in the destination)
    const int c = c0 + ty + 8 * i, row = r0 + tx;
    int b, l;
    if (row_valid(row, Lp1, ndata, &b, &l) && c < C) y[((long)b * C + c) * L + l] = tile[tx][ty + 8 * i];
  }
}

// column sums of an NLC matrix: partial[s][m] = sum over the split's rows of dy[row][m].
// Thread = 8 consecutive channels (one 16-B load per row, rows fully coalesced); grid = (channel groups, splits).
__global__ __launch_bounds__(256) void bias_grad_nlc_partial_kernel(const u16* dy, float* partial, int rows, int Mp,
                                                                    int rows_per_split) {
  const int groups = Mp / 8;                         // 16-B channel groups per row
  const int gpb = groups < 256 ? groups : 256;       // groups handled by this block (threads along channels)
  const int rsub = 256 / gpb;                        // row phases inside the block
  const int gi = threadIdx.x % gpb, rp = threadIdx.x / gpb;
  const int grp = blockIdx.x * gpb + gi;
  const int rb = blockIdx.y * rows_per_split, re = min(rows, rb + rows_per_split);
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (grp < groups && rp < rsub)
    for (int r = rb + rp; r < re; r += rsub) {
      const u16x8 v = *(const u16x8*)(dy + (long)r * Mp + grp * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += bf2f(v[e]);
    }
  __shared__ float red[256][9];
#pragma unroll
  for (int e = 0; e < 8; ++e) red[threadIdx.x][e] = s[e];
  __syncthreads();
  if (rp == 0 && grp < groups) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = 0.f;
      for (int k = 0; k < rsub; ++k) t += red[k * gpb + gi][e];
      partial[(long)blockIdx.y * Mp + grp * 8 + e] = t;
    }
  }
}

// dbias[m] (+)= sum_s partial[s][m]; 64 channels x 4 split-phases per workgroup, fixed order.
__global__ __launch_bounds__(256) void bias_grad_nlc_final_kernel(const float* partial, float* dbias, int splits, int Mp, int M,
                                                                  int accumulate) {
  // 32 channels x 8 split-phases per workgroup (coalesced along m, 8-way parallel along the splits), fixed order
  const int mi = threadIdx.x & 31, ph = threadIdx.x >> 5;
  const int m = blockIdx.x * 32 + mi;
  float s = 0.f;
  if (m < M)
    for (int k = ph; k < splits; k += 8) s += partial[(long)k * Mp + m];
  __shared__ float red[8][32];
  red[ph][mi] = s;
  __syncthreads();
  if (ph == 0 && m < M) {
    const float t = ((red[0][mi] + red[1][mi]) + (red[2][mi] + red[3][mi])) + ((red[4][mi] + red[5][mi]) + (red[6][mi] + red[7][mi]));
    dbias[m] = accumulate ? dbias[m] + t : t;
  }
}

// out = mask > 0 ? dy : 0 on NLC bf16 buffers (whole padded matrix)
__global__ __launch_bounds__(256) void relu_mask_bf16_kernel(const u16* dy, const u16* t, u16* out, long n8) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n8; e += (long)gridDim.x * 256) {
    const u16x8 d = ((const u16x8*)dy)[e], m = ((const u16x8*)t)[e];
    u16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = bf2f(m[i]) > 0.f ? d[i] : (u16)0;
    ((u16x8*)out)[e] = o;
  }
}

constexpr int BIAS_SPLITS = 256;   // row ranges of the bias-grad column sums (one workgroup each)

static int wgrad_b_splits(int total_rows, int M, int C, int* chunks_per_split) {
  const int nchunks = (total_rows + WG_R - 1) / WG_R;
  const int tiles = ((M + WG_M - 1) / WG_M) * ((C + WG_C - 1) / WG_C);
  int want = (512 + tiles - 1) / tiles;  // one full wave of workgroups (2 per CU); keeps the partial slab small
  if (want < 1) want = 1;
  if (want > nchunks) want = nchunks;
  if (want > 64) want = 64;
  const int cps = (nchunks + want - 1) / want;
  *chunks_per_split = cps;
  return (nchunks + cps - 1) / cps;
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
  ALVQ_REQUIRE(w && wp, ALVQ_EINVAL, "alvq_pack_weight_bf16: null pointer");
  ALVQ_REQUIRE(M > 0 && C > 0 && (KW == 1 || KW == 3), ALVQ_EINVAL, "alvq_pack_weight_bf16: bad dims");
  ALVQ_REQUIRE(w_layout == ALVQ_W_OIK || w_layout == ALVQ_W_IOK, ALVQ_EINVAL, "alvq_pack_weight_bf16: w_layout");
  const long total = alvq_packed_weight_elems(M, C, KW);
  int grid = (int)((total + 1023) / 1024);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, (u16*)wp, M, C, KW,
                     pad_to(M, WP_ROWS), pad_to(C, TB_K), w_layout);
  return check_launch("alvq_pack_weight_bf16");
}

extern "C" int alvq_ncl_to_nlc_bf16(const float* x, void* y, int B, int C, int L, void* stream) {
  ALVQ_REQUIRE(x && y, ALVQ_EINVAL, "alvq_ncl_to_nlc_bf16: null pointer");
  ALVQ_REQUIRE(B > 0 && C > 0 && L > 0, ALVQ_EINVAL, "alvq_ncl_to_nlc_bf16: bad dims");
  const int Cp = pad_to(C, TB_K), rows = (int)alvq_nlc_rows(B, L);
  hipLaunchKernelGGL(ncl_to_nlc_kernel, dim3((rows / 32) * (Cp / 32)), dim3(256), 0, (hipStream_t)stream, x, (u16*)y, B, C,
                     L, Cp, rows);
  return check_launch("alvq_ncl_to_nlc_bf16");
}

extern "C" int alvq_nlc_to_ncl_f32(const void* x, float* y, int B, int C, int L, void* stream) {
  ALVQ_REQUIRE(x && y, ALVQ_EINVAL, "alvq_nlc_to_ncl_f32: null pointer");
  ALVQ_REQUIRE(B > 0 && C > 0 && L > 0, ALVQ_EINVAL, "alvq_nlc_to_ncl_f32: bad dims");
  const int Cp = pad_to(C, TB_K), rows = (int)alvq_nlc_rows(B, L);
  hipLaunchKernelGGL(nlc_to_ncl_kernel, dim3((rows / 32) * (Cp / 32)), dim3(256), 0, (hipStream_t)stream, (const u16*)x, y, B,
                     C, L, Cp, rows);
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

extern "C" int alvq_conv1d_bf16(const void* x, const void* wp, const float* bias, const void* skip1, const void* skip2,
                                const void* mask, const void* post, void* y, void* y2, float* y_ncl, int B, int C, int M,
                                int L, int KW, int relu, void* stream) {
  ALVQ_REQUIRE(x && wp && (y || y_ncl), ALVQ_EINVAL, "alvq_conv1d_bf16: null x/wp/y");
  ALVQ_REQUIRE(!(y && y_ncl), ALVQ_EINVAL, "alvq_conv1d_bf16: choose one of y (NLC bf16) and y_ncl (NCL fp32)");
  ALVQ_REQUIRE(B > 0 && C > 0 && M > 0 && L > 0, ALVQ_EINVAL, "alvq_conv1d_bf16: bad dims");
  ALVQ_REQUIRE(KW == 1 || KW == 3, ALVQ_EUNSUPPORTED, "alvq_conv1d_bf16: KW=%d (only 1 and 3)", KW);
  ALVQ_REQUIRE((y2 == nullptr) == (post == nullptr), ALVQ_EINVAL, "alvq_conv1d_bf16: y2 and post go together");
  ALVQ_REQUIRE(!y_ncl || (!skip1 && !skip2 && !mask && !post && !relu), ALVQ_EUNSUPPORTED,
               "alvq_conv1d_bf16: the NCL fp32 epilogue fuses bias only");
  ALVQ_REQUIRE((long)B * (L + 1) < (1L << 30), ALVQ_EUNSUPPORTED, "alvq_conv1d_bf16: problem too large");
  ConvBArgs a{(const u16*)x, (const u16*)wp, bias, (const u16*)skip1, (const u16*)skip2, (const u16*)mask, (const u16*)post,
              (u16*)y, (u16*)y2, y_ncl, B, L, pad_to(C, TB_K), M, pad_to(M, TB_K), pad_to(M, WP_ROWS), relu,
              (int)(alvq_nlc_rows(B, L) / TB_R), pad_to(M, TB_M) / TB_M};
  hipStream_t s = (hipStream_t)stream;
  static int mmajor = -1;
  if (mmajor < 0) mmajor = getenv("ALVQ_MMAJOR") ? atoi(getenv("ALVQ_MMAJOR")) : 0;
  a.relu = (relu ? 1 : 0) | (mmajor ? 2 : 0);   // bit 1: experiment switch for the tile order
  static int use_v2 = -1;
  if (use_v2 < 0) use_v2 = getenv("ALVQ_CONV_V2") ? atoi(getenv("ALVQ_CONV_V2")) : 1;
  // wide layers: 256x256 tiles (v2) whenever the 256-wide m-tile is (nearly) full; narrow or ragged M
  // (128, 192, 201, 64, 1) stays on 128x128 tiles, which waste less there and give more workgroups.
  static int use_k3 = -1;
  if (use_k3 < 0) use_k3 = getenv("ALVQ_CONV_K3") ? atoi(getenv("ALVQ_CONV_K3")) : 1;
  if (use_v2 && pad_to(M, 256) - M <= 32) return (KW == 3 && use_k3) ? conv1d_bf16_k3_launch(a, s) : conv1d_bf16_v2_launch(a, KW, s);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv1d_bf16_kernel<3, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)conv1d_bf16_kernel<3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)conv1d_bf16_kernel<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)conv1d_bf16_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr = true;
  }
  const dim3 grid(a.rtiles * a.mtiles), block(256);
  static int dbg = -1;
  if (dbg < 0) dbg = getenv("ALVQ_DBG") ? atoi(getenv("ALVQ_DBG")) : 0;
  if (dbg == 1 && KW == 3 && y) {
    (void)hipFuncSetAttribute((const void*)conv1d_bf16_kernel<3, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipLaunchKernelGGL((conv1d_bf16_kernel<3, 0, 1>), grid, block, LDS_BYTES, s, a);
    return check_launch("alvq_conv1d_bf16");
  }
  if (dbg == 2 && KW == 3 && y) {
    (void)hipFuncSetAttribute((const void*)conv1d_bf16_kernel<3, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipLaunchKernelGGL((conv1d_bf16_kernel<3, 0, 2>), grid, block, LDS_BYTES, s, a);
    return check_launch("alvq_conv1d_bf16");
  }
  if (KW == 3) {
    if (y) hipLaunchKernelGGL((conv1d_bf16_kernel<3, 0>), grid, block, LDS_BYTES, s, a);
    else hipLaunchKernelGGL((conv1d_bf16_kernel<3, 1>), grid, block, LDS_BYTES, s, a);
  } else {
    if (y) hipLaunchKernelGGL((conv1d_bf16_kernel<1, 0>), grid, block, LDS_BYTES, s, a);
    else hipLaunchKernelGGL((conv1d_bf16_kernel<1, 1>), grid, block, LDS_BYTES, s, a);
  }
  return check_launch("alvq_conv1d_bf16");
}

extern "C" int64_t alvq_conv1d_wgrad_bf16_workspace_bytes(int B, int C, int M, int L, int KW) {
  if (B <= 0 || C <= 0 || M <= 0 || L <= 0 || (KW != 1 && KW != 3)) return -1;
  int cps;
  const int rows = (int)alvq_nlc_rows(B, L);
  const int splits = wgrad_b_splits(rows, M, C, &cps);
  int64_t w = (int64_t)splits * KW * M * C * 4;
  const int64_t w2 = conv1d_wgrad_bf16_v2_workspace_bytes(rows, C, M, KW);
  if (w2 > w) w = w2;
  const int64_t bsz = (int64_t)BIAS_SPLITS * pad_to(M, TB_K) * 4;
  return w + bsz;
}

extern "C" int alvq_conv1d_wgrad_bf16(const void* dy, const void* x, float* dw, float* dbias, void* workspace, int B, int C,
                                      int M, int L, int KW, int w_layout, int accumulate, void* stream) {
  ALVQ_REQUIRE(dy && x && dw && workspace, ALVQ_EINVAL, "alvq_conv1d_wgrad_bf16: null pointer");
  ALVQ_REQUIRE(B > 0 && C > 0 && M > 0 && L > 0, ALVQ_EINVAL, "alvq_conv1d_wgrad_bf16: bad dims");
  ALVQ_REQUIRE(KW == 1 || KW == 3, ALVQ_EUNSUPPORTED, "alvq_conv1d_wgrad_bf16: KW=%d (only 1 and 3)", KW);
  ALVQ_REQUIRE(w_layout == ALVQ_W_OIK || w_layout == ALVQ_W_IOK, ALVQ_EINVAL, "alvq_conv1d_wgrad_bf16: w_layout");
  hipStream_t s = (hipStream_t)stream;
  const int rows = (int)alvq_nlc_rows(B, L);
  int cps;
  const int splits = wgrad_b_splits(rows, M, C, &cps);
  static int use_v2 = -1;
  if (use_v2 < 0) use_v2 = getenv("ALVQ_WGRAD_V2") ? atoi(getenv("ALVQ_WGRAD_V2")) : 1;
  // the bias partials live behind the LARGER of the two kernels' weight slabs
  int64_t wbytes = (int64_t)splits * KW * M * C * 4;
  const int64_t w2 = conv1d_wgrad_bf16_v2_workspace_bytes(rows, C, M, KW);
  if (w2 > wbytes) wbytes = w2;
  if (use_v2) {
    // the bias gradient rides in the same launch (column sums of dY by an all-ones MFMA operand)
    return conv1d_wgrad_bf16_v2_launch(&dy, &x, 1, dw, workspace, rows, C, M, KW, w_layout, accumulate, s, dbias,
                                       (float*)((char*)workspace + wbytes));
  }
  WgradBArgs a{(const u16*)dy, (const u16*)x, (float*)workspace, pad_to(M, TB_K), pad_to(C, TB_K), M, C,
               (M + WG_M - 1) / WG_M, (C + WG_C - 1) / WG_C, splits, cps, rows};
  const int grid = a.mtiles * a.ctiles * splits;
  if (KW == 3) hipLaunchKernelGGL((conv1d_wgrad_bf16_kernel<3>), dim3(grid), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((conv1d_wgrad_bf16_kernel<1>), dim3(grid), dim3(256), 0, s, a);
  int rc = check_launch("alvq_conv1d_wgrad_bf16");
  if (rc) return rc;
  wgrad_reduce_launch((const float*)workspace, dw, splits, KW, M, C, w_layout, accumulate, s);
  rc = check_launch("alvq_conv1d_wgrad_bf16/reduce");
  if (rc) return rc;
  if (dbias) {
    float* bpart = (float*)((char*)workspace + wbytes);
    const int Mp = pad_to(M, TB_K), bs = BIAS_SPLITS, rps = (rows + bs - 1) / bs;
    hipLaunchKernelGGL(bias_grad_nlc_partial_kernel, dim3((Mp / 8 + 255) / 256, bs), dim3(256), 0, s, (const u16*)dy, bpart, rows, Mp, rps);
    hipLaunchKernelGGL(bias_grad_nlc_final_kernel, dim3((M + 31) / 32), dim3(256), 0, s, (const float*)bpart, dbias, bs, Mp, M,
                       accumulate);
    rc = check_launch("alvq_conv1d_wgrad_bf16/bias");
  }
  return rc;
}

extern "C" int alvq_conv1d_wgrad_bf16_multi(const void* const* dy, const void* const* x, int nseg, float* dw, void* workspace,
                                            int B, int C, int M, int L, int KW, int w_layout, int accumulate, void* stream) {
  ALVQ_REQUIRE(dy && x && dw && workspace, ALVQ_EINVAL, "alvq_conv1d_wgrad_bf16_multi: null pointer");
  ALVQ_REQUIRE(nseg >= 1 && nseg <= 4, ALVQ_EUNSUPPORTED, "alvq_conv1d_wgrad_bf16_multi: nseg=%d (1..4)", nseg);
  for (int i = 0; i < nseg; ++i) ALVQ_REQUIRE(dy[i] && x[i], ALVQ_EINVAL, "alvq_conv1d_wgrad_bf16_multi: null segment %d", i);
  ALVQ_REQUIRE(B > 0 && C > 0 && M > 0 && L > 0, ALVQ_EINVAL, "alvq_conv1d_wgrad_bf16_multi: bad dims");
  ALVQ_REQUIRE(KW == 1 || KW == 3, ALVQ_EUNSUPPORTED, "alvq_conv1d_wgrad_bf16_multi: KW=%d (only 1 and 3)", KW);
  ALVQ_REQUIRE(w_layout == ALVQ_W_OIK || w_layout == ALVQ_W_IOK, ALVQ_EINVAL, "alvq_conv1d_wgrad_bf16_multi: w_layout");
  return conv1d_wgrad_bf16_v2_launch(dy, x, nseg, dw, workspace, (int)alvq_nlc_rows(B, L), C, M, KW, w_layout, accumulate,
                                     (hipStream_t)stream);
}
