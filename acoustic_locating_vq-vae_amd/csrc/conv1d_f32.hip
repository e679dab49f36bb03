// fp32 1-D convolution family for gfx950: forward / data-grad (one kernel, two weight layouts) and
// weight-grad, as im2col-free implicit GEMMs on the exact-fp32 matrix cores (v_mfma_f32_16x16x4_f32).
//
// Data layout in HBM: activations (B, C, L) contiguous, L fastest.  The GEMM "N" axis is the flattened
// VIRTUAL position v = b*(L+PAD) + l: one shared zero column is inserted between consecutive batch rows so
// a k=3 halo never reads the neighbouring sample and tiles may straddle samples (B*L = 32000 = 250 full
// 128-wide tiles at the speech config; L=201 of the RIR config would waste 36 % with per-sample tiles).
// The input tile (BK channels x BN+2 positions) is staged ONCE in LDS per K-chunk and reused by all three
// taps -- that is the "im2col-free" part; HBM reads are coalesced along L.
#include "alvq_common.h"
#include "wgrad_reduce.h"

namespace alvq {

constexpr int BM = 128;  // output channels per workgroup
constexpr int BN = 128;  // virtual positions per workgroup

__host__ __device__ constexpr int pad32(int x, int r) {  // smallest y >= x with y % 32 == r
  return x + ((r - x % 32) + 32) % 32;
}

struct ConvArgs {
  const float* x;
  const float* w;
  const float* bias;
  const float* skip1;
  const float* skip2;
  const float* mask;
  const float* post;
  float* y;
  float* y2;
  int B, C, M, L;
  int relu;
  int mtiles, ntiles;
};

// KW: taps (1|3).  WT: weights are [C][M][KW] read flipped (ALVQ_W_IOK).  BK: input channels per chunk.
template <int KW, bool WT, int BK>
__global__ __launch_bounds__(256, 2) void conv1d_f32_kernel(ConvArgs a) {
  constexpr int PAD = (KW - 1) / 2;
  constexpr int XCOLS = BN + KW - 1;
  constexpr int XS = pad32(XCOLS, 16);                            // B-operand rows: banks j and 16+j
  constexpr int WROW = WT ? BM * KW : BK * KW;                    // contiguous floats per staged weight row
  constexpr int WROWS = WT ? BK : BM;
  constexpr int WS = WT ? pad32(WROW, 16) : pad32(WROW, 2);       // conflict-free A-operand reads
  constexpr int WELEMS = BM * BK * KW / 256;                      // per thread per chunk
  constexpr int WPER = (256 % WROW == 0) ? 1 : 3;                 // staging pattern repeats every WPER passes
  static_assert((256 * WPER) % WROW == 0, "weight staging pattern");
  constexpr int WQ = WELEMS / WPER;                               // row steps per period slot
  constexpr int WRSTEP = 256 * WPER / WROW;                       // rows advanced per step
  static_assert(WELEMS % WPER == 0 && (256 * WPER) % WROW == 0, "weight staging pattern");

  __shared__ float Xs[BK * XS];
  __shared__ float Ws[WROWS * WS];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;

  const int tile = xcd_remap(blockIdx.x, a.mtiles * a.ntiles);
  const int m0 = (tile / a.ntiles) * BM;
  const int n0 = (tile % a.ntiles) * BN;

  const int C = a.C, M = a.M, L = a.L;
  const int Lp = L + PAD;
  const int V = a.B * Lp;

  // ---- per-thread staging coordinates (fixed for the whole K loop) ----
  // X main columns: col = tid & 127, rows (tid>>7) + 2*i
  const int xcol = tid & 127, xrow0 = tid >> 7;
  long xoff;
  bool xok;
  {
    const int v = n0 - PAD + xcol;
    const int b = v / Lp, l = v - b * Lp;
    xok = (v >= 0) && (v < V) && (l < L);
    xoff = (long)b * C * L + l;
  }
  // X halo columns: (KW-1)*BK elements
  long hoff = 0;
  bool hok = false;
  int hrow = 0, hcol = 0;
  if (KW > 1 && tid < (KW - 1) * BK) {
    hcol = BN + tid % (KW - 1);
    hrow = tid / (KW - 1);
    const int v = n0 - PAD + hcol;
    const int b = v / Lp, l = v - b * Lp;
    hok = (v >= 0) && (v < V) && (l < L);
    hoff = (long)b * C * L + l;
  }
  // W: element (p, q) -> row = wrow[p] + q*WRSTEP, col = wcol[p]
  int wrow[WPER], wcol[WPER];
#pragma unroll
  for (int p = 0; p < WPER; ++p) {
    const int e = tid + 256 * p;
    wrow[p] = e / WROW;
    wcol[p] = e % WROW;
  }

  float xr[BK / 2], xh = 0.f, wr[WELEMS];

  auto load_chunk = [&](int c0) {
#pragma unroll
    for (int i = 0; i < BK / 2; ++i) {
      const int c = c0 + xrow0 + 2 * i;
      xr[i] = (xok && c < C) ? a.x[xoff + (long)c * L] : 0.f;
    }
    if (KW > 1) {
      const int c = c0 + hrow;
      xh = (hok && c < C) ? a.x[hoff + (long)c * L] : 0.f;
    }
#pragma unroll
    for (int p = 0; p < WPER; ++p) {
#pragma unroll
      for (int q = 0; q < WQ; ++q) {
        const int row = wrow[p] + q * WRSTEP, col = wcol[p];
        float v = 0.f;
        if (!WT) {
          const int m = m0 + row, cc = c0 * KW + col;
          if (m < M && cc < C * KW) v = a.w[(long)m * C * KW + cc];
        } else {
          const int c = c0 + row, mm = m0 * KW + col;
          if (c < C && mm < M * KW) v = a.w[(long)c * M * KW + mm];
        }
        wr[p * WQ + q] = v;
      }
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < BK / 2; ++i) Xs[(xrow0 + 2 * i) * XS + xcol] = xr[i];
    if (KW > 1 && tid < (KW - 1) * BK) Xs[hrow * XS + hcol] = xh;
#pragma unroll
    for (int p = 0; p < WPER; ++p)
#pragma unroll
      for (int q = 0; q < WQ; ++q) Ws[(wrow[p] + q * WRSTEP) * WS + wcol[p]] = wr[p * WQ + q];
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nchunks = (C + BK - 1) / BK;
  load_chunk(0);
  store_chunk();
  __syncthreads();
  for (int ch = 0; ch < nchunks; ++ch) {
    if (ch + 1 < nchunks) load_chunk((ch + 1) * BK);  // in flight under the MFMAs below
#pragma unroll
    for (int s = 0; s < BK / 4; ++s) {
#pragma unroll
      for (int t = 0; t < KW; ++t) {
        float af[4], bf[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          const int ml = wm0 + mi * 16 + li, cl = s * 4 + kq;
          af[mi] = WT ? Ws[cl * WS + ml * KW + (KW - 1 - t)] : Ws[ml * WS + cl * KW + t];
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) bf[ni] = Xs[(s * 4 + kq) * XS + wn0 + ni * 16 + li + t];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
      }
    }
    __syncthreads();
    if (ch + 1 < nchunks) {
      store_chunk();
      __syncthreads();
    }
  }

  // ---- epilogue: D[row = kq*4 + r][col = li] ----
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int v = n0 + wn0 + ni * 16 + li;
    const int b = v / Lp, l = v - b * Lp;
    if (v >= V || l >= L) continue;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm0 + mi * 16 + kq * 4 + r;
        if (m >= M) continue;
        const long o = ((long)b * M + m) * L + l;
        float val = acc[mi][ni][r];
        if (a.bias) val += a.bias[m];
        if (a.skip1) val += a.skip1[o];
        if (a.skip2) val += a.skip2[o];
        if (a.relu) val = fmaxf(val, 0.f);
        if (a.mask) val = a.mask[o] > 0.f ? val : 0.f;
        a.y[o] = val;
        if (a.y2) a.y2[o] = val + a.post[o];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------- wgrad
constexpr int GM = 128;  // dy channels (m) per workgroup
constexpr int GC = 64;   // x channels (c) per workgroup
constexpr int GV = 32;   // virtual positions per chunk

struct WgradArgs {
  const float* dy;
  const float* x;
  float* partial;  // [splits][KW][M][C]
  int B, C, M, L;
  int mtiles, ctiles, splits, chunks_per_split;
};

template <int KW>
__global__ __launch_bounds__(256, 2) void conv1d_wgrad_f32_kernel(WgradArgs a) {
  constexpr int PAD = (KW - 1) / 2;
  constexpr int XCOLS = GV + KW - 1;
  constexpr int YS = pad32(GV, 2);     // 34
  constexpr int XS = pad32(XCOLS, 2);  // 34
  __shared__ float Ys[GM * YS];
  __shared__ float Xs[GC * XS];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int wm0 = (wave >> 1) * 64, wc0 = (wave & 1) * 32;

  const int ntile = a.mtiles * a.ctiles;
  const int id = xcd_remap(blockIdx.x, ntile * a.splits);
  const int split = id / ntile;  // blocks that share an XCD share a position range (same dy/x slabs in L2)
  const int t_id = id % ntile;
  const int m0 = (t_id / a.ctiles) * GM;
  const int c0 = (t_id % a.ctiles) * GC;

  const int C = a.C, M = a.M, L = a.L;
  const int Lp = L + PAD;
  const int V = a.B * Lp;
  const int vbeg = split * a.chunks_per_split * GV;
  const int vend = min(V, vbeg + a.chunks_per_split * GV);

  f32x4 acc[KW][4][2];
#pragma unroll
  for (int t = 0; t < KW; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int col = tid & 31, row0 = tid >> 5;  // main columns: rows row0 + 8*i
  float yr[GM / 8], xr[GC / 8], xh = 0.f;
  const int hrow = tid / (KW > 1 ? KW - 1 : 1), hcol = GV + tid % (KW > 1 ? KW - 1 : 1);
  const bool hthread = KW > 1 && tid < (KW - 1) * GC;

  auto load_chunk = [&](int v0) {
    {  // dy: column v0+col, zero outside this split's range
      const int v = v0 + col;
      const int b = v / Lp, l = v - b * Lp;
      const bool ok = (v < vend) && (l < L);
      const long off = ((long)b * M) * L + l;
#pragma unroll
      for (int i = 0; i < GM / 8; ++i) {
        const int m = m0 + row0 + 8 * i;
        yr[i] = (ok && m < M) ? a.dy[off + (long)m * L] : 0.f;
      }
    }
    {  // x: column v0-PAD+col (valid wherever it lies inside its own sample)
      const int v = v0 - PAD + col;
      const int b = v / Lp, l = v - b * Lp;
      const bool ok = (v >= 0) && (v < V) && (l < L);
      const long off = ((long)b * C) * L + l;
#pragma unroll
      for (int i = 0; i < GC / 8; ++i) {
        const int c = c0 + row0 + 8 * i;
        xr[i] = (ok && c < C) ? a.x[off + (long)c * L] : 0.f;
      }
    }
    if (hthread) {
      const int v = v0 - PAD + hcol;
      const int b = v / Lp, l = v - b * Lp;
      const bool ok = (v >= 0) && (v < V) && (l < L);
      const int c = c0 + hrow;
      xh = (ok && c < C) ? a.x[((long)b * C + c) * L + l] : 0.f;
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < GM / 8; ++i) Ys[(row0 + 8 * i) * YS + col] = yr[i];
#pragma unroll
    for (int i = 0; i < GC / 8; ++i) Xs[(row0 + 8 * i) * XS + col] = xr[i];
    if (hthread) Xs[hrow * XS + hcol] = xh;
  };

  if (vbeg < vend) {
    load_chunk(vbeg);
    store_chunk();
    __syncthreads();
    for (int v0 = vbeg; v0 < vend; v0 += GV) {
      const bool more = v0 + GV < vend;
      if (more) load_chunk(v0 + GV);
#pragma unroll
      for (int s = 0; s < GV / 4; ++s) {
        float af[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) af[mi] = Ys[(wm0 + mi * 16 + li) * YS + s * 4 + kq];
#pragma unroll
        for (int t = 0; t < KW; ++t) {
          float bf[2];
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) bf[ni] = Xs[(wc0 + ni * 16 + li) * XS + s * 4 + kq + t];
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
              acc[t][mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mi], bf[ni], acc[t][mi][ni], 0, 0, 0);
        }
      }
      __syncthreads();
      if (more) {
        store_chunk();
        __syncthreads();
      }
    }
  }

  float* out = a.partial + (long)split * KW * M * C;
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
          if (m < M && c < C) out[((long)t * M + m) * C + c] = acc[t][mi][ni][r];
        }
}

// dbias[m] (+)= sum_b sum_l dy[b,m,l]; one workgroup per channel, fixed order.
__global__ __launch_bounds__(256) void bias_grad_kernel(const float* dy, float* dbias, int B, int M, int L, int accumulate) {
  const int m = blockIdx.x;
  float s = 0.f;
  const int total = B * L;
  for (int e = threadIdx.x; e < total; e += 256) {
    const int b = e / L, l = e - b * L;
    s += dy[((long)b * M + m) * L + l];
  }
  s = wave_sum(s);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float tot = (red[0] + red[1]) + (red[2] + red[3]);
    dbias[m] = accumulate ? dbias[m] + tot : tot;
  }
}

static int wgrad_splits(int B, int C, int M, int L, int KW, int* chunks_per_split) {
  const int PAD = (KW - 1) / 2;
  const long V = (long)B * (L + PAD);
  const int nchunks = (int)((V + GV - 1) / GV);
  const int tiles = ((M + GM - 1) / GM) * ((C + GC - 1) / GC);
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

template <int KW, bool WT>
static void launch_conv(const ConvArgs& a, hipStream_t s) {
  constexpr int BK = 16;
  hipLaunchKernelGGL((conv1d_f32_kernel<KW, WT, BK>), dim3(a.mtiles * a.ntiles), dim3(256), 0, s, a);
}

extern "C" int alvq_conv1d_f32(const float* x, const float* w, const float* bias, const float* skip1,
                               const float* skip2, const float* mask, const float* post, float* y, float* y2, int B,
                               int C, int M, int L, int KW, int w_layout, int relu, void* stream) {
  ALVQ_REQUIRE(x && w && y, ALVQ_EINVAL, "alvq_conv1d_f32: null x/w/y");
  ALVQ_REQUIRE(B > 0 && C > 0 && M > 0 && L > 0, ALVQ_EINVAL, "alvq_conv1d_f32: bad dims B=%d C=%d M=%d L=%d", B, C, M, L);
  ALVQ_REQUIRE(KW == 1 || KW == 3, ALVQ_EUNSUPPORTED, "alvq_conv1d_f32: KW=%d (only 1 and 3)", KW);
  ALVQ_REQUIRE(w_layout == ALVQ_W_OIK || w_layout == ALVQ_W_IOK, ALVQ_EINVAL, "alvq_conv1d_f32: w_layout=%d", w_layout);
  ALVQ_REQUIRE((y2 == nullptr) == (post == nullptr), ALVQ_EINVAL, "alvq_conv1d_f32: y2 and post go together");
  const long V = (long)B * (L + (KW - 1) / 2);
  ALVQ_REQUIRE(V < (1L << 31) - BN && (long)B * (C > M ? C : M) * L < (1L << 40), ALVQ_EUNSUPPORTED,
               "alvq_conv1d_f32: problem too large");
  ConvArgs a{x, w, bias, skip1, skip2, mask, post, y, y2, B, C, M, L, relu, (M + BM - 1) / BM, (int)((V + BN - 1) / BN)};
  hipStream_t s = (hipStream_t)stream;
  if (KW == 3) {
    if (w_layout == ALVQ_W_OIK) launch_conv<3, false>(a, s); else launch_conv<3, true>(a, s);
  } else {
    if (w_layout == ALVQ_W_OIK) launch_conv<1, false>(a, s); else launch_conv<1, true>(a, s);
  }
  return check_launch("alvq_conv1d_f32");
}

extern "C" int64_t alvq_conv1d_wgrad_workspace_bytes(int B, int C, int M, int L, int KW) {
  if (B <= 0 || C <= 0 || M <= 0 || L <= 0 || (KW != 1 && KW != 3)) return -1;
  int cps;
  const int splits = wgrad_splits(B, C, M, L, KW, &cps);
  return (int64_t)splits * KW * M * C * (int64_t)sizeof(float);
}

extern "C" int alvq_conv1d_wgrad_f32(const float* dy, const float* x, float* dw, float* dbias, void* workspace, int B,
                                     int C, int M, int L, int KW, int w_layout, int accumulate, void* stream) {
  ALVQ_REQUIRE(dy && x && dw && workspace, ALVQ_EINVAL, "alvq_conv1d_wgrad_f32: null pointer");
  ALVQ_REQUIRE(B > 0 && C > 0 && M > 0 && L > 0, ALVQ_EINVAL, "alvq_conv1d_wgrad_f32: bad dims");
  ALVQ_REQUIRE(KW == 1 || KW == 3, ALVQ_EUNSUPPORTED, "alvq_conv1d_wgrad_f32: KW=%d (only 1 and 3)", KW);
  ALVQ_REQUIRE(w_layout == ALVQ_W_OIK || w_layout == ALVQ_W_IOK, ALVQ_EINVAL, "alvq_conv1d_wgrad_f32: w_layout");
  ALVQ_REQUIRE((long)B * (L + 1) < (1L << 31) - 64, ALVQ_EUNSUPPORTED, "alvq_conv1d_wgrad_f32: problem too large");
  hipStream_t s = (hipStream_t)stream;
  int cps;
  const int splits = wgrad_splits(B, C, M, L, KW, &cps);
  WgradArgs a{dy, x, (float*)workspace, B, C, M, L, (M + GM - 1) / GM, (C + GC - 1) / GC, splits, cps};
  const int grid = a.mtiles * a.ctiles * splits;
  if (KW == 3)
    hipLaunchKernelGGL((conv1d_wgrad_f32_kernel<3>), dim3(grid), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((conv1d_wgrad_f32_kernel<1>), dim3(grid), dim3(256), 0, s, a);
  int rc = check_launch("alvq_conv1d_wgrad_f32");
  if (rc) return rc;
  wgrad_reduce_launch((const float*)workspace, dw, splits, KW, M, C, w_layout, accumulate, s);
  rc = check_launch("alvq_conv1d_wgrad_f32/reduce");
  if (rc) return rc;
  if (dbias) {
    hipLaunchKernelGGL(bias_grad_kernel, dim3(M), dim3(256), 0, s, dy, dbias, B, M, L, accumulate);
    rc = check_launch("alvq_conv1d_wgrad_f32/bias");
  }
  return rc;
}
