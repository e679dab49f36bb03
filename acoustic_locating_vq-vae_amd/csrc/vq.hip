// Vector-quantiser kernels (reference: vq_vae/vector_quantizer.py:29-58).
//
// Rows are D-float chunks of the contiguous (B,D,L) activation buffer in memory order (no permute, :32),
// so every row read/write below is a fully coalesced contiguous segment.
//
// argmin: pairwise-L2 as an exact-fp32 MFMA GEMM (x . E^T) with the x row-block stationary in LDS and the
// codebook streamed through LDS in (128 codes x 32 dims) tiles -- no default codebook fits 160 KiB
// (1024x128 fp32 = 512 KiB), so "LDS-resident" means tiled.  The distance is formed exactly as the
// reference does, d = fl(fl(|x|^2 + |e|^2) - 2 x.e), and reduced on (d, k) with lowest k on ties
// (torch.argmin semantics), first along each lane's own columns, then across the 16 lanes that share a row.
#include "alvq_common.h"

namespace alvq {

constexpr int VQ_RB_MAX = 128;  // rows per workgroup: 16 per wave, 4 or 8 waves
constexpr int VQ_CT = 128;   // codes per tile
constexpr int VQ_DK = 32;    // dims per staged chunk
constexpr int VQ_ES = 34;    // Es row stride: 34*li mod 32 = 2*li -> conflict-free with kq in {0,1}
constexpr int VQ_PARTIALS = 1024;

__host__ __device__ constexpr int vq_pad32(int x, int r) { return x + ((r - x % 32) + 32) % 32; }

// out[r] = sum_d x[r][d]^2, one wave per row, fixed order.
__global__ __launch_bounds__(256) void row_sqnorm_kernel(const float* x, float* out, long rows, int D) {
  const int lane = threadIdx.x & 63;
  for (long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (long)gridDim.x * 4) {
    float s = 0.f;
    for (int d = lane; d < D; d += 64) {
      const float v = x[r * D + d];
      s += v * v;
    }
    s = wave_sum(s);
    if (lane == 0) out[r] = s;
  }
}

struct ArgminArgs {
  const float* x;
  const float* e;
  const float* xn;  // |x_n|^2
  const float* en;  // |e_k|^2
  int64_t* idx;
  float* min_dist;
  long N;
  int K, D, Dp, XSTR;
};

// NW waves per workgroup, 16 rows each.  NW = 4 for D <= 128 (50 KB LDS, three workgroups per CU); NW = 8 for wider
// rows, where the stationary block would otherwise allow one 4-wave workgroup per CU (one wave per SIMD).
template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void vq_argmin_f32_kernel(ArgminArgs a) {
  constexpr int NT = 64 * NW, VQ_RB = 16 * NW, EPT = VQ_CT * VQ_DK / NT, ERS = NT / 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xs = smem;                          // [VQ_RB][XSTR]
  float* Es = smem + VQ_RB * a.XSTR;         // [VQ_CT][VQ_ES]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const long r0 = (long)blockIdx.x * VQ_RB;
  const int D = a.D, K = a.K, XSTR = a.XSTR;

  // stage the stationary x block (zero padded to Dp columns / missing rows)
  for (int e = tid; e < VQ_RB * a.Dp; e += NT) {
    const int r = e / a.Dp, d = e - r * a.Dp;
    const long row = r0 + r;
    Xs[r * XSTR + d] = (row < a.N && d < D) ? a.x[row * D + d] : 0.f;
  }

  float xn[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const long row = r0 + wave * 16 + kq * 4 + r;
    xn[r] = row < a.N ? a.xn[row] : 0.f;
  }
  float best[4];
  int bidx[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    best[r] = __builtin_inff();
    bidx[r] = 0x7fffffff;
  }

  // codebook staging: 128 codes x 32 dims = 4096 floats -> EPT per thread; thread -> (code = e/32, d = e%32)
  const int ecol = tid & 31, erow0 = tid >> 5;  // rows erow0 + ERS*i
  float er[EPT];
  auto load_e = [&](int k0, int d0) {
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      const int k = k0 + erow0 + ERS * i, d = d0 + ecol;
      er[i] = (k < K && d < D) ? a.e[(long)k * D + d] : 0.f;
    }
  };
  auto store_e = [&]() {
#pragma unroll
    for (int i = 0; i < EPT; ++i) Es[(erow0 + ERS * i) * VQ_ES + ecol] = er[i];
  };

  const int nd = a.Dp / VQ_DK;
  const int nkt = (K + VQ_CT - 1) / VQ_CT;
  const int total = nkt * nd;
  load_e(0, 0);
  store_e();
  __syncthreads();
  f32x4 acc[8];
  for (int it = 0; it < total; ++it) {
    const int kt = it / nd, dc = it - kt * nd;
    if (dc == 0) {
#pragma unroll
      for (int ni = 0; ni < 8; ++ni) acc[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (it + 1 < total) {
      const int kt2 = (it + 1) / nd, dc2 = (it + 1) - kt2 * nd;
      load_e(kt2 * VQ_CT, dc2 * VQ_DK);
    }
#pragma unroll
    for (int s = 0; s < VQ_DK / 4; ++s) {
      const float af = Xs[(wave * 16 + li) * XSTR + dc * VQ_DK + s * 4 + kq];
#pragma unroll
      for (int ni = 0; ni < 8; ++ni) {
        const float bf = Es[(ni * 16 + li) * VQ_ES + s * 4 + kq];
        acc[ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[ni], 0, 0, 0);
      }
    }
    if (dc == nd - 1) {
      // distances for this code tile: D[row = kq*4 + r][col = li]
#pragma unroll
      for (int ni = 0; ni < 8; ++ni) {
        const int k = kt * VQ_CT + ni * 16 + li;
        if (k < K) {
          const float bn = a.en[k];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float d = (xn[r] + bn) - 2.0f * acc[ni][r];
            if (d < best[r]) {
              best[r] = d;
              bidx[r] = k;
            }
          }
        }
      }
    }
    __syncthreads();
    if (it + 1 < total) {
      store_e();
      __syncthreads();
    }
  }

  // reduce over the 16 lanes (li) that hold different columns of the same rows
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float d2 = __shfl_xor(best[r], o, 64);
      const int k2 = __shfl_xor(bidx[r], o, 64);
      if (d2 < best[r] || (d2 == best[r] && k2 < bidx[r])) {
        best[r] = d2;
        bidx[r] = k2;
      }
    }
  }
  if (li == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long row = r0 + wave * 16 + kq * 4 + r;
      if (row < a.N) {
        a.idx[row] = (int64_t)(bidx[r] == 0x7fffffff ? 0 : bidx[r]);
        if (a.min_dist) a.min_dist[row] = best[r];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Round 4: the same search with the x rows of a wave held in REGISTERS (D <= 256).  The stationary block in LDS was what
// limited the stress config (BASELINE configs[3]: 256 000 x 256 against 4 096 codes): 128 rows x 256 dims = 128 KB left room
// for ONE workgroup per CU and a single-buffered codebook tile, so every one of its two barriers per 64 MFMAs was exposed
// (0.55 of the fp32 MFMA peak).  A 16x16x4 MFMA takes ONE float of A per lane -- lane (li, kq) supplies x[row li][4 s + kq] --
// so a wave's 16 rows are D / 4 registers per lane (64 at D = 256), loaded once; LDS then holds only the codebook tile,
// double-buffered (37 KB: several workgroups per CU, one barrier per tile chunk), stored k-major ([code][kq][s]) so that a
// lane fetches its 8 values of a chunk with two 16-byte reads instead of eight 4-byte ones (row stride 36 floats:
// conflict-free).  Accumulation order per distance is unchanged (dims ascending, 4 per MFMA): results are bit-identical to
// vq_argmin_f32_kernel's, and so are the indices.
//   ND = 32-dim chunks per row (D <= 32 ND; 2, 4 or 8), RW = 16-row fragments per wave (2 where the registers allow: the
//   codebook tile is then read half as often per row).
constexpr int VQR_ES = 36;
template <int ND, int RW>
__global__ __launch_bounds__(256, (ND == 8 && RW == 1) ? 3 : 2) void vq_argmin_f32_reg_kernel(ArgminArgs a) {
  __shared__ __attribute__((aligned(16))) float Es[2][VQ_CT * VQR_ES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const long r0 = (long)blockIdx.x * (64 * RW) + wave * (16 * RW);
  const int D = a.D, K = a.K;
  const bool vec = (D & 3) == 0;

  float xa[RW][ND * 8];
#pragma unroll
  for (int f = 0; f < RW; ++f) {
    const long row = r0 + f * 16 + li;
    const float* xr = a.x + row * D;
#pragma unroll
    for (int j = 0; j < ND * 8; ++j) {
      const int d = 4 * j + kq;
      xa[f][j] = (row < a.N && d < D) ? xr[d] : 0.f;
    }
  }
  float xn[RW][4], best[RW][4];
  int bidx[RW][4];
#pragma unroll
  for (int f = 0; f < RW; ++f)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long row = r0 + f * 16 + kq * 4 + r;
      xn[f][r] = row < a.N ? a.xn[row] : 0.f;
      best[f][r] = __builtin_inff();
      bidx[f][r] = 0x7fffffff;
    }

  // codebook staging: a chunk = 128 codes x 32 dims; thread -> (code = (tid >> 3) + 32 i, dims 4 c8 .. 4 c8 + 3), i = 0..3
  const int c8 = tid & 7, ecode = tid >> 3;
  f32x4 er[4];
  auto load_e = [&](int kt, int dc) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = kt * VQ_CT + ecode + 32 * i, d = dc * VQ_DK + 4 * c8;
      const float* p = a.e + (long)k * D + d;
      if (k < K && vec && d + 3 < D) er[i] = *(const f32x4*)p;
      else {
#pragma unroll
        for (int q = 0; q < 4; ++q) er[i][q] = (k < K && d + q < D) ? p[q] : 0.f;
      }
    }
  };
  auto store_e = [&](int buf) {        // dim 4 s + kq of a chunk -> position kq * 8 + s
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) Es[buf][(ecode + 32 * i) * VQR_ES + q * 8 + c8] = er[i][q];
  };

  const int nkt = (K + VQ_CT - 1) / VQ_CT;
  load_e(0, 0);
  store_e(0);
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    f32x4 acc[RW][8];
#pragma unroll
    for (int f = 0; f < RW; ++f)
#pragma unroll
      for (int ni = 0; ni < 8; ++ni) acc[f][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dc = 0; dc < ND; ++dc) {        // ND is even: chunk dc of every tile sits in buffer dc & 1
      const bool more = !(kt == nkt - 1 && dc == ND - 1);
      if (more) load_e(dc == ND - 1 ? kt + 1 : kt, dc == ND - 1 ? 0 : dc + 1);
      const float* eb = &Es[dc & 1][li * VQR_ES + kq * 8];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        f32x4 b0[4], b1[4];
#pragma unroll
        for (int n4 = 0; n4 < 4; ++n4) {
          const float* p = eb + (half * 4 + n4) * 16 * VQR_ES;
          b0[n4] = *(const f32x4*)p;
          b1[n4] = *(const f32x4*)(p + 4);
        }
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
          for (int n4 = 0; n4 < 4; ++n4) {
            const float bf = s < 4 ? b0[n4][s & 3] : b1[n4][s & 3];
#pragma unroll
            for (int f = 0; f < RW; ++f)
              acc[f][half * 4 + n4] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[f][dc * 8 + s], bf, acc[f][half * 4 + n4], 0, 0, 0);
          }
      }
      if (more) store_e((dc + 1) & 1);
      __syncthreads();
    }
    // distances of this code tile: acc[f][ni][r] = x[row f*16 + kq*4 + r] . e[code ni*16 + li]
#pragma unroll
    for (int ni = 0; ni < 8; ++ni) {
      const int k = kt * VQ_CT + ni * 16 + li;
      if (k < K) {
        const float bn = a.en[k];
#pragma unroll
        for (int f = 0; f < RW; ++f)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float d = (xn[f][r] + bn) - 2.0f * acc[f][ni][r];
            if (d < best[f][r]) {
              best[f][r] = d;
              bidx[f][r] = k;
            }
          }
      }
    }
  }
#pragma unroll
  for (int o = 1; o < 16; o <<= 1)
#pragma unroll
    for (int f = 0; f < RW; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d2 = __shfl_xor(best[f][r], o, 64);
        const int k2 = __shfl_xor(bidx[f][r], o, 64);
        if (d2 < best[f][r] || (d2 == best[f][r] && k2 < bidx[f][r])) {
          best[f][r] = d2;
          bidx[f][r] = k2;
        }
      }
  if (li == 0) {
#pragma unroll
    for (int f = 0; f < RW; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long row = r0 + f * 16 + kq * 4 + r;
        if (row < a.N) {
          a.idx[row] = (int64_t)(bidx[f][r] == 0x7fffffff ? 0 : bidx[f][r]);
          if (a.min_dist) a.min_dist[row] = best[f][r];
        }
      }
  }
}

// q_st = x + (E[idx] - x); per-workgroup partial of sum (E[idx]-x)^2; histogram of idx.
__global__ __launch_bounds__(256) void vq_gather_loss_kernel(const float* x, const float* e, const int64_t* idx,
                                                             float* q_st, float* partials, int32_t* hist, long N, int K,
                                                             int D) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // consecutive rows per workgroup (coalesced), histogram privatised in LDS when it fits (hot codes would
  // otherwise serialise on one global counter)
  extern __shared__ int lhist[];
  const bool use_lds = hist && K <= 8192;
  if (use_lds) {
    for (int k = threadIdx.x; k < K; k += 256) lhist[k] = 0;
    __syncthreads();
  }
  const long per = (N + gridDim.x - 1) / gridDim.x;
  const long rb = (long)blockIdx.x * per, re = rb + per < N ? rb + per : N;
  float s = 0.f;
  for (long r = rb + wave; r < re; r += 4) {
    const long k = idx[r];
    for (int d = lane; d < D; d += 64) {
      const float xv = x[r * D + d];
      const float diff = e[k * D + d] - xv;
      q_st[r * D + d] = xv + diff;
      s += diff * diff;
    }
    if (lane == 0 && hist) atomicAdd(use_lds ? &lhist[k] : &hist[k], 1);
  }
  if (use_lds) {
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += 256)
      if (lhist[k]) atomicAdd(&hist[k], lhist[k]);
  }
  s = wave_sum(s);
  __shared__ float red[4];
  if (lane == 0) red[wave] = s;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void vq_finalize_kernel(const float* partials, int nparts, const int32_t* hist,
                                                          float* out, long N, int K, int D, float beta) {
  __shared__ float red[256];
  const int t = threadIdx.x;
  float s = 0.f;
  for (int i = t; i < nparts; i += 256) s += partials[i];
  red[t] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) red[t] += red[t + o];
    __syncthreads();
  }
  const float sq = red[0];
  __syncthreads();
  float h = 0.f;
  for (int k = t; k < K; k += 256) {
    const float p = (float)hist[k] / (float)N;
    h += p * logf(p + 1e-10f);
  }
  red[t] = h;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) red[t] += red[t + o];
    __syncthreads();
  }
  if (t == 0) {
    const float m = sq / (float)((double)N * (double)D);
    out[0] = m + beta * m;  // q_latent + beta * e_latent, both equal m in value (:46-52)
    out[1] = expf(-red[0]);
  }
}

// dx = g - gl*cx*(E[idx] - x)  (one wave per row)
__global__ __launch_bounds__(256) void vq_backward_dx_kernel(const float* g, const float* grad_loss, const float* x,
                                                             const float* e, const int64_t* idx, float* dx, long N, int D,
                                                             float cx) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float sx = (grad_loss ? grad_loss[0] : 1.f) * cx;
  for (long r = (long)blockIdx.x * 4 + wave; r < N; r += (long)gridDim.x * 4) {
    const long k = idx[r];
    for (int d = lane; d < D; d += 64) {
      const float diff = e[k * D + d] - x[r * D + d];  // q - x
      dx[r * D + d] = (g ? g[r * D + d] : 0.f) - sx * diff;
    }
  }
}

// dE[k] += gl*ce * sum_{n: idx_n = k} (E[k] - x_n), WITHOUT atomics.  Workgroup (k, p) owns code k on the p-th of P
// contiguous row segments: it scans the segment's idx in row order, compacts the matching row numbers into an LDS
// list (block prefix sum, order preserved) and sums those rows in a fixed pattern -- G thread groups take list
// positions g, g+G, ... and are combined in group order -- into partial[p][k][:].  A second tiny kernel adds the P
// partials in segment order.  Every sum's order depends on the data only: bitwise reproducible from run to run
// (float atomics are not), and a hot code is spread over P workgroups x G groups instead of serialising.
constexpr int VQL_CAP = 2048;       // list capacity (rows of one code gathered before a flush)
constexpr int VQL_CHUNK = 1024;     // rows scanned per iteration (4 per thread)
template <int VEC>                  // VEC = 4: D % 4 == 0, float4 lanes; VEC = 1: any D <= 256
__global__ __launch_bounds__(256) void vq_codebook_grad_kernel(const float* grad_loss, const float* x, const float* e,
                                                               const int64_t* idx, float* partial, long N, int K, int D,
                                                               long rows_per_part, float ce) {
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  __shared__ int list[VQL_CAP];
  __shared__ int wave_tot[4];
  __shared__ vec_t comb[256];
  const int k = blockIdx.x, part = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float se = (grad_loss ? grad_loss[0] : 1.f) * ce;
  const int TD = D / VEC;                          // lanes along d (<= 128 for VEC = 4, <= 256 for VEC = 1)
  const int G = 256 / TD;                          // thread groups
  const int grp = tid / TD, ld = tid - grp * TD;
  const bool summer = grp < G;
  vec_t ek, acc;
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    ek[v] = summer ? e[(long)k * D + ld * VEC + v] : 0.f;
    acc[v] = 0.f;
  }
  int nlist = 0;
  auto flush = [&]() {
    if (!summer) return;
    int i = grp;
    for (; i + G < nlist; i += 2 * G) {            // two rows in flight per thread
      const vec_t a = *(const vec_t*)(x + (long)list[i] * D + ld * VEC);
      const vec_t b = *(const vec_t*)(x + (long)list[i + G] * D + ld * VEC);
      acc += se * (ek - a);
      acc += se * (ek - b);
    }
    if (i < nlist) acc += se * (ek - *(const vec_t*)(x + (long)list[i] * D + ld * VEC));
  };
  const long rbeg = (long)part * rows_per_part;
  const long rend = rbeg + rows_per_part < N ? rbeg + rows_per_part : N;
  for (long base = rbeg; base < rend; base += VQL_CHUNK) {
    int mine[4], cnt = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long r = base + 4 * tid + j;
      if (r < rend && idx[r] == k) mine[cnt++] = (int)r;
    }
    int incl = cnt;                                // inclusive prefix over the wave, then over the block
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int up = __shfl_up(incl, off, 64);
      if (lane >= off) incl += up;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      if (w < wave) before += wave_tot[w];
      total += wave_tot[w];
    }
    const int pos = nlist + before + incl - cnt;
    for (int j = 0; j < cnt; ++j) list[pos + j] = mine[j];
    nlist += total;
    __syncthreads();
    if (nlist > VQL_CAP - VQL_CHUNK) {             // the next chunk might not fit: sum what is gathered
      flush();
      nlist = 0;
      __syncthreads();
    }
  }
  flush();
  comb[tid] = acc;
  __syncthreads();
  if (tid < TD) {
    vec_t s = comb[tid];
    for (int g = 1; g < G; ++g) s += comb[g * TD + tid];
    *(vec_t*)(partial + ((long)part * K + k) * D + tid * VEC) = s;
  }
}

// dE (+)= sum_p partial[p], segment order
__global__ __launch_bounds__(256) void vq_codebook_grad_reduce_kernel(const float* partial, float* dE, long kd, int P) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < kd; i += (long)gridDim.x * 256) {
    float s = 0.f;
    for (int p = 0; p < P; ++p) s += partial[(long)p * kd + i];
    dE[i] += s;
  }
}

__global__ __launch_bounds__(256) void onehot_kernel(const int64_t* idx, float* enc, long N, int K) {
  const long total = N * (long)K;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long r = e / K;
    const int k = (int)(e - r * K);
    enc[e] = (idx[r] == k) ? 1.f : 0.f;
  }
}

}  // namespace alvq

using namespace alvq;

static int grid_for(long work_items, int per_block) {
  long g = (work_items + per_block - 1) / per_block;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

extern "C" int64_t alvq_vq_argmin_workspace_bytes(int64_t N, int K, int D) {
  if (N <= 0 || K <= 0 || D <= 0) return -1;
  return (int64_t)(N + K) * (int64_t)sizeof(float);
}

extern "C" int alvq_vq_argmin_f32(const float* x, const float* codebook, int64_t* idx, float* min_dist,
                                  void* workspace, int64_t N, int K, int D, void* stream) {
  ALVQ_REQUIRE(x && codebook && idx && workspace, ALVQ_EINVAL, "alvq_vq_argmin_f32: null pointer");
  ALVQ_REQUIRE(N > 0 && K > 0 && D > 0, ALVQ_EINVAL, "alvq_vq_argmin_f32: bad dims N=%ld K=%d D=%d", (long)N, K, D);
  ALVQ_REQUIRE(D <= 512, ALVQ_EUNSUPPORTED, "alvq_vq_argmin_f32: D=%d > 512 does not fit the stationary LDS tile", D);
  ALVQ_REQUIRE(N / 64 < (1L << 31) - 2, ALVQ_EUNSUPPORTED, "alvq_vq_argmin_f32: N too large");
  hipStream_t s = (hipStream_t)stream;
  float* xn = (float*)workspace;
  float* en = xn + N;
  hipLaunchKernelGGL(row_sqnorm_kernel, dim3(grid_for(N, 4)), dim3(256), 0, s, x, xn, (long)N, D);
  hipLaunchKernelGGL(row_sqnorm_kernel, dim3(grid_for(K, 4)), dim3(256), 0, s, codebook, en, (long)K, D);
  const int Dp = (D + VQ_DK - 1) / VQ_DK * VQ_DK;
  const int XSTR = vq_pad32(Dp, 2);
  ArgminArgs a{x, codebook, xn, en, idx, min_dist, (long)N, K, D, Dp, XSTR};
  static DeviceOnce attr_set;
  if (attr_set.need()) {
    (void)hipFuncSetAttribute((const void*)vq_argmin_f32_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)vq_argmin_f32_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  // D <= 256: the x rows in registers, codebook tile double-buffered (option "vq_reg" = 0: the LDS-stationary kernel)
  if (D <= 256 && option(OPT_VQ_REG) != 0) {
    const int nd = D <= 64 ? 2 : (D <= 128 ? 4 : 8);
    // two row fragments per wave (the codebook tile read half as often per row) where the registers allow (D <= 128) and the
    // problem still gives every CU two workgroups; otherwise 64-row workgroups
    const bool two = nd != 8 && N >= 128L * 512;
    const int rows_wg = two ? 128 : 64;
    const dim3 grid((unsigned)((N + rows_wg - 1) / rows_wg));
    if (nd == 2 && two) hipLaunchKernelGGL((vq_argmin_f32_reg_kernel<2, 2>), grid, dim3(256), 0, s, a);
    else if (nd == 2) hipLaunchKernelGGL((vq_argmin_f32_reg_kernel<2, 1>), grid, dim3(256), 0, s, a);
    else if (nd == 4 && two) hipLaunchKernelGGL((vq_argmin_f32_reg_kernel<4, 2>), grid, dim3(256), 0, s, a);
    else if (nd == 4) hipLaunchKernelGGL((vq_argmin_f32_reg_kernel<4, 1>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((vq_argmin_f32_reg_kernel<8, 1>), grid, dim3(256), 0, s, a);
    return check_launch("alvq_vq_argmin_f32");
  }
  // 8 waves (128 rows) from D = 128 up: a 4-wave workgroup would leave one wave per SIMD there, and 128-row blocks
  // halve the codebook-tile traffic per row (+7 % at the speech size)
  const bool wide = D >= 128 && (size_t)(128 * XSTR + VQ_CT * VQ_ES) * sizeof(float) <= 160 * 1024;
  const int rb = wide ? 128 : 64;
  const size_t lds = (size_t)(rb * XSTR + VQ_CT * VQ_ES) * sizeof(float);
  const dim3 grid((unsigned)((N + rb - 1) / rb));
  if (wide) hipLaunchKernelGGL(vq_argmin_f32_kernel<8>, grid, dim3(512), lds, s, a);
  else hipLaunchKernelGGL(vq_argmin_f32_kernel<4>, grid, dim3(256), lds, s, a);
  return check_launch("alvq_vq_argmin_f32");
}

extern "C" int alvq_vq_gather_loss_f32(const float* x, const float* codebook, const int64_t* idx, float* q_st,
                                       float* sq_partials, int32_t* hist, int64_t N, int K, int D, void* stream) {
  ALVQ_REQUIRE(x && codebook && idx && q_st && sq_partials, ALVQ_EINVAL, "alvq_vq_gather_loss_f32: null pointer");
  ALVQ_REQUIRE(N > 0 && K > 0 && D > 0, ALVQ_EINVAL, "alvq_vq_gather_loss_f32: bad dims");
  hipLaunchKernelGGL(vq_gather_loss_kernel, dim3(VQ_PARTIALS), dim3(256), (hist && K <= 8192) ? K * sizeof(int) : 0,
                     (hipStream_t)stream, x, codebook, idx, q_st, sq_partials, hist, (long)N, K, D);
  return check_launch("alvq_vq_gather_loss_f32");
}

extern "C" int alvq_vq_finalize_f32(const float* sq_partials, const int32_t* hist, float* out, int64_t N, int K, int D,
                                    float beta, void* stream) {
  ALVQ_REQUIRE(sq_partials && hist && out, ALVQ_EINVAL, "alvq_vq_finalize_f32: null pointer");
  ALVQ_REQUIRE(N > 0 && K > 0 && D > 0, ALVQ_EINVAL, "alvq_vq_finalize_f32: bad dims");
  hipLaunchKernelGGL(vq_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, sq_partials, VQ_PARTIALS, hist,
                     out, (long)N, K, D, beta);
  return check_launch("alvq_vq_finalize_f32");
}

constexpr int VQ_PARTS_MAX = 16;   // row segments a code's gather is split over

static int vq_parts(int64_t N) {
  int64_t p = N / 4096;
  return p < 1 ? 1 : (p > VQ_PARTS_MAX ? VQ_PARTS_MAX : (int)p);
}

extern "C" int64_t alvq_vq_backward_workspace_bytes(int K, int D) {
  return (K <= 0 || D <= 0) ? -1 : (int64_t)VQ_PARTS_MAX * K * D * (int64_t)sizeof(float);
}

extern "C" int alvq_vq_backward_f32(const float* g, const float* grad_loss, const float* x, const float* codebook,
                                    const int64_t* idx, float* dx, float* dE, void* workspace, int64_t N, int K, int D,
                                    float beta, void* stream) {
  ALVQ_REQUIRE(x && codebook && idx && (dx || dE), ALVQ_EINVAL, "alvq_vq_backward_f32: null pointer");
  ALVQ_REQUIRE(N > 0 && K > 0 && D > 0, ALVQ_EINVAL, "alvq_vq_backward_f32: bad dims");
  ALVQ_REQUIRE(!dE || workspace, ALVQ_EINVAL, "alvq_vq_backward_f32: the codebook gradient needs the workspace");
  ALVQ_REQUIRE(N < (1L << 31) && K <= 65535 && ((D % 4 == 0 && D <= 512) || D <= 256), ALVQ_EUNSUPPORTED,
               "alvq_vq_backward_f32: N=%ld K=%d D=%d outside the supported range", (long)N, K, D);
  const double nd = (double)N * (double)D;
  hipStream_t s = (hipStream_t)stream;
  if (dx)
    hipLaunchKernelGGL(vq_backward_dx_kernel, dim3(grid_for(N, 4)), dim3(256), 0, s, g, grad_loss, x, codebook, idx, dx,
                       (long)N, D, (float)(2.0 * beta / nd));
  if (dE) {
    const int P = vq_parts(N);
    const long rpp = ((N + P - 1) / P + VQL_CHUNK - 1) / VQL_CHUNK * VQL_CHUNK;
    float* partial = (float*)workspace;
    if (D % 4 == 0)
      hipLaunchKernelGGL(vq_codebook_grad_kernel<4>, dim3(K, P), dim3(256), 0, s, grad_loss, x, codebook, idx, partial, (long)N,
                         K, D, rpp, (float)(2.0 / nd));
    else
      hipLaunchKernelGGL(vq_codebook_grad_kernel<1>, dim3(K, P), dim3(256), 0, s, grad_loss, x, codebook, idx, partial, (long)N,
                         K, D, rpp, (float)(2.0 / nd));
    hipLaunchKernelGGL(vq_codebook_grad_reduce_kernel, dim3(grid_for((long)K * D, 256)), dim3(256), 0, s,
                       (const float*)partial, dE, (long)K * D, P);
  }
  return check_launch("alvq_vq_backward_f32");
}

extern "C" int alvq_onehot_f32(const int64_t* idx, float* encodings, int64_t N, int K, void* stream) {
  ALVQ_REQUIRE(idx && encodings, ALVQ_EINVAL, "alvq_onehot_f32: null pointer");
  ALVQ_REQUIRE(N > 0 && K > 0, ALVQ_EINVAL, "alvq_onehot_f32: bad dims");
  hipLaunchKernelGGL(onehot_kernel, dim3(grid_for(N * (long)K, 256 * 8)), dim3(256), 0, (hipStream_t)stream, idx,
                     encodings, (long)N, K);
  return check_launch("alvq_onehot_f32");
}
