// HBM-bound helpers around the conv/VQ core: jitter gather, per-frame standardise, MSE, add, transpose, Adam.
// All are coalesced along the fastest (L) axis; reductions use fixed-order trees (bitwise reproducible).
#include "alvq_common.h"

namespace alvq {

constexpr int EW_PARTIALS = 1024;

// y[row][l] = x[row][src[l]]  (forward)   |   dx[row][l] = (src[l]==l) ? dy[row][l] : 0  (backward)
// reference: vq_vae/modules/jitter.py:50-68 (replaced columns are copies of detached values).
__global__ __launch_bounds__(256) void jitter_kernel(const float* x, const int32_t* src, float* y, long rows, int L,
                                                     int backward) {
  const long total = rows * (long)L;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long r = e / L;
    const int l = (int)(e - r * L);
    const int sl = src[l];
    y[e] = backward ? (sl == l ? x[e] : 0.f) : x[r * L + sl];
  }
}

// Per (b, l): mean and unbiased std over the C channels (scripts/train_speech.py:63-64).
// Workgroup = 64 positions x 4 channel groups; two passes (mean, then squared deviations).
__global__ __launch_bounds__(256) void standardise_kernel(const float* x, float* y, int B, int C, int L, int take_abs) {
  const int lt = threadIdx.x & 63, cg = threadIdx.x >> 6;
  const int ltiles = (L + 63) / 64;
  const int b = blockIdx.x / ltiles, l = (blockIdx.x % ltiles) * 64 + lt;
  const bool ok = l < L;
  const float* xb = x + (long)b * C * L;
  float* yb = y + (long)b * C * L;
  __shared__ float red[4][64];
  float s = 0.f;
  if (ok)
    for (int c = cg; c < C; c += 4) {
      const float v = xb[(long)c * L + l];
      s += take_abs ? fabsf(v) : v;
    }
  red[cg][lt] = s;
  __syncthreads();
  const float mean = ((red[0][lt] + red[1][lt]) + (red[2][lt] + red[3][lt])) / (float)C;
  __syncthreads();
  float q = 0.f;
  if (ok)
    for (int c = cg; c < C; c += 4) {
      float v = xb[(long)c * L + l];
      v = (take_abs ? fabsf(v) : v) - mean;
      q += v * v;
    }
  red[cg][lt] = q;
  __syncthreads();
  const float var = ((red[0][lt] + red[1][lt]) + (red[2][lt] + red[3][lt])) / (float)(C - 1);
  const float inv = 1.f / (sqrtf(var) + 1e-8f);
  if (ok)
    for (int c = cg; c < C; c += 4) {
      float v = xb[(long)c * L + l];
      v = take_abs ? fabsf(v) : v;
      yb[(long)c * L + l] = (v - mean) / (sqrtf(var) + 1e-8f);
    }
  (void)inv;
}

// Same arithmetic for C <= 256 with the column held in registers: one read of x instead of three, and all of a
// thread's loads are in flight at once (the three-pass form is latency-bound at ~1 TB/s).
__global__ __launch_bounds__(256) void standardise_regs_kernel(const float* x, float* y, int B, int C, int L, int take_abs) {
  const int lt = threadIdx.x & 63, cg = threadIdx.x >> 6;
  const int ltiles = (L + 63) / 64;
  const int b = blockIdx.x / ltiles, l = (blockIdx.x % ltiles) * 64 + lt;
  const bool ok = l < L;
  const float* xb = x + (long)b * C * L + l;
  float* yb = y + (long)b * C * L + l;
  __shared__ float red[4][64];
  float v[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    const int c = cg + 4 * i;
    float t = (ok && c < C) ? xb[(long)c * L] : 0.f;
    v[i] = take_abs ? fabsf(t) : t;
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 64; ++i) s += v[i];            // same order as the three-pass kernel (zeros past C add nothing)
  red[cg][lt] = s;
  __syncthreads();
  const float mean = ((red[0][lt] + red[1][lt]) + (red[2][lt] + red[3][lt])) / (float)C;
  __syncthreads();
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    const float d = v[i] - mean;
    q += (cg + 4 * i < C) ? d * d : 0.f;
  }
  red[cg][lt] = q;
  __syncthreads();
  const float var = ((red[0][lt] + red[1][lt]) + (red[2][lt] + red[3][lt])) / (float)(C - 1);
  const float den = sqrtf(var) + 1e-8f;
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    const int c = cg + 4 * i;
    if (ok && c < C) yb[(long)c * L] = (v[i] - mean) / den;
  }
}

__global__ __launch_bounds__(256) void mse_partial_kernel(const float* a, const float* b, float* partials, long n) {
  float s = 0.f;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    const float d = a[e] - b[e];
    s += d * d;
  }
  s = wave_sum(s);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void mse_final_kernel(const float* partials, int nparts, float* loss, long n) {
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
  if (t == 0) loss[0] = red[0] / (float)n;
}

__global__ __launch_bounds__(256) void mse_backward_kernel(const float* a, const float* b, const float* gscale,
                                                           float* grad, long n, float two_over_n) {
  const float g = (gscale ? gscale[0] : 1.f) * two_over_n;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) grad[e] = g * (a[e] - b[e]);
}

// out = t > 0 ? dy : 0   (ReLU backward with the saved post-ReLU activation; with dy == t it is ReLU itself)
__global__ __launch_bounds__(256) void relu_mask_kernel(const float* dy, const float* t, float* out, long n) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) out[e] = t[e] > 0.f ? dy[e] : 0.f;
}

__global__ __launch_bounds__(256) void fill_kernel(float* out, float v, long n) {
  const f32x4 v4 = {v, v, v, v};
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) ((f32x4*)out)[i] = v4;
  for (long i = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = v;
}

__global__ __launch_bounds__(256) void add_kernel(const float* a, const float* b, float* out, long n) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) out[e] = a[e] + b[e];
}

// (B,R,C) -> (B,C,R) through a 32x33 LDS tile (coalesced on both sides).
__global__ __launch_bounds__(256) void transpose_kernel(const float* x, float* y, int B, int R, int C) {
  __shared__ float tile[32][33];
  const int ct = (C + 31) / 32, rt = (R + 31) / 32;
  int id = blockIdx.x;
  const int b = id / (ct * rt);
  id -= b * ct * rt;
  const int r0 = (id / ct) * 32, c0 = (id % ct) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const float* xb = x + (long)b * R * C;
  float* yb = y + (long)b * R * C;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    tile[ty + 8 * i][tx] = (r < R && c < C) ? xb[(long)r * C + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, r = r0 + tx;
    if (r < R && c < C) yb[(long)c * R + r] = tile[tx][ty + 8 * i];
  }
}

// torch.optim.Adam single-tensor arithmetic (amsgrad=False, no weight decay, maximize=False).
__global__ __launch_bounds__(256) void adam_kernel(float* p, const float* g, float* m, float* v, long n, float lr_bc1,
                                                   float beta1, float beta2, float eps, float bc2_sqrt, float gscale) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    const float gr = g[e] * gscale;
    const float mm = m[e] + (gr - m[e]) * (1.f - beta1);            // exp_avg.lerp_(grad, 1-beta1)
    const float vv = v[e] * beta2 + (1.f - beta2) * gr * gr;        // exp_avg_sq.mul_(b2).addcmul_(g,g,1-b2)
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    m[e] = mm;
    v[e] = vv;
    p[e] = p[e] - lr_bc1 * (mm / denom);                            // addcdiv_(m, denom, value=-lr/bc1)
  }
}

// Same arithmetic with the step-dependent scalars read from device memory, so a captured hipGraph can be
// replayed every step: sc = {lr/bias_correction1, sqrt(bias_correction2), grad_scale}.
// ``skip`` (nullable): a device float; non-zero = this step's values saturated an fp16-range format on some rank (the
// slot travels through the step's all-reduce inside the flat gradient buffer) -- the update is NOT applied: parameters and
// moments stay bit for bit what they were (the dynamic-loss-scaling "skip" pattern, no host sync).
__global__ __launch_bounds__(256) void adam_dev_kernel(float* p, const float* g, float* m, float* v, long n, const float* sc,
                                                       float beta1, float beta2, float eps, const float* skip) {
  if (skip && *skip != 0.f) return;
  const float lr_bc1 = sc[0], bc2_sqrt = sc[1], gscale = sc[2];
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    const float gr = g[e] * gscale;
    const float mm = m[e] + (gr - m[e]) * (1.f - beta1);
    const float vv = v[e] * beta2 + (1.f - beta2) * gr * gr;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    m[e] = mm;
    v[e] = vv;
    p[e] = p[e] - lr_bc1 * (mm / denom);
  }
}

// Advance the device-side step counter and derive that step's scalars from it (double arithmetic, as the host
// would): sc = {lr/(1-beta1^t), sqrt(1-beta2^t), grad_scale, t}.  No host buffer is involved, so nothing races when
// the host queues many steps ahead of the device, and the launch can be replayed from a graph.
// Guarded form (``prev_skip`` non-null: the skip slot of the flat gradient buffer, still holding the PREVIOUS step's
// verdict because the buffer is zeroed inside the step's body): a skipped step does not count -- t stays where it was,
// sc[4] (skipped steps since the counter was last read) goes up by one -- and the range flag's bits of everything that
// ran since the last step (evaluation passes, the previous step itself) move into the sticky word, so that the step about
// to run sees only its own.
__global__ void adam_advance_kernel(float* sc, double lr, double beta1, double beta2, double grad_scale, const float* prev_skip,
                                    int* range_flag, int* range_sticky) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double t = (double)sc[3];
  if (prev_skip && *prev_skip != 0.f) sc[4] += 1.f;
  else t += 1.0;
  if (prev_skip && range_flag) {
    const int f = *range_flag;
    if (f) { *range_sticky |= f; *range_flag = 0; }
  }
  const double tt = t < 1.0 ? 1.0 : t;        // the very first step was skipped: the retry is step 1 again (t itself stays 0,
                                              // so that the first APPLIED step is counted -- and bias-corrected -- as step 1)
  sc[0] = (float)(lr / (1.0 - pow(beta1, tt)));
  sc[1] = (float)sqrt(1.0 - pow(beta2, tt));
  sc[2] = (float)grad_scale;
  sc[3] = (float)t;
}

// slot = 1 if the range flag holds any bit (set since the step's alvq_adam_advance_f32 cleared it), else 0: the last
// launch of a step's backward; the slot is then summed over the ranks with the gradients
__global__ void range_flag_to_slot_kernel(float* slot, const int* range_flag) { *slot = *range_flag ? 1.f : 0.f; }


// y[r] = mean_l x[r][l] (one wave per row; fixed-order sum: lane partials over l = lane, lane + 64, ... then the butterfly)
__global__ __launch_bounds__(256) void row_mean_kernel(const float* x, float* y, long rows, int L) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* p = x + r * L;
  float s = 0.f;
  for (int l = lane; l < L; l += 64) s += p[l];
  s = wave_sum(s);
  if (lane == 0) y[r] = s / (float)L;
}

// dx[r][l] = dy[r] / L
__global__ __launch_bounds__(256) void row_mean_backward_kernel(const float* dy, float* dx, long n, int L) {
  const float inv = 1.f / (float)L;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dx[i] = dy[i / L] * inv;
}

}  // namespace alvq

using namespace alvq;

static int ew_grid(long n) {
  long g = (n + 256L * 4 - 1) / (256L * 4);
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

extern "C" int alvq_jitter_gather_f32(const float* x, const int32_t* src, float* y, int64_t rows, int L, int backward,
                                      void* stream) {
  ALVQ_REQUIRE(x && src && y, ALVQ_EINVAL, "alvq_jitter_gather_f32: null pointer");
  ALVQ_REQUIRE(rows > 0 && L > 0, ALVQ_EINVAL, "alvq_jitter_gather_f32: bad dims");
  hipLaunchKernelGGL(jitter_kernel, dim3(ew_grid(rows * L)), dim3(256), 0, (hipStream_t)stream, x, src, y, (long)rows, L,
                     backward);
  return check_launch("alvq_jitter_gather_f32");
}

extern "C" int alvq_standardise_f32(const float* x, float* y, int B, int C, int L, int take_abs, void* stream) {
  ALVQ_REQUIRE(x && y, ALVQ_EINVAL, "alvq_standardise_f32: null pointer");
  ALVQ_REQUIRE(B > 0 && C > 1 && L > 0, ALVQ_EINVAL, "alvq_standardise_f32: bad dims (C must be > 1)");
  if (C <= 256)
    hipLaunchKernelGGL(standardise_regs_kernel, dim3(B * ((L + 63) / 64)), dim3(256), 0, (hipStream_t)stream, x, y, B, C, L,
                       take_abs);
  else
    hipLaunchKernelGGL(standardise_kernel, dim3(B * ((L + 63) / 64)), dim3(256), 0, (hipStream_t)stream, x, y, B, C, L,
                       take_abs);
  return check_launch("alvq_standardise_f32");
}

extern "C" int alvq_mse_f32(const float* a, const float* b, float* loss, void* workspace, int64_t n, void* stream) {
  ALVQ_REQUIRE(a && b && loss && workspace, ALVQ_EINVAL, "alvq_mse_f32: null pointer");
  ALVQ_REQUIRE(n > 0, ALVQ_EINVAL, "alvq_mse_f32: n <= 0");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(mse_partial_kernel, dim3(EW_PARTIALS), dim3(256), 0, s, a, b, (float*)workspace, (long)n);
  hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(256), 0, s, (const float*)workspace, EW_PARTIALS, loss, (long)n);
  return check_launch("alvq_mse_f32");
}

extern "C" int alvq_mse_backward_f32(const float* a, const float* b, const float* grad_loss, float* grad, int64_t n,
                                     void* stream) {
  ALVQ_REQUIRE(a && b && grad, ALVQ_EINVAL, "alvq_mse_backward_f32: null pointer");
  ALVQ_REQUIRE(n > 0, ALVQ_EINVAL, "alvq_mse_backward_f32: n <= 0");
  hipLaunchKernelGGL(mse_backward_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, a, b, grad_loss, grad,
                     (long)n, (float)(2.0 / (double)n));
  return check_launch("alvq_mse_backward_f32");
}

extern "C" int alvq_fill_f32(float* out, float value, int64_t n, void* stream) {
  ALVQ_REQUIRE(out, ALVQ_EINVAL, "alvq_fill_f32: null pointer");
  ALVQ_REQUIRE(n > 0, ALVQ_EINVAL, "alvq_fill_f32: n <= 0");
  ALVQ_REQUIRE(((uintptr_t)out & 15) == 0, ALVQ_EINVAL, "alvq_fill_f32: out must be 16-byte aligned");
  hipLaunchKernelGGL(fill_kernel, dim3(ew_grid((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, out, value, (long)n);
  return check_launch("alvq_fill_f32");
}

extern "C" int alvq_add_f32(const float* a, const float* b, float* out, int64_t n, void* stream) {
  ALVQ_REQUIRE(a && b && out, ALVQ_EINVAL, "alvq_add_f32: null pointer");
  ALVQ_REQUIRE(n > 0, ALVQ_EINVAL, "alvq_add_f32: n <= 0");
  hipLaunchKernelGGL(add_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, a, b, out, (long)n);
  return check_launch("alvq_add_f32");
}

extern "C" int alvq_relu_mask_f32(const float* dy, const float* t, float* out, int64_t n, void* stream) {
  ALVQ_REQUIRE(dy && t && out, ALVQ_EINVAL, "alvq_relu_mask_f32: null pointer");
  ALVQ_REQUIRE(n > 0, ALVQ_EINVAL, "alvq_relu_mask_f32: n <= 0");
  hipLaunchKernelGGL(relu_mask_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, dy, t, out, (long)n);
  return check_launch("alvq_relu_mask_f32");
}

extern "C" int alvq_row_mean_f32(const float* x, float* y, int64_t rows, int L, void* stream) {
  ALVQ_REQUIRE(x && y, ALVQ_EINVAL, "alvq_row_mean_f32: null pointer");
  ALVQ_REQUIRE(rows > 0 && L > 0, ALVQ_EINVAL, "alvq_row_mean_f32: bad dims");
  ALVQ_REQUIRE((rows + 3) / 4 < (1L << 31), ALVQ_EUNSUPPORTED, "alvq_row_mean_f32: too many rows");
  hipLaunchKernelGGL(row_mean_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, y, (long)rows, L);
  return check_launch("alvq_row_mean_f32");
}

extern "C" int alvq_row_mean_backward_f32(const float* dy, float* dx, int64_t rows, int L, void* stream) {
  ALVQ_REQUIRE(dy && dx, ALVQ_EINVAL, "alvq_row_mean_backward_f32: null pointer");
  ALVQ_REQUIRE(rows > 0 && L > 0, ALVQ_EINVAL, "alvq_row_mean_backward_f32: bad dims");
  hipLaunchKernelGGL(row_mean_backward_kernel, dim3(ew_grid(rows * L)), dim3(256), 0, (hipStream_t)stream, dy, dx, (long)rows * L, L);
  return check_launch("alvq_row_mean_backward_f32");
}

extern "C" int alvq_transpose_f32(const float* x, float* y, int B, int R, int C, void* stream) {
  ALVQ_REQUIRE(x && y, ALVQ_EINVAL, "alvq_transpose_f32: null pointer");
  ALVQ_REQUIRE(B > 0 && R > 0 && C > 0, ALVQ_EINVAL, "alvq_transpose_f32: bad dims");
  const long blocks = (long)B * ((R + 31) / 32) * ((C + 31) / 32);
  ALVQ_REQUIRE(blocks < (1L << 31), ALVQ_EUNSUPPORTED, "alvq_transpose_f32: too large");
  hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, B, R, C);
  return check_launch("alvq_transpose_f32");
}

extern "C" int alvq_adam_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int step,
                             float lr, float beta1, float beta2, float eps, float grad_scale, void* stream) {
  ALVQ_REQUIRE(param && grad && exp_avg && exp_avg_sq, ALVQ_EINVAL, "alvq_adam_f32: null pointer");
  ALVQ_REQUIRE(n > 0 && step >= 1, ALVQ_EINVAL, "alvq_adam_f32: bad n/step");
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq,
                     (long)n, (float)((double)lr / bc1), beta1, beta2, eps, (float)sqrt(bc2), grad_scale);
  return check_launch("alvq_adam_f32");
}

extern "C" int alvq_adam_dev_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                                 const float* scalars, float beta1, float beta2, float eps, const float* skip, void* stream) {
  ALVQ_REQUIRE(param && grad && exp_avg && exp_avg_sq && scalars, ALVQ_EINVAL, "alvq_adam_dev_f32: null pointer");
  ALVQ_REQUIRE(n > 0, ALVQ_EINVAL, "alvq_adam_dev_f32: n <= 0");
  hipLaunchKernelGGL(adam_dev_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq,
                     (long)n, scalars, beta1, beta2, eps, skip);
  return check_launch("alvq_adam_dev_f32");
}

extern "C" int alvq_adam_advance_f32(float* scalars, double lr, double beta1, double beta2, double grad_scale,
                                     const float* prev_skip, void* stream) {
  ALVQ_REQUIRE(scalars, ALVQ_EINVAL, "alvq_adam_advance_f32: null pointer");
  int* flag = prev_skip ? fx_range_flag_ptr() : nullptr;
  int* sticky = prev_skip ? fx_range_sticky_ptr() : nullptr;
  ALVQ_REQUIRE(!prev_skip || (flag && sticky), ALVQ_EINVAL, "alvq_adam_advance_f32: the range flag's device address is unavailable");
  hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, scalars, lr, beta1, beta2, grad_scale,
                     prev_skip, flag, sticky);
  return check_launch("alvq_adam_advance_f32");
}

extern "C" int alvq_range_flag_to_slot(float* slot, void* stream) {
  ALVQ_REQUIRE(slot, ALVQ_EINVAL, "alvq_range_flag_to_slot: null pointer");
  int* flag = fx_range_flag_ptr();
  ALVQ_REQUIRE(flag, ALVQ_EINVAL, "alvq_range_flag_to_slot: the range flag's device address is unavailable");
  hipLaunchKernelGGL(range_flag_to_slot_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, slot, (const int*)flag);
  return check_launch("alvq_range_flag_to_slot");
}
