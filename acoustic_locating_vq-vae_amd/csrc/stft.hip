// STFT power-spectrogram front end (reference: scripts/genereate_dataset.py:90-91,37,39,47-49 --
// torchaudio.transforms.Spectrogram(n_fft, hop, power=None, center=True, pad=0, normalized=True) then |.|^2).
//
// One workgroup = 8 consecutive frames of one waveform.  The reflect-padded, Hann-windowed frames and a
// cos/sin twiddle table (built in fp64, stored fp32) live in LDS; thread k accumulates bin k of all 8 frames
// (frame samples are LDS broadcasts).  Traffic per utterance is 0.3 MB in / 0.8 MB out: this is HBM/launch
// bound and 1e4 x fewer FLOPs than one model step, so a direct DFT on the vector ALU is the right size.
#include "alvq_common.h"

namespace alvq {

constexpr int ST_FT = 8;  // frames per workgroup

// R = float: the speech spectrogram (fp32 waveform -> complex64 in the reference); R = double: the echoed
// signal, which the reference keeps in float64 (scipy convolve output, genereate_dataset.py:38-39).
// CPLX = false: power[b][k][t] = |X|^2 / sum(w^2);  CPLX = true: spec[b][k][t] = (re, im) / sqrt(sum(w^2)), interleaved.
template <typename R, bool CPLX>
__global__ __launch_bounds__(256) void stft_power_kernel(const R* wave, R* power, int B, int S, int N, int hop, int T) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm_raw[];
  R* sm = (R*)sm_raw;
  R* cs = sm;               // [N]
  R* sn = sm + N;           // [N]
  R* fr = sm + 2 * N;       // [ST_FT][N]
  __shared__ R wsum_s;
  const int tid = threadIdx.x;
  const int ttiles = (T + ST_FT - 1) / ST_FT;
  const int b = blockIdx.x / ttiles, t0 = (blockIdx.x % ttiles) * ST_FT;
  const R* wv = wave + (long)b * S;
  const int F = N / 2 + 1;

  for (int j = tid; j < N; j += 256) {
    const double ang = 2.0 * (double)j / (double)N;
    cs[j] = (R)cospi(ang);
    sn[j] = (R)sinpi(ang);
  }
  if (tid == 0) {
    double s = 0.0;
    for (int j = 0; j < N; ++j) {
      const double w = 0.5 - 0.5 * cospi(2.0 * (double)j / (double)N);
      s += w * w;
    }
    wsum_s = (R)s;
  }
  __syncthreads();
  for (int e = tid; e < ST_FT * N; e += 256) {
    const int f = e / N, n = e - f * N;
    const int t = t0 + f;
    R v = 0;
    if (t < T) {
      int i = t * hop + n - N / 2;  // center=True, reflect padding
      if (i < 0) i = -i;
      if (i >= S) i = 2 * (S - 1) - i;
      const R w = (R)0.5 - (R)0.5 * cs[n];  // periodic Hann
      v = wv[i] * w;
    }
    fr[e] = v;
  }
  __syncthreads();
  const R inv = CPLX ? (R)1 / (R)sqrt((double)wsum_s) : (R)1 / wsum_s;
  for (int k = tid; k < F; k += 256) {
    R re[ST_FT], im[ST_FT];
#pragma unroll
    for (int f = 0; f < ST_FT; ++f) re[f] = im[f] = 0;
    int idx = 0;
    for (int n = 0; n < N; ++n) {
      const R c = cs[idx], s = sn[idx];
#pragma unroll
      for (int f = 0; f < ST_FT; ++f) {
        const R x = fr[f * N + n];
        re[f] += x * c;
        im[f] -= x * s;
      }
      idx += k;
      if (idx >= N) idx -= N;
    }
#pragma unroll
    for (int f = 0; f < ST_FT; ++f)
      if (t0 + f < T) {
        const long o = ((long)b * F + k) * T + t0 + f;
        if (CPLX) {
          power[2 * o] = re[f] * inv;
          power[2 * o + 1] = im[f] * inv;
        } else {
          power[o] = (re[f] * re[f] + im[f] * im[f]) * inv;
        }
      }
  }
}

// scipy.signal.convolve(wave, h, mode="same") for a float32 waveform and a float64 impulse response, in float64
// (genereate_dataset.py:38): out[i] = sum_j h[j] * wave[i + (Nh-1)/2 - j].  One workgroup = 1024 outputs, 4 consecutive
// ones per thread; the impulse response is walked in 256-tap chunks staged in LDS together with the waveform samples
// the chunk touches.  A thread keeps a sliding 4-sample window in registers, so each tap costs one broadcast read of
// h and ONE new waveform sample for its four multiply-adds.
constexpr int FIR_OUT = 1024, FIR_TAPS = 256;
__global__ __launch_bounds__(256) void fir_same_kernel(const float* wave, const double* h, double* out, int S, int Nh, int h_stride) {
  __shared__ double hs[FIR_TAPS];
  __shared__ double ws[FIR_OUT + FIR_TAPS];
  const int tid = threadIdx.x, b = blockIdx.y;
  const int i0 = blockIdx.x * FIR_OUT, off = (Nh - 1) / 2;
  const float* wv = wave + (long)b * S;
  const double* hv = h + (long)b * h_stride;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int j0 = 0; j0 < Nh; j0 += FIR_TAPS) {
    hs[tid] = (j0 + tid < Nh) ? hv[j0 + tid] : 0.0;
    // outputs i0 .. i0+1023 and taps j0 .. j0+255 read wave[base .. base + 1278], base = i0 + off - j0 - 255
    const int base = i0 + off - j0 - (FIR_TAPS - 1);
    for (int e = tid; e < FIR_OUT + FIR_TAPS - 1; e += 256) {
      const int idx = base + e;
      ws[e] = (idx >= 0 && idx < S) ? (double)wv[idx] : 0.0;
    }
    __syncthreads();
    // output i0 + 4*tid + k at tap j0 + jj reads ws[4*tid + k + 255 - jj]
    const double* wp = ws + 4 * tid + (FIR_TAPS - 1);
    double w0 = wp[0], w1 = wp[1], w2 = wp[2], w3 = wp[3];
#pragma unroll 8
    for (int jj = 0; jj < FIR_TAPS; ++jj) {
      const double hj = hs[jj];
      acc[0] += hj * w0;
      acc[1] += hj * w1;
      acc[2] += hj * w2;
      acc[3] += hj * w3;
      w3 = w2;
      w2 = w1;
      w1 = w0;
      w0 = (jj + 1 < FIR_TAPS) ? wp[-(jj + 1)] : 0.0;
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (i0 + 4 * tid + k < S) out[(long)b * S + i0 + 4 * tid + k] = acc[k];
}

// The spectrogram arithmetic of the dataset generator (genereate_dataset.py:41-49) on complex STFTs:
//   r = S / (E + 1e-8);  rir = |r / max|r||^2;  wiener[f] = |sum_t E conj(S) / (sum_t S conj(S) + 1e-8)|^2;
//   speech = |S|^2 (fp32, S is complex64);  echoed = |E|^2 (fp64, E is complex128).
// Pass 1 (one workgroup per frequency bin): wiener[f], the bin's max |r| and the two power spectrograms.
// Pass 2: the global max (fixed-order scan of the per-bin maxima) and rir.
struct cplx {
  double re, im;
};
__device__ __forceinline__ cplx ratio(float sr, float si, double er, double ei) {
  const double dr = er + 1e-8, d = dr * dr + ei * ei;       // S / (E + 1e-8)
  return cplx{((double)sr * dr + (double)si * ei) / d, ((double)si * dr - (double)sr * ei) / d};
}
__device__ __forceinline__ double block_sum(double v, double* red, int tid) {
  red[tid] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  const double r = red[0];
  __syncthreads();
  return r;
}
__global__ __launch_bounds__(256) void spec_stats_kernel(const float* S, const double* E, float* speech_pow, double* echoed_pow,
                                                         double* wiener, double* binmax, int F, int T) {
  __shared__ double red[256];
  const int tid = threadIdx.x, f = blockIdx.x, b = blockIdx.y;
  const long row = ((long)b * F + f) * T;
  double nr = 0, ni = 0, den = 0, mx = 0;
  for (int t = tid; t < T; t += 256) {
    const float sr = S[2 * (row + t)], si = S[2 * (row + t) + 1];
    const double er = E[2 * (row + t)], ei = E[2 * (row + t) + 1];
    nr += er * (double)sr + ei * (double)si;               // E * conj(S)
    ni += ei * (double)sr - er * (double)si;
    den += (double)sr * (double)sr + (double)si * (double)si;
    const cplx r = ratio(sr, si, er, ei);
    const double m = sqrt(r.re * r.re + r.im * r.im);
    mx = m > mx ? m : mx;
    speech_pow[row + t] = sr * sr + si * si;
    echoed_pow[row + t] = er * er + ei * ei;
  }
  nr = block_sum(nr, red, tid);
  ni = block_sum(ni, red, tid);
  den = block_sum(den, red, tid);
  red[tid] = mx;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] = red[tid] > red[tid + s] ? red[tid] : red[tid + s];
    __syncthreads();
  }
  if (tid == 0) {
    const double d = den + 1e-8, wr = nr / d, wi = ni / d;
    wiener[(long)b * F + f] = wr * wr + wi * wi;
    binmax[(long)b * F + f] = red[0];
  }
}
__global__ __launch_bounds__(256) void spec_rir_kernel(const float* S, const double* E, const double* binmax, double* rir_pow,
                                                       int F, int T) {
  __shared__ double gmax_s;
  const int tid = threadIdx.x, f = blockIdx.x, b = blockIdx.y;
  if (tid == 0) {
    double g = 0;
    for (int k = 0; k < F; ++k) g = binmax[(long)b * F + k] > g ? binmax[(long)b * F + k] : g;
    gmax_s = g;
  }
  __syncthreads();
  const double g = gmax_s;
  const long row = ((long)b * F + f) * T;
  for (int t = tid; t < T; t += 256) {
    const cplx r = ratio(S[2 * (row + t)], S[2 * (row + t) + 1], E[2 * (row + t)], E[2 * (row + t) + 1]);
    const double a = r.re / g, c = r.im / g;
    rir_pow[row + t] = a * a + c * c;
  }
}

}  // namespace alvq

using namespace alvq;

template <typename R, bool CPLX>
static int stft_launch(const R* wave, R* power, int B, int S, int n_fft, int hop, void* stream, const char* who) {
  ALVQ_REQUIRE(wave && power, ALVQ_EINVAL, "%s: null pointer", who);
  ALVQ_REQUIRE(B > 0 && hop > 0 && n_fft >= 4 && n_fft % 2 == 0, ALVQ_EINVAL, "%s: bad dims", who);
  ALVQ_REQUIRE(S > n_fft / 2, ALVQ_EINVAL, "%s: reflect padding needs S > n_fft/2 (S=%d)", who, S);
  ALVQ_REQUIRE(n_fft <= 2048 * (int)(sizeof(float)) / (int)sizeof(R) * 1, ALVQ_EUNSUPPORTED, "%s: n_fft=%d too large", who, n_fft);
  const int T = 1 + S / hop;
  const size_t lds = (size_t)(2 + ST_FT) * n_fft * sizeof(R);
  (void)hipFuncSetAttribute((const void*)stft_power_kernel<R, CPLX>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  hipLaunchKernelGGL((stft_power_kernel<R, CPLX>), dim3(B * ((T + ST_FT - 1) / ST_FT)), dim3(256), lds, (hipStream_t)stream, wave,
                     power, B, S, n_fft, hop, T);
  return check_launch(who);
}

extern "C" int alvq_stft_power_f32(const float* wave, float* power, int B, int S, int n_fft, int hop, void* stream) {
  return stft_launch<float, false>(wave, power, B, S, n_fft, hop, stream, "alvq_stft_power_f32");
}

extern "C" int alvq_stft_power_f64(const double* wave, double* power, int B, int S, int n_fft, int hop, void* stream) {
  return stft_launch<double, false>(wave, power, B, S, n_fft, hop, stream, "alvq_stft_power_f64");
}

extern "C" int alvq_stft_complex_f32(const float* wave, float* spec, int B, int S, int n_fft, int hop, void* stream) {
  return stft_launch<float, true>(wave, spec, B, S, n_fft, hop, stream, "alvq_stft_complex_f32");
}

extern "C" int alvq_stft_complex_f64(const double* wave, double* spec, int B, int S, int n_fft, int hop, void* stream) {
  return stft_launch<double, true>(wave, spec, B, S, n_fft, hop, stream, "alvq_stft_complex_f64");
}

extern "C" int alvq_fir_same_f64(const float* wave, const double* h, double* out, int B, int S, int Nh, int h_batch_stride,
                                 void* stream) {
  ALVQ_REQUIRE(wave && h && out, ALVQ_EINVAL, "alvq_fir_same_f64: null pointer");
  ALVQ_REQUIRE(B > 0 && S > 0 && Nh > 0 && Nh <= S, ALVQ_EINVAL, "alvq_fir_same_f64: bad dims (B=%d S=%d Nh=%d; Nh <= S)", B, S, Nh);
  ALVQ_REQUIRE(h_batch_stride == 0 || h_batch_stride >= Nh, ALVQ_EINVAL, "alvq_fir_same_f64: h_batch_stride");
  hipLaunchKernelGGL(fir_same_kernel, dim3((S + FIR_OUT - 1) / FIR_OUT, B), dim3(256), 0, (hipStream_t)stream, wave, h, out, S, Nh,
                     h_batch_stride);
  return check_launch("alvq_fir_same_f64");
}

extern "C" int alvq_spec_rir_wiener_f64(const float* speech_spec, const double* echoed_spec, float* speech_pow,
                                        double* echoed_pow, double* rir_pow, double* wiener, double* workspace, int B, int F,
                                        int T, void* stream) {
  ALVQ_REQUIRE(speech_spec && echoed_spec && speech_pow && echoed_pow && rir_pow && wiener && workspace, ALVQ_EINVAL,
               "alvq_spec_rir_wiener_f64: null pointer");
  ALVQ_REQUIRE(B > 0 && F > 0 && T > 0, ALVQ_EINVAL, "alvq_spec_rir_wiener_f64: bad dims");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(spec_stats_kernel, dim3(F, B), dim3(256), 0, s, speech_spec, echoed_spec, speech_pow, echoed_pow, wiener,
                     workspace, F, T);
  hipLaunchKernelGGL(spec_rir_kernel, dim3(F, B), dim3(256), 0, s, speech_spec, echoed_spec, (const double*)workspace, rir_pow,
                     F, T);
  return check_launch("alvq_spec_rir_wiener_f64");
}
