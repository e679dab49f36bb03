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
template <typename R>
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
  const R inv = (R)1 / wsum_s;
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
      if (t0 + f < T) power[((long)b * F + k) * T + t0 + f] = (re[f] * re[f] + im[f] * im[f]) * inv;
  }
}

}  // namespace alvq

using namespace alvq;

template <typename R>
static int stft_launch(const R* wave, R* power, int B, int S, int n_fft, int hop, void* stream, const char* who) {
  ALVQ_REQUIRE(wave && power, ALVQ_EINVAL, "%s: null pointer", who);
  ALVQ_REQUIRE(B > 0 && hop > 0 && n_fft >= 4 && n_fft % 2 == 0, ALVQ_EINVAL, "%s: bad dims", who);
  ALVQ_REQUIRE(S > n_fft / 2, ALVQ_EINVAL, "%s: reflect padding needs S > n_fft/2 (S=%d)", who, S);
  ALVQ_REQUIRE(n_fft <= 2048 * (int)(sizeof(float)) / (int)sizeof(R) * 1, ALVQ_EUNSUPPORTED, "%s: n_fft=%d too large", who, n_fft);
  const int T = 1 + S / hop;
  const size_t lds = (size_t)(2 + ST_FT) * n_fft * sizeof(R);
  (void)hipFuncSetAttribute((const void*)stft_power_kernel<R>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  hipLaunchKernelGGL((stft_power_kernel<R>), dim3(B * ((T + ST_FT - 1) / ST_FT)), dim3(256), lds, (hipStream_t)stream, wave, power,
                     B, S, n_fft, hop, T);
  return check_launch(who);
}

extern "C" int alvq_stft_power_f32(const float* wave, float* power, int B, int S, int n_fft, int hop, void* stream) {
  return stft_launch<float>(wave, power, B, S, n_fft, hop, stream, "alvq_stft_power_f32");
}

extern "C" int alvq_stft_power_f64(const double* wave, double* power, int B, int S, int n_fft, int hop, void* stream) {
  return stft_launch<double>(wave, power, B, S, n_fft, hop, stream, "alvq_stft_power_f64");
}
