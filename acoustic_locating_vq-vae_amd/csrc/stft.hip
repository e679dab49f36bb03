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

__global__ __launch_bounds__(256) void stft_power_kernel(const float* wave, float* power, int B, int S, int N, int hop,
                                                         int T) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* cs = sm;               // [N]
  float* sn = sm + N;           // [N]
  float* fr = sm + 2 * N;       // [ST_FT][N]
  __shared__ float wsum_s;
  const int tid = threadIdx.x;
  const int ttiles = (T + ST_FT - 1) / ST_FT;
  const int b = blockIdx.x / ttiles, t0 = (blockIdx.x % ttiles) * ST_FT;
  const float* wv = wave + (long)b * S;
  const int F = N / 2 + 1;

  for (int j = tid; j < N; j += 256) {
    const double ang = 2.0 * (double)j / (double)N;
    cs[j] = (float)cospi(ang);
    sn[j] = (float)sinpi(ang);
  }
  if (tid == 0) {
    double s = 0.0;
    for (int j = 0; j < N; ++j) {
      const double w = 0.5 - 0.5 * cospi(2.0 * (double)j / (double)N);
      s += w * w;
    }
    wsum_s = (float)s;
  }
  __syncthreads();
  for (int e = tid; e < ST_FT * N; e += 256) {
    const int f = e / N, n = e - f * N;
    const int t = t0 + f;
    float v = 0.f;
    if (t < T) {
      int i = t * hop + n - N / 2;  // center=True, reflect padding
      if (i < 0) i = -i;
      if (i >= S) i = 2 * (S - 1) - i;
      const float w = 0.5f - 0.5f * cs[n];  // periodic Hann
      v = wv[i] * w;
    }
    fr[e] = v;
  }
  __syncthreads();
  const float inv = 1.f / wsum_s;
  for (int k = tid; k < F; k += 256) {
    float re[ST_FT], im[ST_FT];
#pragma unroll
    for (int f = 0; f < ST_FT; ++f) re[f] = im[f] = 0.f;
    int idx = 0;
    for (int n = 0; n < N; ++n) {
      const float c = cs[idx], s = sn[idx];
#pragma unroll
      for (int f = 0; f < ST_FT; ++f) {
        const float x = fr[f * N + n];
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

extern "C" int alvq_stft_power_f32(const float* wave, float* power, int B, int S, int n_fft, int hop, void* stream) {
  ALVQ_REQUIRE(wave && power, ALVQ_EINVAL, "alvq_stft_power_f32: null pointer");
  ALVQ_REQUIRE(B > 0 && hop > 0 && n_fft >= 4 && n_fft % 2 == 0, ALVQ_EINVAL, "alvq_stft_power_f32: bad dims");
  ALVQ_REQUIRE(S > n_fft / 2, ALVQ_EINVAL, "alvq_stft_power_f32: reflect padding needs S > n_fft/2 (S=%d)", S);
  ALVQ_REQUIRE(n_fft <= 2048, ALVQ_EUNSUPPORTED, "alvq_stft_power_f32: n_fft=%d > 2048", n_fft);
  const int T = 1 + S / hop;
  const size_t lds = (size_t)(2 + ST_FT) * n_fft * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)stft_power_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(stft_power_kernel, dim3(B * ((T + ST_FT - 1) / ST_FT)), dim3(256), lds, (hipStream_t)stream, wave,
                     power, B, S, n_fft, hop, T);
  return check_launch("alvq_stft_power_f32");
}
