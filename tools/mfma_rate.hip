// Probe (diagnostic tool, not product code): issue rate of the 32x32 MFMA forms of the f16mx kernels with the kernels'
// register geometry (8 accumulators of 32x32 per wave, tied inline asm), on random-looking operands.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_rate.hip -o acoustic_locating_vq-vae_amd/build/mfma_rate
// Prints, per mode / waves per SIMD / grid: shader cycles per MFMA per SIMD (s_memtime) and the clock (vs s_memrealtime).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v6i __attribute__((ext_vector_type(6)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned hsh(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// MODE 0: 16 fp16 MFMAs (two k-steps over 8 accumulators)   1: 8 block-scaled fp8 MFMAs   2: both (one K-tile of the kernel)
//      3: 8 block-scaled fp6 MFMAs   4: fp16 with FOUR distinct A/B fragment pairs (as in the kernel: 4 A x 2 B)
template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void rate_kernel(unsigned long long* stamps, float* out, int iters) {
  v8i qa[4], qb[2];
  v8h ha[4][2], hb[2][2];
  for (int f = 0; f < 4; ++f)
    for (int i = 0; i < 8; ++i) {
      qa[f][i] = (int)((hsh(threadIdx.x * 64 + f * 8 + i + blockIdx.x * 7919) & 0x87878787u) | 0x28282828u);
      if (f < 2) qb[f][i] = (int)((hsh(threadIdx.x * 64 + f * 8 + i + 77777) & 0x87878787u) | 0x30303030u);
      for (int k = 0; k < 2; ++k) {
        ha[f][k][i] = (_Float16)(((int)(hsh(threadIdx.x * 128 + f * 16 + k * 8 + i) & 0xffff) - 32768) / 32768.f);
        if (f < 2) hb[f][k][i] = (_Float16)(((int)(hsh(threadIdx.x * 128 + f * 16 + k * 8 + i + 4242) & 0xffff) - 32768) / 16384.f);
      }
    }
  int sa = 127, sb = 127;
  asm volatile("" : "+v"(sa), "+v"(sb));
  v16f c[4][2];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 2; ++j)
      for (int q = 0; q < 16; ++q) c[i][j][q] = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#define H(MI, NI, KS) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c[MI][NI]) : "v"(ha[MODE >= 4 ? MI : 0][KS]), "v"(hb[MODE >= 4 ? NI : 0][KS]));
#define Q(MI, NI) asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(c[MI][NI]) : "v"(qa[MI]), "v"(qb[NI]), "v"(sa), "v"(sb));
#define Q6(MI, NI) { const v6i a6 = __builtin_shufflevector(qa[MI], qa[MI], 0, 1, 2, 3, 4, 5), b6 = __builtin_shufflevector(qb[NI], qb[NI], 0, 1, 2, 3, 4, 5); \
    asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:2 blgp:2" : "+v"(c[MI][NI]) : "v"(a6), "v"(b6), "v"(sa), "v"(sb)); }
#define ALL(X) X(0, 0) X(0, 1) X(1, 0) X(1, 1) X(2, 0) X(2, 1) X(3, 0) X(3, 1)
#define SERP(X) X(0, 0) X(1, 0) X(2, 0) X(3, 0) X(3, 1) X(2, 1) X(1, 1) X(0, 1)   /* B-stationary serpentine */
#define H0(MI, NI) H(MI, NI, 0)
#define H1(MI, NI) H(MI, NI, 1)
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0 || MODE == 2 || MODE == 4) { ALL(H0) ALL(H1) }
    if (MODE == 1 || MODE == 2) { ALL(Q) }
    if (MODE == 3) { ALL(Q6) }
    if (MODE == 5) { SERP(H0) SERP(H1) }
    if (MODE == 6) { SERP(H0) SERP(H1) SERP(Q) }
    if (MODE == 7) { ALL(H0) ALL(H1) ALL(Q) }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 2; ++j) s += c[i][j][0] + c[i][j][7];
  if (s == 12345.678f) out[0] = s;
}

template <int MODE, int THREADS>
static void run(const char* name, int grid, int iters) {
  unsigned long long* st; float* out;
  CK(hipMalloc(&st, grid * 16)); CK(hipMalloc(&out, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((rate_kernel<MODE, THREADS>), dim3(grid), dim3(THREADS), 0, 0, st, out, 10);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((rate_kernel<MODE, THREADS>), dim3(grid), dim3(THREADS), 0, 0, st, out, iters);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long* h = (unsigned long long*)malloc(grid * 16);
  CK(hipMemcpy(h, st, grid * 16, hipMemcpyDeviceToHost));
  double cyc = 0, ref = 0;
  for (int i = 0; i < grid; ++i) { cyc += h[2 * i]; ref += h[2 * i + 1]; }
  cyc /= grid; ref /= grid;
  const int per_iter = (MODE == 2 || MODE == 6 || MODE == 7) ? 24 : (MODE == 1 || MODE == 3) ? 8 : 16;
  const int waves_per_simd = THREADS / 256;
  const double us = ref / 100.0;                      // s_memrealtime: 100 MHz
  const double mfma_per_simd = (double)iters * per_iter * waves_per_simd;
  // nominal pipe cycles: fp16 32, fp8 64, fp6 32
  const double nominal = (MODE == 2 || MODE == 6 || MODE == 7) ? (16 * 32 + 8 * 64) / 24.0 : (MODE == 1) ? 64 : 32;
  printf("%-34s grid %4d waves/SIMD %d: %7.1f us  s_memtime/MFMA %6.2f  ns/MFMA %6.2f  => %5.2f GHz-equivalent of nominal %4.1f cycles  (event %.3f ms)\n",
         name, grid, waves_per_simd, us, cyc / mfma_per_simd, us * 1000.0 / mfma_per_simd, nominal / (us * 1000.0 / mfma_per_simd), nominal, ms);
  hipFree(st); hipFree(out); free(h);
}

int main() {
  const int iters = 3000;
  for (int grid : {32, 256}) {
    run<0, 512>("fp16 32x32x16 (one fragment pair)", grid, iters);
    run<4, 512>("fp16 32x32x16 (4 A x 2 B fragments)", grid, iters);
    run<1, 512>("fp8 MX 32x32x64", grid, iters);
    run<3, 512>("fp6 MX 32x32x64", grid, iters);
    run<2, 512>("K-tile mix: 16 fp16 + 8 fp8", grid, iters);
    run<7, 512>("mix, 4A x 2B, row-major order", grid, iters);
    run<6, 512>("mix, 4A x 2B, B-stationary order", grid, iters);
    run<5, 512>("fp16, 4A x 2B, B-stationary order", grid, iters);
    run<7, 512>("mix, 4A x 2B, row-major order", grid, iters);
    run<6, 512>("mix, 4A x 2B, B-stationary order", grid, iters);
    run<0, 256>("fp16 32x32x16 (one fragment pair)", grid, iters);
    run<4, 256>("fp16 32x32x16 (4 A x 2 B fragments)", grid, iters);
    run<1, 256>("fp8 MX 32x32x64", grid, iters);
    run<2, 256>("K-tile mix: 16 fp16 + 8 fp8", grid, iters);
  }
  return 0;
}
