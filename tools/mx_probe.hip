// Probe (diagnostic tool, not product code): operand / scale layout of v_mfma_scale_f32_16x16x128_f8f6f4 with fp8
// e4m3 operands, the f16 16x16x32 MFMA, the packed fp8 conversions and ds_read_b64_tr_b8 on gfx950.
//   hipcc --offload-arch=gfx950 -O2 tools/mx_probe.hip -o gpurun_out/mx_probe && gpurun_out/mx_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// ------------------------------------------------------------------ fp8 e4m3fn on the host
static float e4m3_to_f(uint8_t v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float r;
  if (e == 0) r = ldexpf((float)m, -9);
  else if (e == 15 && m == 7) r = NAN;
  else r = ldexpf(1.0f + m / 8.0f, e - 7);
  return s ? -r : r;
}
static uint8_t f_to_e4m3(float f) {   // RNE, saturating
  if (f != f) return 0x7f;
  const uint8_t s = f < 0 ? 0x80 : 0;
  float a = fabsf(f);
  if (a >= 464.f) return s | 0x7e;     // 448 = max; halfway to the next (480) rounds to even = 448... saturate
  uint8_t best = 0;
  float bd = 1e30f;
  for (int c = 0; c < 0x7f; ++c) {
    const float d = fabsf(e4m3_to_f((uint8_t)c) - a);
    if (d < bd || (d == bd && (c & 1) == 0)) { bd = d; best = (uint8_t)c; }
  }
  return s | best;
}

// ------------------------------------------------------------------ kernels
__global__ void mx_kernel(const uint8_t* A, const uint8_t* B, const int* sa, const int* sb, float* D, int opsel) {
  const int l = threadIdx.x;
  v8i a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = ((const int*)(A + l * 32))[i];
    b[i] = ((const int*)(B + l * 32))[i];
  }
  v4f c = {0.f, 0.f, 0.f, 0.f};
  if (opsel == 0) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sa[l], 0, sb[l]);
  else if (opsel == 1) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 1, sa[l], 1, sb[l]);
  else if (opsel == 2) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 2, sa[l], 2, sb[l]);
  else c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 3, sa[l], 3, sb[l]);
  for (int i = 0; i < 4; ++i) D[l * 4 + i] = c[i];
}

__global__ void f16_kernel(const _Float16* A, const _Float16* B, float* D) {
  const int l = threadIdx.x;
  v8h a, b;
  for (int i = 0; i < 8; ++i) { a[i] = A[l * 8 + i]; b[i] = B[l * 8 + i]; }
  v4f c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) D[l * 4 + i] = c[i];
}

__global__ void cvt_kernel(const float* x, int n, uint32_t* out_pk, uint32_t* out_sc, float scale) {
  const int i = threadIdx.x;
  if (2 * i + 1 < n) {
    out_pk[i] = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(x[2 * i], x[2 * i + 1], 0, false);
    typedef short v2s __attribute__((ext_vector_type(2)));
    v2s old = {0, 0};
    v2s r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(old, x[2 * i], x[2 * i + 1], scale, false);
    out_sc[i] = (uint32_t)(uint16_t)r[0] | ((uint32_t)(uint16_t)r[1] << 16);
  }
}

__global__ void tr8_kernel(uint32_t* out, int row_stride) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[8192];
  const int l = threadIdx.x;
  for (int i = l; i < 8192; i += 64) lds[i] = (uint8_t)(((i / row_stride) << 4) | ((i % row_stride) & 15));   // hi nibble = row, lo = col
  __syncthreads();
  const int g = l >> 4, i = l & 15;
  // hypothesis: lane 2q+p of a 16-lane group supplies row q (0..7), bytes 8p..8p+7 of a 16-byte-wide block
  const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) uint8_t*)lds + (i >> 1) * row_stride + (i & 1) * 8 + g * 16;
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[l * 2] = (uint32_t)v;
  out[l * 2 + 1] = (uint32_t)(v >> 32);
}

// MFMA issue-rate microbenchmark on hashed (random-looking) operands: 256 workgroups x 8 waves (two waves per SIMD,
// the geometry of the convolution kernels), 8 accumulators per wave.  Wave 0 of every workgroup stamps the shader
// clock (s_memtime) and the 100 MHz reference (s_memrealtime) around its loop.
typedef __bf16 v8b __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned hsh(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
template <int MODE>
__global__ __launch_bounds__(512, 2) void rate_kernel(unsigned long long* stamps, float* out, int iters) {
  v8i a, b;
  for (int i = 0; i < 8; ++i) {   // fp8 bytes with exponents 5..10: finite, mixed signs
    unsigned ra = hsh(threadIdx.x * 16 + i + blockIdx.x * 7919), rb = hsh(threadIdx.x * 16 + i + 77777);
    a[i] = (int)((ra & 0x87878787u) | 0x28282828u);
    b[i] = (int)((rb & 0x87878787u) | 0x30303030u);
  }
  v8h ha, hb; v8b ba, bb;
  for (int i = 0; i < 8; ++i) {
    const float fa = ((int)(hsh(threadIdx.x * 8 + i) & 0xffff) - 32768) / 32768.f, fb = ((int)(hsh(threadIdx.x * 8 + i + 4242) & 0xffff) - 32768) / 16384.f;
    ha[i] = (_Float16)fa; hb[i] = (_Float16)fb; ba[i] = (__bf16)fa; bb[i] = (__bf16)fb;
  }
  v4f c[8];
  for (int j = 0; j < 8; ++j) c[j] = v4f{0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (MODE == 0) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c[j], 0, 0, 0);
      else if (MODE == 1) c[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c[j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      else if (MODE == 2) {   // the planned mix per 64 channels: 2 f16 MFMAs + 1 MX MFMA
        c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c[j], 0, 0, 0);
        c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hb, ha, c[j], 0, 0, 0);
        c[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c[j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      } else if (MODE == 3) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, c[j], 0, 0, 0);
      else {                  // today's split-bf16 mix per 32 channels: 3 bf16 MFMAs
        c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, c[j], 0, 0, 0);
        c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bb, ba, c[j], 0, 0, 0);
        c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, ba, c[j], 0, 0, 0);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
  float s = 0.f;
  for (int j = 0; j < 8; ++j) s += c[j][0] + c[j][1] + c[j][2] + c[j][3];
  if (s == 12345.678f) out[0] = s;
}

static void run_mx(const uint8_t* A, const uint8_t* B, const int* sa, const int* sb, float* D, int opsel) {
  uint8_t *dA, *dB; int *dsa, *dsb; float* dD;
  CK(hipMalloc(&dA, 2048)); CK(hipMalloc(&dB, 2048)); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256)); CK(hipMalloc(&dD, 1024));
  CK(hipMemcpy(dA, A, 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B, 2048, hipMemcpyHostToDevice));
  CK(hipMemcpy(dsa, sa, 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, sb, 256, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(mx_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD, opsel);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(D, dD, 1024, hipMemcpyDeviceToHost));
  hipFree(dA); hipFree(dB); hipFree(dsa); hipFree(dsb); hipFree(dD);
}

int main() {
  srand(7);
  // ---------------------------------------------------------------- 1. f16 MFMA, bf16-style lane map hypothesis
  {
    _Float16 A[64 * 8], B[64 * 8];
    float Am[16][32], Bm[32][16];
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 8; ++j) {
        const float va = (float)((rand() % 17) - 8) / 8.f, vb = (float)((rand() % 17) - 8) / 4.f;
        A[l * 8 + j] = (_Float16)va; B[l * 8 + j] = (_Float16)vb;
        Am[l & 15][8 * (l >> 4) + j] = va; Bm[8 * (l >> 4) + j][l & 15] = vb;
      }
    _Float16 *dA, *dB; float* dD; float D[256];
    CK(hipMalloc(&dA, sizeof(A))); CK(hipMalloc(&dB, sizeof(B))); CK(hipMalloc(&dD, 1024));
    CK(hipMemcpy(dA, A, sizeof(A), hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B, sizeof(B), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(f16_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    CK(hipDeviceSynchronize()); CK(hipMemcpy(D, dD, 1024, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 4; ++r) {
        const int row = (l >> 4) * 4 + r, col = l & 15;
        float ref = 0; for (int k = 0; k < 32; ++k) ref += Am[row][k] * Bm[k][col];
        if (fabsf(ref - D[l * 4 + r]) > 1e-4f) ++bad;
      }
    printf("[f16 16x16x32] lane map A[l&15][8(l>>4)+j], B[8(l>>4)+j][l&15], D[(l>>4)*4+r][l&15]: %s (%d bad)\n", bad ? "MISMATCH" : "OK", bad);
  }
  // ---------------------------------------------------------------- 2. MX fp8: random data, hypotheses on the K map
  // H2 (from the one-hot diagnostics): byte j of lane group g sits at k = 64*(j>>4) + 16*g + (j&15); the scale of
  // (row, k-block kb) is byte `opsel` of the scale register of lane row + 16*kb.
  for (int hyp = 1; hyp <= 2; ++hyp)
  for (int opsel = 0; opsel < 4; ++opsel) {
    uint8_t A[2048], B[2048]; int sa[64], sb[64];
    float Am[16][128], Bm[128][16], SA[16][4], SB[16][4];
    for (int l = 0; l < 64; ++l) {
      int ea[4], eb[4];
      for (int q = 0; q < 4; ++q) { ea[q] = 120 + rand() % 12; eb[q] = 122 + rand() % 10; }
      sa[l] = ea[0] | (ea[1] << 8) | (ea[2] << 16) | (ea[3] << 24);
      sb[l] = eb[0] | (eb[1] << 8) | (eb[2] << 16) | (eb[3] << 24);
      SA[l & 15][l >> 4] = ldexpf(1.f, ea[opsel] - 127);
      SB[l & 15][l >> 4] = ldexpf(1.f, eb[opsel] - 127);
      for (int j = 0; j < 32; ++j) {
        const uint8_t ca = (uint8_t)((rand() % 2 ? 0x80 : 0) | (0x28 + rand() % 0x18)), cb = (uint8_t)((rand() % 2 ? 0x80 : 0) | (0x30 + rand() % 0x10));
        A[l * 32 + j] = ca; B[l * 32 + j] = cb;
        const int g = l >> 4;
        const int k = hyp == 1 ? 32 * g + j : 64 * (j >> 4) + 16 * g + (j & 15);
        Am[l & 15][k] = e4m3_to_f(ca);
        Bm[k][l & 15] = e4m3_to_f(cb);
      }
    }
    float D[256];
    run_mx(A, B, sa, sb, D, opsel);
    int bad = 0; double maxrel = 0;
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 4; ++r) {
        const int row = (l >> 4) * 4 + r, col = l & 15;
        double ref = 0;
        for (int k = 0; k < 128; ++k) ref += (double)Am[row][k] * SA[row][k >> 5] * Bm[k][col] * SB[col][k >> 5];
        const double rel = fabs(ref - D[l * 4 + r]) / (fabs(ref) + 1e-3);
        if (rel > maxrel) maxrel = rel;
        if (rel > 1e-4) ++bad;
      }
    printf("[mx fp8 16x16x128 H%d opsel=%d] %s (%d bad, max rel %.2e)\n", hyp, opsel, bad ? "MISMATCH" : "OK", bad, maxrel);
  }
  // ---------------------------------------------------------------- 2b. diagnostics: one-hot A, structured B
  {
    printf("[mx diag] A one-hot (value 1.0) at (lane la, byte ja), B = 1.0 everywhere except byte 5 of every lane = 2.0; sb(lane) = 2^(lane>>4), sa = 1\n");
    printf("          prints: la ja -> nonzero D rows, value at col 0 (expect 2^kblock, doubled if within-block index == 5)\n");
    for (int la = 0; la < 64; la += 21)
      for (int ja = 0; ja < 32; ja += 15) {
        uint8_t A[2048], B[2048]; int sa[64], sb[64];
        memset(A, 0, sizeof(A));
        for (int i = 0; i < 2048; ++i) B[i] = ((i & 31) == 5) ? 0x40 : 0x38;
        A[la * 32 + ja] = 0x38;
        for (int l = 0; l < 64; ++l) { sa[l] = 0x7f7f7f7f; sb[l] = (127 + (l >> 4)) * 0x01010101; }
        float D[256];
        run_mx(A, B, sa, sb, D, 0);
        printf("   la=%2d ja=%2d:", la, ja);
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (D[l * 4 + r] != 0.f && (l & 15) == 0) printf(" D[row %d]=%g", (l >> 4) * 4 + r, D[l * 4 + r]);
        printf("\n");
      }
  }
  // ---------------------------------------------------------------- 3. fp8 conversion
  {
    const float xs[] = {0.f, 1.f, -1.f, 1.0625f, 1.1875f, 448.f, 464.f, 500.f, 1e6f, 0.001953125f, 0.0009765625f, 0.0029296875f, 0.015625f, 17.f, 18.f, 19.f, -0.3f, 3.3e-4f};
    const int n = sizeof(xs) / sizeof(float);
    float* dx; uint32_t *dp, *ds; uint32_t hp[32], hs[32];
    CK(hipMalloc(&dx, sizeof(xs))); CK(hipMalloc(&dp, 128)); CK(hipMalloc(&ds, 128));
    CK(hipMemcpy(dx, xs, sizeof(xs), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(cvt_kernel, dim3(1), dim3(32), 0, 0, dx, n, dp, ds, 4.0f);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hp, dp, 128, hipMemcpyDeviceToHost)); CK(hipMemcpy(hs, ds, 128, hipMemcpyDeviceToHost));
    printf("[cvt] x -> cvt_pk_fp8_f32 byte (value) | host RNE-sat byte | cvt_scalef32_pk_fp8_f32(scale=4.0) byte (value)\n");
    for (int i = 0; i < n; ++i) {
      const uint8_t b = (uint8_t)(hp[i / 2] >> (8 * (i & 1))), h = f_to_e4m3(xs[i]), s = (uint8_t)(hs[i / 2] >> (8 * (i & 1)));
      printf("   %12g -> 0x%02x (%g) | 0x%02x (%g) | 0x%02x (%g)\n", xs[i], b, e4m3_to_f(b), h, e4m3_to_f(h), s, e4m3_to_f(s));
    }
  }
  // ---------------------------------------------------------------- 4. ds_read_b64_tr_b8
  for (int stride = 64; stride <= 64; stride += 64) {
    uint32_t* d; uint32_t h[128];
    CK(hipMalloc(&d, 512));
    hipLaunchKernelGGL(tr8_kernel, dim3(1), dim3(64), 0, 0, d, stride);
    CK(hipDeviceSynchronize()); CK(hipMemcpy(h, d, 512, hipMemcpyDeviceToHost));
    printf("[tr_b8 stride %d] LDS byte = (row<<4)|col; lane 2q+p of each 16-group gives row q, cols 8p..; per lane the 8 result bytes:\n", stride);
    for (int l = 0; l < 64; l += 13) {
      printf("   lane %2d:", l);
      for (int k = 0; k < 8; ++k) printf(" %02x", (h[l * 2 + (k >> 2)] >> (8 * (k & 3))) & 0xff);
      printf("\n");
    }
  }
  // ---------------------------------------------------------------- 5. issue rates, cycles and clock
  {
    float* d; CK(hipMalloc(&d, 16));
    unsigned long long* st; CK(hipMalloc(&st, 256 * 16));
    const int iters = 40000;
    const char* names[5] = {"f16 16x16x32", "mx fp8 16x16x128", "2 f16 + 1 mx (64 ch)", "bf16 16x16x32", "3 bf16 (32 ch)"};
    for (int mode = 0; mode < 5; ++mode) {
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        if (mode == 0) hipLaunchKernelGGL((rate_kernel<0>), dim3(256), dim3(512), 0, 0, st, d, iters);
        else if (mode == 1) hipLaunchKernelGGL((rate_kernel<1>), dim3(256), dim3(512), 0, 0, st, d, iters);
        else if (mode == 2) hipLaunchKernelGGL((rate_kernel<2>), dim3(256), dim3(512), 0, 0, st, d, iters);
        else if (mode == 3) hipLaunchKernelGGL((rate_kernel<3>), dim3(256), dim3(512), 0, 0, st, d, iters);
        else hipLaunchKernelGGL((rate_kernel<4>), dim3(256), dim3(512), 0, 0, st, d, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      }
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned long long h[512]; CK(hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost));
      double cyc = 0, ref = 0; for (int i = 0; i < 256; ++i) { cyc += h[2 * i]; ref += h[2 * i + 1]; }
      cyc /= 256; ref /= 256;
      const double per = (mode == 2 || mode == 4) ? 3.0 : 1.0;
      const double groups = (double)iters * 8;            // per wave
      printf("[rate %-22s] %.3f ms; per wave %.0f cycles per group of %d MFMA(s) (two waves share a SIMD -> %.1f cycles of pipe per group); clock %.2f GHz\n",
             names[mode], ms, cyc / groups, (int)per, cyc / groups / 2, cyc / ref * 0.1);
    }
  }
  return 0;
}
