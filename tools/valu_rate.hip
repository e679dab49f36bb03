// Probe (diagnostic tool): issue cost of the VALU instructions the f16mx epilogue is made of (cycles per wave-instruction
// per SIMD, two waves per SIMD, dependent chains of 8 independent registers).
//   hipcc --offload-arch=gfx950 -O2 tools/valu_rate.hip -o acoustic_locating_vq-vae_amd/build/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

#define OPS8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters, float seed) {
  float r[8]; unsigned u[8];
  for (int i = 0; i < 8; ++i) { r[i] = seed + threadIdx.x * 0.001f + i; u[i] = 0x3c003c00u + threadIdx.x + i; }
  float s = 1.0f;
  asm volatile("" : "+v"(s));
  unsigned long long msk = 0x5555555555555555ull + (unsigned long long)iters;
  asm volatile("" : "+s"(msk));
  double d[8];
  for (int i = 0; i < 8; ++i) d[i] = __longlong_as_double(0x3f8000003f800000ll + i);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#define A0(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(s));
#define A1(i) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(r[i]) : "v"(u[i]));
#define A2(i) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(r[i]) : "v"(u[i]));
#define A3(i) asm volatile("v_cvt_f32_fp8_sdwa %0, %1 src0_sel:BYTE_1" : "=v"(r[i]) : "v"(u[i]));
#define A4(i) asm volatile("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %3" : "+v"(u[i]) : "v"(r[i]), "v"(r[(i + 1) & 7]), "v"(s));
#define A5(i) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(r[i]), "v"(r[(i + 1) & 7]));
#define A6(i) asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r[i]) : "v"(u[i]), "v"(s), "v"(r[i]));
#define A7(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(u[i]), "+v"(u[(i + 4) & 7]));
#define A8(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(s) : );
#define A9(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(s), "v"(r[(i + 1) & 7]));
#define A10(i) asm volatile("v_cvt_scalef32_pk_fp8_f16 %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(s));
    if (MODE == 0) { OPS8(A0) OPS8(A0) }
    if (MODE == 1) { OPS8(A1) OPS8(A1) }
    if (MODE == 2) { OPS8(A2) OPS8(A2) }
    if (MODE == 3) { OPS8(A3) OPS8(A3) }
    if (MODE == 4) { OPS8(A4) OPS8(A4) }
    if (MODE == 5) { OPS8(A5) OPS8(A5) }
    if (MODE == 6) { OPS8(A6) OPS8(A6) }
    if (MODE == 7) { OPS8(A7) OPS8(A7) }
    if (MODE == 8) { OPS8(A8) OPS8(A8) }
    if (MODE == 9) { OPS8(A9) OPS8(A9) }
    if (MODE == 10) { OPS8(A10) OPS8(A10) }
#define A11(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r[i]) : "v"(s), "s"(msk));
#define A12(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
#define A13(i) asm volatile("v_bfe_i32 %0, %1, 3, 1" : "=v"(u[i]) : "v"(u[(i + 1) & 7]));
#define A14(i) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
#define A15(i) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(s) : "vcc");
#define A16(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
#define A17(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(r[i]) : "v"(s));
    if (MODE == 11) { OPS8(A11) OPS8(A11) }
    if (MODE == 12) { OPS8(A12) OPS8(A12) }
    if (MODE == 13) { OPS8(A13) OPS8(A13) }
    if (MODE == 14) { OPS8(A14) OPS8(A14) }
    if (MODE == 15) { OPS8(A15) OPS8(A15) }
    if (MODE == 16) { OPS8(A16) OPS8(A16) }
    if (MODE == 17) { OPS8(A17) OPS8(A17) }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float acc = 0;
  for (int i = 0; i < 8; ++i) acc += r[i] + __uint_as_float(u[i]) + (float)d[i];
  if (acc == 12345.678f) out[0] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = (float)(t1 - t0);
}

template <int MODE>
static void run(const char* name) {
  float* out; CK(hipMalloc(&out, 64));
  const int iters = 2000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<MODE>, dim3(32), dim3(512), 0, 0, out, 10, 1.0f);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<MODE>, dim3(32), dim3(512), 0, 0, out, iters, 1.0f);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  // 2 waves per SIMD, 16 instructions per iteration each
  const double ns_per_inst = ms * 1e6 / (2.0 * iters * 16);
  printf("%-34s %6.2f ns per wave-instruction per SIMD = %5.1f cycles at 2.4 GHz\n", name, ns_per_inst, ns_per_inst * 2.4);
  hipFree(out);
}

int main() {
  run<0>("v_add_f32");
  run<1>("v_cvt_f32_f16");
  run<2>("v_cvt_f32_f16_sdwa (WORD_1)");
  run<3>("v_cvt_f32_fp8_sdwa");
  run<4>("v_cvt_scalef32_pk_fp8_f32");
  run<5>("v_cvt_pk_f16_f32");
  run<6>("v_fma_mix_f32");
  run<7>("v_permlane32_swap_b32");
  run<8>("v_cndmask_b32");
  run<9>("v_med3_f32");
  run<10>("v_cvt_scalef32_pk_fp8_f16");
  run<11>("v_cndmask_b32_e64 (sgpr pair)");
  run<12>("v_and_b32");
  run<13>("v_bfe_i32");
  run<14>("v_alignbit_b32");
  run<15>("v_cmp_gt_f32 + v_cndmask (pair)");
  run<16>("v_pk_add_f32");
  run<17>("v_max_f32");
  return 0;
}
