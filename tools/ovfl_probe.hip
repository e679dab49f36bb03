// Probe (diagnostic tool): do MODE.FP16_OVFL (hwreg MODE bit 23) make the f32 -> f16 / fp8
// conversions saturate instead of returning inf / NaN?
//   hipcc --offload-arch=gfx950 -O2 tools/ovfl_probe.hip -o acoustic_locating_vq-vae_amd/build/ovfl_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* x, unsigned* out, int ovfl) {
  if (ovfl) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);
  const float a = x[threadIdx.x * 2], b = x[threadIdx.x * 2 + 1];
  s16x2 r = {0, 0};
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, a, b, 1.0f, false);
  const int p = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  const f2 f = {a, b};
  const h2 hh = __builtin_convertvector(f, h2);
  unsigned c = 0;
  out[threadIdx.x * 4 + 0] = (unsigned)(unsigned short)r[0];
  out[threadIdx.x * 4 + 1] = (unsigned)p & 0xffff;
  out[threadIdx.x * 4 + 2] = __builtin_bit_cast(unsigned, hh);
  out[threadIdx.x * 4 + 3] = c & 0xffff;
  if (ovfl) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 0);
}
int main() {
  float hx[8] = {1000.f, -1000.f, 460.f, 470.f, 1e6f, -7e4f, 65520.f, 3.0f};
  float* dx; unsigned* dout; unsigned ho[16];
  hipMalloc(&dx, sizeof(hx)); hipMalloc(&dout, sizeof(ho));
  hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice);
  for (int ovfl = 0; ovfl < 2; ++ovfl) {
    hipLaunchKernelGGL(k, dim3(1), dim3(4), 0, 0, dx, dout, ovfl);
    hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
    for (int i = 0; i < 4; ++i)
      printf("FP16_OVFL=%d in (%g, %g): scalef32_pk_fp8 %04x  cvt_pk_fp8 %04x  f16 pair %08x  cvt_pk_fp8 clamp %04x\n", ovfl, hx[2 * i], hx[2 * i + 1],
             ho[4 * i], ho[4 * i + 1], ho[4 * i + 2], ho[4 * i + 3]);
  }
  return 0;
}
