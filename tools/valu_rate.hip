// tools/valu_rate.hip -- how many cycles does a wave64 VALU instruction occupy its SIMD on gfx950?  (round 3: several kernels of this
// library are bound by VALU issue, not by HBM; this measures the rate the estimates in DESIGN.md use.)
// Register-only loops of independent FMA chains, every CU full (8 waves per SIMD), no memory traffic:
//   v_fma_f32 (scalar fp32), v_pk_fma_f32 (two fp32 per lane), v_exp_f32 (transcendental), v_pk_min_u16 (packed 16-bit).
// Output: wave-instructions per SIMD per microsecond and the implied cycles per instruction at the clock read from the device.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
constexpr int ITERS = 2048, CHAINS = 8;

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, float a, float b) {
  float r[CHAINS];
  f32x2 p[CHAINS];
  unsigned int w[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; c++) { r[c] = threadIdx.x * 1e-3f + c; p[c] = (f32x2){r[c], r[c] + 1.0f}; w[c] = threadIdx.x * 77u + c; }
  for (int i = 0; i < ITERS; i++) {
#pragma unroll
    for (int c = 0; c < CHAINS; c++) {
      if (KIND == 0) r[c] = __builtin_fmaf(r[c], a, b);
      else if (KIND == 1) p[c] = __builtin_elementwise_fma(p[c], (f32x2){a, a}, (f32x2){b, b});
      else if (KIND == 2) r[c] = __builtin_amdgcn_exp2f(r[c]);
      else w[c] = __builtin_bit_cast(unsigned int, __builtin_elementwise_min(__builtin_bit_cast(u16x2, w[c] + 1u), __builtin_bit_cast(u16x2, 0x7F007F00u)));
    }
  }
  float s = 0.0f;
#pragma unroll
  for (int c = 0; c < CHAINS; c++) s += r[c] + p[c].x + p[c].y + (float)w[c];
  if (s == 12345.678f) out[0] = s;
}

template <int KIND>
static void run(const char* name, int instr_per_iter_per_chain, double mhz, int cus) {
  float* out; CK(hipMalloc(&out, 4));
  const int blocks = cus * 8;  // 8 workgroups of 4 waves per CU = 8 waves per SIMD
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 1e-7f);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 1e-7f);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / 5;
  const double wave_instr_per_simd = 8.0 * ITERS * CHAINS * instr_per_iter_per_chain;  // 8 waves per SIMD
  printf("%-34s %9.1f us   %8.1f wave-instr / SIMD / us   %5.2f cycles per wave-instruction at %.0f MHz\n", name, us, wave_instr_per_simd / us,
         us * mhz / wave_instr_per_simd, mhz);
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const double mhz = p.clockRate / 1e3;
  printf("# %s, %d CUs, clockRate %.0f MHz\n", p.name, p.multiProcessorCount, mhz);
  run<0>("v_fma_f32", 1, mhz, p.multiProcessorCount);
  run<1>("v_pk_fma_f32 (2 fp32 per lane)", 1, mhz, p.multiProcessorCount);
  run<2>("v_exp_f32", 1, mhz, p.multiProcessorCount);
  run<3>("v_add_u32 + v_pk_min_u16", 2, mhz, p.multiProcessorCount);
  return 0;
}
