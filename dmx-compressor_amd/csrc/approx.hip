// csrc/approx.hip — approximator-slot ops (GELU / Softmax / LayerNorm) for gfx950.
//
// In the reference these modules compute the exact torch.nn.functional result first
// (functional/approximate.py:300-304) and then overwrite it with a vsimd approximation from a private
// package that is not part of the repository (approximate.py:9-14, 145-147).  With vsimd absent — the public
// state of the reference — every approximator is NONE and the exact function IS the result.  These kernels
// implement that exact-function contract in fp32 (one read, one write per element); approximation
// arithmetic is PARITY-UNPINNED and is not invented here.
//
//   gelu       elementwise, erf or tanh form (GeluOp in elementwise.hip)  (torch.nn.functional.gelu)
//   softmax    over the contiguous last dim, optional input clamp  (modeling/nn/torch_modules.py:989-994)
//   layernorm  over the contiguous last dim, affine optional       (modeling/nn/torch_modules.py:1062-1082)
// Row kernels: one workgroup per row, the row is staged ONCE in LDS as fp32 (gfx950 has 160 KiB per CU), the
// reductions are wave shuffles + a 4-entry LDS exchange, and the result is written from LDS: 1 read + 1 write
// of HBM per element.  Rows too long for LDS take a 3-pass global fallback.
#include <math.h>

#include "common.hpp"

namespace dmxq {

constexpr int kRowLdsFloats = 16 * 1024;  // 64 KiB per workgroup -> 2 workgroups per CU

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_maxf(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
// workgroup all-reduce through LDS scratch (kThreads/kWave floats)
template <bool IS_MAX>
__device__ __forceinline__ float block_allreduce(float v, float* scratch) {
  v = IS_MAX ? wave_maxf(v) : wave_sum(v);
  const int w = threadIdx.x / kWave;
  __syncthreads();  // scratch reuse
  if ((threadIdx.x & (kWave - 1)) == 0) scratch[w] = v;
  __syncthreads();
  float r = scratch[0];
#pragma unroll
  for (int i = 1; i < kThreads / kWave; i++) r = IS_MAX ? fmaxf(r, scratch[i]) : r + scratch[i];
  return r;
}

template <bool LDS_ROW>
__global__ __launch_bounds__(kThreads) void softmax_rows_kernel(const void* __restrict__ in, void* __restrict__ out,
                                                               int dti, int dto, int64_t rows, int64_t cols,
                                                               float clamp_min) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* scratch = smem;             // kThreads / kWave
  float* row = smem + kThreads / kWave;
  for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
    const int64_t base = r * cols;
    float m = -INFINITY;
    for (int64_t c = threadIdx.x; c < cols; c += kThreads) {
      float v = load_rt(in, dti, base + c);
      v = fmaxf(v, clamp_min);  // torch.clamp(x, min=input_clamp); clamp_min = -inf disables
      if (LDS_ROW) row[c] = v;
      m = fmaxf(m, v);
    }
    m = block_allreduce<true>(m, scratch);
    float s = 0.0f;
    for (int64_t c = threadIdx.x; c < cols; c += kThreads) {
      const float v = LDS_ROW ? row[c] : fmaxf(load_rt(in, dti, base + c), clamp_min);
      const float e = expf(v - m);
      if (LDS_ROW) row[c] = e;
      s += e;
    }
    s = block_allreduce<false>(s, scratch);
    for (int64_t c = threadIdx.x; c < cols; c += kThreads) {
      const float e = LDS_ROW ? row[c] : expf(fmaxf(load_rt(in, dti, base + c), clamp_min) - m);
      store_rt(out, dto, base + c, e / s);
    }
    __syncthreads();
  }
}

template <bool LDS_ROW>
__global__ __launch_bounds__(kThreads) void layernorm_rows_kernel(const void* __restrict__ in, void* __restrict__ out,
                                                                 int dti, int dto, int64_t rows, int64_t cols,
                                                                 const void* __restrict__ w,
                                                                 const void* __restrict__ b, int dtw, float eps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* scratch = smem;
  float* row = smem + kThreads / kWave;
  for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
    const int64_t base = r * cols;
    float s = 0.0f;
    for (int64_t c = threadIdx.x; c < cols; c += kThreads) {
      const float v = load_rt(in, dti, base + c);
      if (LDS_ROW) row[c] = v;
      s += v;
    }
    const float mean = block_allreduce<false>(s, scratch) / (float)cols;
    float q = 0.0f;
    for (int64_t c = threadIdx.x; c < cols; c += kThreads) {
      const float d = (LDS_ROW ? row[c] : load_rt(in, dti, base + c)) - mean;
      q += d * d;
    }
    const float var = block_allreduce<false>(q, scratch) / (float)cols;  // biased, as F.layer_norm
    const float rstd = 1.0f / sqrtf(var + eps);
    for (int64_t c = threadIdx.x; c < cols; c += kThreads) {
      float y = ((LDS_ROW ? row[c] : load_rt(in, dti, base + c)) - mean) * rstd;
      if (w) y *= load_rt(w, dtw, c);
      if (b) y += load_rt(b, dtw, c);
      store_rt(out, dto, base + c, y);
    }
    __syncthreads();
  }
}

}  // namespace dmxq

using namespace dmxq;

static inline int row_grid(int64_t rows) { return (int)(rows < 256 * 16 ? (rows < 1 ? 1 : rows) : 256 * 16); }

extern "C" int dmxq_softmax(const void* in, void* out, int dtype_in, int dtype_out, int64_t rows, int64_t cols,
                            float input_clamp_min, void* stream) {
  if (!valid_dtype(dtype_in) || !valid_dtype(dtype_out) || rows < 0 || cols < 0) return DMXQ_ERR_BAD_ARG;
  if (rows * cols == 0) return DMXQ_OK;
  if (!in || !out) return DMXQ_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  const size_t scratch = (kThreads / kWave) * sizeof(float);
  if (cols <= kRowLdsFloats)
    hipLaunchKernelGGL(softmax_rows_kernel<true>, dim3(row_grid(rows)), dim3(kThreads), scratch + cols * sizeof(float), s,
                       in, out, dtype_in, dtype_out, rows, cols, input_clamp_min);
  else
    hipLaunchKernelGGL(softmax_rows_kernel<false>, dim3(row_grid(rows)), dim3(kThreads), scratch, s, in, out, dtype_in,
                       dtype_out, rows, cols, input_clamp_min);
  return launch_status();
}

extern "C" int dmxq_layernorm(const void* in, void* out, int dtype_in, int dtype_out, int64_t rows, int64_t cols,
                              const void* weight, const void* bias, int dtype_wb, float eps, void* stream) {
  if (!valid_dtype(dtype_in) || !valid_dtype(dtype_out) || rows < 0 || cols < 0) return DMXQ_ERR_BAD_ARG;
  if ((weight || bias) && !valid_dtype(dtype_wb)) return DMXQ_ERR_BAD_ARG;
  if (rows * cols == 0) return DMXQ_OK;
  if (!in || !out) return DMXQ_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  const size_t scratch = (kThreads / kWave) * sizeof(float);
  if (cols <= kRowLdsFloats)
    hipLaunchKernelGGL(layernorm_rows_kernel<true>, dim3(row_grid(rows)), dim3(kThreads), scratch + cols * sizeof(float),
                       s, in, out, dtype_in, dtype_out, rows, cols, weight, bias, dtype_wb, eps);
  else
    hipLaunchKernelGGL(layernorm_rows_kernel<false>, dim3(row_grid(rows)), dim3(kThreads), scratch, s, in, out,
                       dtype_in, dtype_out, rows, cols, weight, bias, dtype_wb, eps);
  return launch_status();
}
