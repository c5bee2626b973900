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

// ---------------------------------------------------------------------------------------------------------
// Wave-per-row versions for rows of up to 64 * 4 * VPL elements with cols % 4 == 0: the row lives in REGISTERS
// (VPL 4-element vectors per lane, 8-byte accesses for 16-bit dtypes, 16-byte for fp32, a wave reads 512 B / 1 KiB
// contiguous per step), reductions are wave shuffles only, no LDS and no workgroup barrier; 4 rows per workgroup.
__device__ __forceinline__ void load4_rt(const void* p, int dt, int64_t e, float (&v)[4]) {
  if (dt == DMXQ_F32) {
    const f32x4 t = *(const f32x4*)((const float*)p + e);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
    const u32x2 t = *(const u32x2*)((const uint16_t*)p + e);
    if (dt == DMXQ_BF16) {
      v[0] = u2f(t.x << 16); v[1] = u2f(t.x & 0xFFFF0000u); v[2] = u2f(t.y << 16); v[3] = u2f(t.y & 0xFFFF0000u);
    } else {
      v[0] = half_lo(t.x); v[1] = half_hi(t.x); v[2] = half_lo(t.y); v[3] = half_hi(t.y);
    }
  }
}
__device__ __forceinline__ void store4_rt(void* p, int dt, int64_t e, const float (&v)[4]) {
  if (dt == DMXQ_F32) {
    *(f32x4*)((float*)p + e) = f32x4{v[0], v[1], v[2], v[3]};
  } else if (dt == DMXQ_BF16) {
    *(u32x2*)((uint16_t*)p + e) = u32x2{pack2<DMXQ_BF16>(v[0], v[1]), pack2<DMXQ_BF16>(v[2], v[3])};
  } else {
    *(u32x2*)((uint16_t*)p + e) = u32x2{pack2<DMXQ_F16>(v[0], v[1]), pack2<DMXQ_F16>(v[2], v[3])};
  }
}

template <int VPL>
__global__ __launch_bounds__(kThreads) void softmax_wave_kernel(const void* __restrict__ in, void* __restrict__ out,
                                                               int dti, int dto, int64_t rows, int64_t cols,
                                                               float clamp_min) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t wave = (int64_t)blockIdx.x * (kThreads / kWave) + threadIdx.x / kWave;
  const int64_t n_waves = (int64_t)gridDim.x * (kThreads / kWave);
  for (int64_t r = wave; r < rows; r += n_waves) {
    const int64_t base = r * cols;
    float x[VPL][4];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < VPL; i++) {
      const int64_t c = ((int64_t)i * kWave + lane) * 4;
      if (c < cols) {
        load4_rt(in, dti, base + c, x[i]);
#pragma unroll
        for (int k = 0; k < 4; k++) { x[i][k] = fmaxf(x[i][k], clamp_min); m = fmaxf(m, x[i][k]); }
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++) x[i][k] = -INFINITY;
      }
    }
    m = wave_maxf(m);
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < VPL; i++)
#pragma unroll
      for (int k = 0; k < 4; k++) { x[i][k] = expf(x[i][k] - m); s += x[i][k]; }  // exp(-inf) = 0 for padding
    s = wave_sum(s);
#pragma unroll
    for (int i = 0; i < VPL; i++) {
      const int64_t c = ((int64_t)i * kWave + lane) * 4;
      if (c < cols) {
        float y[4];
#pragma unroll
        for (int k = 0; k < 4; k++) y[k] = x[i][k] / s;
        store4_rt(out, dto, base + c, y);
      }
    }
  }
}

template <int VPL>
__global__ __launch_bounds__(kThreads) void layernorm_wave_kernel(const void* __restrict__ in, void* __restrict__ out,
                                                                 int dti, int dto, int64_t rows, int64_t cols,
                                                                 const void* __restrict__ w,
                                                                 const void* __restrict__ b, int dtw, float eps) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t wave = (int64_t)blockIdx.x * (kThreads / kWave) + threadIdx.x / kWave;
  const int64_t n_waves = (int64_t)gridDim.x * (kThreads / kWave);
  for (int64_t r = wave; r < rows; r += n_waves) {
    const int64_t base = r * cols;
    float x[VPL][4];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < VPL; i++) {
      const int64_t c = ((int64_t)i * kWave + lane) * 4;
      if (c < cols) {
        load4_rt(in, dti, base + c, x[i]);
        s += (x[i][0] + x[i][1]) + (x[i][2] + x[i][3]);
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++) x[i][k] = 0.0f;
      }
    }
    const float mean = wave_sum(s) / (float)cols;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < VPL; i++) {
      const int64_t c = ((int64_t)i * kWave + lane) * 4;
      if (c < cols) {
#pragma unroll
        for (int k = 0; k < 4; k++) { const float d = x[i][k] - mean; q += d * d; }
      }
    }
    const float var = wave_sum(q) / (float)cols;
    const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int i = 0; i < VPL; i++) {
      const int64_t c = ((int64_t)i * kWave + lane) * 4;
      if (c < cols) {
        float y[4], ww[4], bb[4];
        if (w) load4_rt(w, dtw, c, ww);
        if (b) load4_rt(b, dtw, c, bb);
#pragma unroll
        for (int k = 0; k < 4; k++) {
          y[k] = (x[i][k] - mean) * rstd;
          if (w) y[k] *= ww[k];
          if (b) y[k] += bb[k];
        }
        store4_rt(out, dto, base + c, y);
      }
    }
  }
}

// vector width 4 usable: cols % 4 == 0 and every base pointer aligned to 4 elements of its dtype
static inline bool vec4_ok(const void* p, int dt, int64_t cols) {
  const uintptr_t a = dt == DMXQ_F32 ? 16 : 8;
  return cols % 4 == 0 && (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0;
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
  if (cols <= 64 * 4 * 16 && vec4_ok(in, dtype_in, cols) && vec4_ok(out, dtype_out, cols)) {
    const int64_t vpl = (cols / 4 + kWave - 1) / kWave;
    const int grid = (int)((rows + 3) / 4 < 256 * 32 ? (rows + 3) / 4 : 256 * 32);
#define DMXQ_SM(V_) hipLaunchKernelGGL(softmax_wave_kernel<V_>, dim3(grid), dim3(kThreads), 0, s, in, out, dtype_in, dtype_out, rows, cols, input_clamp_min)
    if (vpl <= 1) DMXQ_SM(1); else if (vpl <= 2) DMXQ_SM(2); else if (vpl <= 4) DMXQ_SM(4); else if (vpl <= 8) DMXQ_SM(8); else DMXQ_SM(16);
#undef DMXQ_SM
    return launch_status();
  }
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
  if (cols <= 64 * 4 * 16 && vec4_ok(in, dtype_in, cols) && vec4_ok(out, dtype_out, cols) &&
      (!weight || vec4_ok(weight, dtype_wb, cols)) && (!bias || vec4_ok(bias, dtype_wb, cols))) {
    const int64_t vpl = (cols / 4 + kWave - 1) / kWave;
    const int grid = (int)((rows + 3) / 4 < 256 * 32 ? (rows + 3) / 4 : 256 * 32);
#define DMXQ_LN(V_) hipLaunchKernelGGL(layernorm_wave_kernel<V_>, dim3(grid), dim3(kThreads), 0, s, in, out, dtype_in, dtype_out, rows, cols, weight, bias, dtype_wb, eps)
    if (vpl <= 1) DMXQ_LN(1); else if (vpl <= 2) DMXQ_LN(2); else if (vpl <= 4) DMXQ_LN(4); else if (vpl <= 8) DMXQ_LN(8); else DMXQ_LN(16);
#undef DMXQ_LN
    return launch_status();
  }
  const size_t scratch = (kThreads / kWave) * sizeof(float);
  if (cols <= kRowLdsFloats)
    hipLaunchKernelGGL(layernorm_rows_kernel<true>, dim3(row_grid(rows)), dim3(kThreads), scratch + cols * sizeof(float),
                       s, in, out, dtype_in, dtype_out, rows, cols, weight, bias, dtype_wb, eps);
  else
    hipLaunchKernelGGL(layernorm_rows_kernel<false>, dim3(row_grid(rows)), dim3(kThreads), scratch, s, in, out,
                       dtype_in, dtype_out, rows, cols, weight, bias, dtype_wb, eps);
  return launch_status();
}
