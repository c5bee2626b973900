// csrc/elementwise.hip — per-element fused quantize->dequantize kernels for gfx950:
//   * low-bit floating point  (numerical/format.py:208-233 -> quant_cpu.cpp:359-402, bit_helper.cpp:4-22)
//   * fixed point + affine     (numerical/format.py:134-142 -> quant_cpu.cpp:127-209, sim_helper.cpp:5-38;
//                               affine wrapper numerical/cast.py:278-296)
//   * per-channel scaling      (numerical/smoothquant.py:255-283)
// All share one skeleton: each lane moves 16 B of input per step (global_load_dwordx4), UNROLL steps in flight,
// fp32 arithmetic, one RNE narrowing to the output dtype, 16-byte stores.  HBM-bound: 2+2 B/elem for 16-bit I/O.
#include "common.hpp"

namespace dmxq {

// ------------------------------------------------------------------------------------------------- float
struct FloatFmt {
  int man, exp_bits, bias, flush, unsigned_abs, rounding;
  uint64_t seed;
};

// quant_cpu.cpp:359-402 for one element (oracle/oracle.c float_q1)
template <int RND>
__device__ __forceinline__ float float_q1(float a, const FloatFmt& f, uint32_t rnd) {
  const uint32_t target = f2u(a);
  const int target_exp = (int)((target & 0x7FFFFFFFu) >> 23) - 127;
  const int min_exp = -(f.bias - 1);
  float q;
  if (target_exp < min_exp) {
    if (f.flush) {
      q = 0.0f;
    } else {
      // subnormal of the simulated format: add +-2^min_exp so the kept mantissa bits line up with the
      // subnormal quantum, round, subtract (never saturates)
      const float shift = u2f(((uint32_t)(127 + min_exp) << 23) | (target & 0x80000000u));
      const float val = a + shift;
      q = u2f(round_bitwise<RND>(f2u(val), f.man, f.rounding, rnd)) - shift;
    }
  } else {
    uint32_t qb = round_bitwise<RND>(target, f.man, f.rounding, rnd);
    // bit_helper.cpp:4-22 clip_exponent: saturate (with the INPUT's sign) at 2^(2^(e-1)) * (2 - 2^-m);
    // no inf/nan codes are reserved
    const int max_e = (1 << (f.exp_bits - 1)) + 127;
    if (qb != 0u && (int)((qb & 0x7FFFFFFFu) >> 23) > max_e) {
      const uint32_t max_man = (0x007FFFFFu >> (23 - f.man)) << (23 - f.man);
      qb = (target & 0x80000000u) | ((uint32_t)max_e << 23) | max_man;
    }
    q = u2f(qb);
  }
  return f.unsigned_abs ? fabsf(q) : q;
}

// ------------------------------------------------------------------------------------------------- fixed
struct FixedFmt {
  int sigma, clamp, rounding;
  float t_min, t_max;
  uint64_t seed;
};

// sim_helper.cpp:14-21 round(a, 0.5, sigma): ldexp, (float)(a + 0.5f) - 0.5 in double, nearbyint (half-even),
// narrow to float, ldexp.  The double step is kept literally (v_add_f64 + v_rndne_f64): it is what makes
// e.g. 0.5 + 2^-24 round to 0.  sim_helper.cpp:24-38 for up (ceil) / down (floor).
__device__ __forceinline__ float fixed_q1(float a, const FixedFmt& f, float r) {
  a = ldexpf(a, -f.sigma);
  if (f.rounding == DMXQ_ROUND_UP) a = ceilf(a);
  else if (f.rounding == DMXQ_ROUND_DOWN) a = floorf(a);
  else a = (float)__builtin_rint((double)(a + r) - 0.5);
  a = ldexpf(a, f.sigma);
  if (f.clamp) a = a > f.t_max ? f.t_max : (a < f.t_min ? f.t_min : a);
  return a;
}

__device__ __forceinline__ float rnd_unit(uint64_t seed, uint64_t idx) {
  return (float)(rnd_bits(seed, idx) >> 8) * (1.0f / 16777216.0f);
}

// ------------------------------------------------------------------------------------------------- skeleton
// OP::apply(x, flat element index) -> y.   n_vec 16-byte input vectors + scalar tail handled by the caller.
template <int DTI, int DTO, int UNROLL, class OP>
__global__ __launch_bounds__(kThreads) void ew_vec_kernel(const void* __restrict__ in, void* __restrict__ out,
                                                         int64_t n, OP op) {
  constexpr int EPL = 16 / Elem<DTI>::bytes;
  const int64_t n_vec = n / EPL;
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  const int64_t tid = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  int64_t v = tid;
  for (; v + (UNROLL - 1) * stride < n_vec; v += UNROLL * stride) {
    float x[UNROLL][EPL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) load_vec<DTI, EPL>(in, (v + u * stride) * EPL, x[u]);
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      float y[EPL];
      const int64_t e0 = (v + u * stride) * EPL;
#pragma unroll
      for (int k = 0; k < EPL; k++) y[k] = op.apply(x[u][k], e0 + k);
      store_vec<DTO, EPL>(out, e0, y);
    }
  }
  for (; v < n_vec; v += stride) {
    float x[EPL], y[EPL];
    load_vec<DTI, EPL>(in, v * EPL, x);
#pragma unroll
    for (int k = 0; k < EPL; k++) y[k] = op.apply(x[k], v * EPL + k);
    store_vec<DTO, EPL>(out, v * EPL, y);
  }
  // scalar tail (n % EPL elements)
  const int64_t e = n_vec * EPL + tid;
  if (e < n) store1<DTO>(out, e, op.apply(load1<DTI>(in, e), e));
}

// unaligned pointers: scalar accesses
template <int DTI, int DTO, class OP>
__global__ __launch_bounds__(kThreads) void ew_scalar_kernel(const void* __restrict__ in, void* __restrict__ out,
                                                            int64_t n, OP op) {
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < n; e += stride)
    store1<DTO>(out, e, op.apply(load1<DTI>(in, e), e));
}

template <int RND>
struct FloatOp {
  FloatFmt f;
  __device__ __forceinline__ float apply(float x, int64_t e) const {
    const bool stoch = (RND == kRuntimeRounding) && f.rounding == DMXQ_ROUND_STOCHASTIC;
    return float_q1<RND>(x, f, stoch ? rnd_bits(f.seed, (uint64_t)e) : 0u);
  }
};

// channel lookup for [outer, C, inner]: c = (e / inner) % C ; group = c / group_size
struct ChannelMap {
  int64_t C, inner, group_size;
  __device__ __forceinline__ int64_t group(int64_t e) const { return ((e / inner) % C) / group_size; }
};

template <bool AFFINE>
struct FixedOp {
  FixedFmt f;
  ChannelMap cm;
  const float* scale;
  const int64_t* zp;
  __device__ __forceinline__ float apply(float x, int64_t e) const {
    float sc = 1.0f, z = 0.0f;
    if (AFFINE) {
      const int64_t g = cm.group(e);
      sc = scale[g];
      z = (float)zp[g];
      x = x / sc + z;  // IEEE division, as torch CPU (cast.py:293)
    }
    const float r = (f.rounding == DMXQ_ROUND_STOCHASTIC) ? rnd_unit(f.seed, (uint64_t)e) : 0.5f;
    float q = fixed_q1(x, f, r);
    if (AFFINE) q = (q - z) * sc;
    return q;
  }
};

template <bool DIVIDE>
struct ScaleOp {
  ChannelMap cm;
  const float* scale;
  __device__ __forceinline__ float apply(float x, int64_t e) const {
    const float s = scale[cm.group(e)];
    return DIVIDE ? x / s : x * s;
  }
};

template <int DTI, int DTO, class OP>
static int launch_ew(const void* in, void* out, int64_t n, const OP& op, hipStream_t s) {
  constexpr int EPL = 16 / Elem<DTI>::bytes;
  constexpr int UNROLL = 4;
  if (aligned16(in) && aligned16(out)) {
    const int grid = grid_for((n / EPL + UNROLL - 1) / UNROLL + 1);
    hipLaunchKernelGGL((ew_vec_kernel<DTI, DTO, UNROLL, OP>), dim3(grid), dim3(kThreads), 0, s, in, out, n, op);
  } else {
    hipLaunchKernelGGL((ew_scalar_kernel<DTI, DTO, OP>), dim3(grid_for(n)), dim3(kThreads), 0, s, in, out, n, op);
  }
  return launch_status();
}

template <class OP>
static int dispatch_dtypes(const void* in, void* out, int dti, int dto, int64_t n, const OP& op, hipStream_t s) {
#define DMXQ_DT(I_, O_) \
  if (dti == I_ && dto == O_) return launch_ew<I_, O_, OP>(in, out, n, op, s);
  DMXQ_DT(DMXQ_BF16, DMXQ_BF16)
  DMXQ_DT(DMXQ_F16, DMXQ_F16)
  DMXQ_DT(DMXQ_F32, DMXQ_F32)
  DMXQ_DT(DMXQ_BF16, DMXQ_F32)
  DMXQ_DT(DMXQ_F16, DMXQ_F32)
  DMXQ_DT(DMXQ_F32, DMXQ_BF16)
  DMXQ_DT(DMXQ_F32, DMXQ_F16)
#undef DMXQ_DT
  return DMXQ_ERR_BAD_ARG;
}

}  // namespace dmxq

using namespace dmxq;

extern "C" int dmxq_float_qdq(const void* in, void* out, int dtype_in, int dtype_out, int64_t n, int man_bits,
                              int exp_bits, int exp_bias, int flush_subnormal, int unsigned_abs, int rounding,
                              uint64_t seed, void* stream) {
  if (!valid_dtype(dtype_in) || !valid_dtype(dtype_out) || !valid_rounding(rounding) || n < 0) return DMXQ_ERR_BAD_ARG;
  if (exp_bits < 1 || exp_bits > 8 || man_bits < 0) return DMXQ_ERR_BAD_ARG;
  if (man_bits > 22) return DMXQ_ERR_UNSUPPORTED;  // quant_cpu.cpp:211-237 shifts by 23 - man_bits - 1 = -1: UB
  if (n == 0) return DMXQ_OK;
  if (!in || !out) return DMXQ_ERR_BAD_ARG;
  const FloatFmt f{man_bits, exp_bits, exp_bias, flush_subnormal ? 1 : 0, unsigned_abs ? 1 : 0, rounding, seed};
  hipStream_t s = (hipStream_t)stream;
  if (rounding == DMXQ_ROUND_NEAREST) return dispatch_dtypes(in, out, dtype_in, dtype_out, n, FloatOp<DMXQ_ROUND_NEAREST>{f}, s);
  return dispatch_dtypes(in, out, dtype_in, dtype_out, n, FloatOp<kRuntimeRounding>{f}, s);
}

extern "C" int dmxq_fixed_qdq(const void* in, void* out, int dtype_in, int dtype_out, int64_t outer, int64_t C,
                              int64_t inner, int precision, int fraction, int clamp, int symmetric, int rounding,
                              const float* scale, const int64_t* zero_point, int64_t group_size, uint64_t seed,
                              void* stream) {
  if (!valid_dtype(dtype_in) || !valid_dtype(dtype_out) || !valid_rounding(rounding)) return DMXQ_ERR_BAD_ARG;
  if (outer < 0 || C < 0 || inner < 0 || precision < 1 || group_size < 1) return DMXQ_ERR_BAD_ARG;
  if ((scale == nullptr) != (zero_point == nullptr)) return DMXQ_ERR_BAD_ARG;
  const int64_t n = outer * C * inner;
  if (n == 0) return DMXQ_OK;
  if (!in || !out) return DMXQ_ERR_BAD_ARG;
  // sim_helper.cpp:5-12 fixed_min_max, evaluated on the host in the same float/double mix
  const int sigma = -fraction;
  float t_min = (float)(-ldexp(1.0, precision - fraction - 1));
  const float t_max = (float)(-(double)t_min - ldexp(1.0, sigma));
  if (symmetric) t_min = (float)((double)t_min + ldexp(1.0, sigma));
  const FixedFmt f{sigma, clamp ? 1 : 0, rounding, t_min, t_max, seed};
  const ChannelMap cm{C, inner, group_size};
  hipStream_t s = (hipStream_t)stream;
  if (scale) return dispatch_dtypes(in, out, dtype_in, dtype_out, n, FixedOp<true>{f, cm, scale, zero_point}, s);
  return dispatch_dtypes(in, out, dtype_in, dtype_out, n, FixedOp<false>{f, cm, nullptr, nullptr}, s);
}

extern "C" int dmxq_scale_channels(const void* in, void* out, int dtype_in, int dtype_out, int64_t outer, int64_t C,
                                   int64_t inner, const float* scale, int divide, void* stream) {
  if (!valid_dtype(dtype_in) || !valid_dtype(dtype_out) || outer < 0 || C < 0 || inner < 0) return DMXQ_ERR_BAD_ARG;
  const int64_t n = outer * C * inner;
  if (n == 0) return DMXQ_OK;
  if (!in || !out || !scale) return DMXQ_ERR_BAD_ARG;
  const ChannelMap cm{C, inner, 1};
  hipStream_t s = (hipStream_t)stream;
  if (divide) return dispatch_dtypes(in, out, dtype_in, dtype_out, n, ScaleOp<true>{cm, scale}, s);
  return dispatch_dtypes(in, out, dtype_in, dtype_out, n, ScaleOp<false>{cm, scale}, s);
}

extern "C" const char* dmxq_status_string(int status) {
  switch (status) {
    case DMXQ_OK: return "ok";
    case DMXQ_ERR_BAD_ARG: return "bad argument";
    case DMXQ_ERR_UNSUPPORTED: return "unsupported parameter (undefined behaviour in the reference)";
    case DMXQ_ERR_LAUNCH: return "HIP kernel launch failed";
  }
  return "unknown status";
}

extern "C" int dmxq_abi_version(void) { return 1; }
