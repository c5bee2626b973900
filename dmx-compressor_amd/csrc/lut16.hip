// csrc/lut16.hip -- an activation-function DmxModule on a 16-bit tensor as a TABLE (round 4).
//
//     out = cast_out( f( cast_in(x) ) )       x, out: bf16 or fp16         (modeling/nn/core.py:228-264 around a per-element function)
//
// is a function of 16 bits: 65,536 inputs.  Two entry points:
//   dmxq_unary_cast_table  fills table[p] = the module's result for input PATTERN p, for ANY FloatingPoint casts (rounding ones too:
//                          the table is built with the bit-level cast of floatq.hpp), the function evaluated in FLOAT64 and rounded
//                          ONCE to the tensor dtype (round-to-odd to float32 first, so the float32 -> 16-bit step cannot round
//                          twice).  That is the correctly rounded f(cast_in(x)) -- what torch's CPU evaluation (the reference's
//                          device) returns for silu / exp and in all but double-rounding cases for gelu -- instead of "within one ulp".
//   dmxq_lut16_apply       out[i] = table[in[i]]: a workgroup of 1024 lanes copies the 128 KiB table into the LDS (CDNA4: 160 KiB per
//                          CU) behind its own input loads and then reads one 16-bit word per element; the arithmetic of the function
//                          (GELU: ~20 VALU-equivalents per element incl. two transcendentals, 56-58 % of the roofline) is gone.
// The table belongs to the CALLER (a module builds it once per (function, casts, dtype) and keeps it: dmx_compressor_amd/nn.py);
// the library stays stateless.  Worth it from a few MiB up (every workgroup pays 128 KiB of L2 -> LDS traffic): the host mirror
// takes this path for tensors >= 4 MiB and the direct kernels of act_cast.hip below.
#include <math.h>

#include "floatq.hpp"

namespace dmxq {

// double -> float32 with round-to-ODD (truncate toward zero, set the last bit when inexact): a following round-to-nearest-even to a
// format with at least two fewer mantissa bits then equals ONE rounding of the double (Boldo & Melquiond)
__device__ __forceinline__ float to_odd_f32(double d) {
  float f = (float)d;                       // RNE
  if ((double)f == d || d != d) return f;   // exact (incl. +-Inf from an Inf input), or NaN
  if (__builtin_isinf(f)) return f;         // overflow: stays Inf for the 16-bit step as well
  uint32_t b = f2u(f);
  if (fabs((double)f) > fabs(d)) b -= 1u;   // rounded away from zero: step back to the truncation
  return u2f(b | 1u);
}
template <int DT>
__device__ __forceinline__ float pattern_value(uint32_t p) { return DT == DMXQ_BF16 ? u2f(p << 16) : half_lo(p); }
template <int DT>
__device__ __forceinline__ uint32_t value_pattern(float v) {   // v is representable in DT
  if (DT == DMXQ_BF16) return f2u(v) >> 16;
  return (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)opaque(v));
}
__device__ __forceinline__ double sigmoid64(double t) { return 1.0 / (1.0 + exp(-t)); }

// f on a value of the tensor dtype, rounded once to the tensor dtype (QUICK_GELU: three roundings in the tensor dtype, like
// transformers' module and UnaryOp in unary_ops.hpp)
template <int DT>
__device__ __forceinline__ float eval_unary(int kind, float x, float param) {
  const double xd = (double)x;
  double y;
  switch (kind) {
    case DMXQ_UNARY_GELU: y = xd * (0.5 * erfc(-xd * 0.70710678118654752440)); break;            // x Phi(x), no cancellation in the tail
    case DMXQ_UNARY_GELU_TANH: {
      const double u = 0.79788456080286535588 * (xd + 0.044715 * xd * xd * xd);                 // 0.5 x (1 + tanh u) = x sigmoid(2u)
      y = xd * sigmoid64(2.0 * u);
      break;
    }
    case DMXQ_UNARY_SILU: y = xd * sigmoid64(xd); break;
    case DMXQ_UNARY_EXP: y = exp(xd); break;
    case DMXQ_UNARY_QUICK_GELU: {
      const float t = castg_dt<DT>(1.702f * x);
      const float s = castg_dt<DT>(to_odd_f32(sigmoid64((double)t)));
      return castg_dt<DT>(x * s);
    }
    default: {  // DMXQ_UNARY_SILU_EXPERIMENTAL: relu(half(x)) * scale (functional/functions.py:7-21), UnaryOp's arithmetic
      const float h = (float)(_Float16)opaque(x);
      const float r = h < 0.0f ? 0.0f : h;
      return castg_dt<DT>(r * param);
    }
  }
  // x = +-Inf: Inf * 0 = NaN where torch gives NaN too (gelu(-inf), silu(-inf)); the products above already do that
  return castg_dt<DT>(to_odd_f32(y));
}

template <int DT>
__global__ __launch_bounds__(256) void unary_table_kernel(int kind, float param, CastG gi, CastG go, uint16_t* __restrict__ table) {
  const uint32_t p = blockIdx.x * 256 + threadIdx.x;   // 65536 patterns
  float v = pattern_value<DT>(p);
  if (gi.active) v = castg_dt<DT>(float_q1<DMXQ_ROUND_NEAREST>(v, gi.f, 0u));   // the input CastTo: bit-level cast, then `.to(dtype)`
  v = eval_unary<DT>(kind, v, param);
  if (go.active) v = castg_dt<DT>(float_q1<DMXQ_ROUND_NEAREST>(v, go.f, 0u));
  table[p] = (uint16_t)value_pattern<DT>(v);
}

// out[i] = table[in[i]].  One workgroup per CU (128 KiB of LDS), tiles of 1024 x 8 lane-vectors: 4096 x 4096 is exactly one tile per
// workgroup.  Memory schedule of the first tile, written with explicit instructions because it rests on the ORDER of the loads
// (vector memory returns in order; left to the compiler, a table load sank behind the inputs and everything waited for HBM):
//   8 table loads (served by L2 after the first workgroup of an XCD), 8 input loads (HBM)  ->  s_waitcnt vmcnt(8): the table alone
//   -> copied to the LDS, barrier  ->  input vector u is looked up as soon as IT has arrived (vmcnt(7 - u))  ->  one store burst.
constexpr int kLutThreads = 1024;
__device__ __forceinline__ u32x4 asm_load16(const void* p) {
  u32x4 r;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(r) : "v"(p) : "memory");
  return r;
}
__device__ __forceinline__ u32x4 asm_load16_nt(const void* p) {
  u32x4 r;
  asm volatile("global_load_dwordx4 %0, %1, off nt" : "=&v"(r) : "v"(p) : "memory");
  return r;
}
// "the value in x is valid once at most N later loads are outstanding": the register is an in/out operand of the wait, so every use
// of it is ordered behind the wait
template <int N>
__device__ __forceinline__ void wait_vm(u32x4& x) {
  static_assert(N >= 0 && N <= 7, "vmcnt literal");
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" : "+v"(x)::"memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" : "+v"(x)::"memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" : "+v"(x)::"memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" : "+v"(x)::"memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" : "+v"(x)::"memory");
  else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" : "+v"(x)::"memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" : "+v"(x)::"memory");
  else asm volatile("s_waitcnt vmcnt(7)" : "+v"(x)::"memory");
}
template <int U>
__device__ __forceinline__ void wait_table(u32x4 (&t)[8]) {   // the 8 table vectors: at most the U input loads still outstanding
#define DMXQ_T8 "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7])
  if constexpr (U == 1) asm volatile("s_waitcnt vmcnt(1)" : DMXQ_T8::"memory");
  else if constexpr (U == 2) asm volatile("s_waitcnt vmcnt(2)" : DMXQ_T8::"memory");
  else if constexpr (U == 4) asm volatile("s_waitcnt vmcnt(4)" : DMXQ_T8::"memory");
  else asm volatile("s_waitcnt vmcnt(8)" : DMXQ_T8::"memory");
#undef DMXQ_T8
}
__device__ __forceinline__ u32x4 lut_lookup(const uint16_t* s_lut, const u32x4& w) {
  u32x4 o;
#pragma unroll
  for (int j = 0; j < 4; j++) o[j] = (uint32_t)s_lut[w[j] & 0xFFFFu] | ((uint32_t)s_lut[w[j] >> 16] << 16);
  return o;
}
template <int U, int I>
__device__ __forceinline__ void lookup_in_order(const uint16_t* s_lut, u32x4 (&raw)[U]) {
  if constexpr (I < U) {
    wait_vm<U - 1 - I>(raw[I]);
    raw[I] = lut_lookup(s_lut, raw[I]);
    lookup_in_order<U, I + 1>(s_lut, raw);
  }
}
// U = vectors per lane of a tile, chosen by the host so that a small tensor still spreads over the CUs (each workgroup copies the whole
// table: with 8 vectors per lane a 3.7 MB activation ran on 28 workgroups, 7.8 us; 1 vector per lane: 224 workgroups)
template <int U>
__global__ __launch_bounds__(kLutThreads) void lut16_apply_kernel(const void* __restrict__ in, void* __restrict__ out, int64_t n_vec,
                                                                 const uint16_t* __restrict__ table) {
  extern __shared__ uint16_t s_lut[];   // 65536 entries
  constexpr int T = kLutThreads;
  const int64_t tile = (int64_t)T * U;
  int64_t base = (int64_t)blockIdx.x * tile;
  {
    u32x4 t[8], raw[U];
#pragma unroll
    for (int k = 0; k < 8; k++) t[k] = asm_load16((const u32x4*)table + k * T + threadIdx.x);   // 8192 vectors of 16 bytes
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t v = base + (int64_t)u * T + threadIdx.x;
      raw[u] = asm_load16_nt((const u32x4*)in + (v < n_vec ? v : n_vec - 1));
    }
    wait_table<U>(t);
#pragma unroll
    for (int k = 0; k < 8; k++) *((u32x4*)s_lut + k * T + threadIdx.x) = t[k];
    __syncthreads();
    lookup_in_order<U, 0>(s_lut, raw);
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t v = base + (int64_t)u * T + threadIdx.x;
      if (v < n_vec) __builtin_nontemporal_store(raw[u], (u32x4*)out + v);
    }
  }
  // further tiles of a tensor beyond 256 tiles (> 32 MiB at U = 8): the table is in place
  for (base += (int64_t)gridDim.x * tile; base < n_vec; base += (int64_t)gridDim.x * tile) {
    u32x4 raw[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t v = base + (int64_t)u * T + threadIdx.x;
      raw[u] = __builtin_nontemporal_load((const u32x4*)in + (v < n_vec ? v : n_vec - 1));
    }
#pragma unroll
    for (int u = 0; u < U; u++) raw[u] = lut_lookup(s_lut, raw[u]);
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t v = base + (int64_t)u * T + threadIdx.x;
      if (v < n_vec) __builtin_nontemporal_store(raw[u], (u32x4*)out + v);
    }
  }
}

}  // namespace dmxq

using namespace dmxq;

extern "C" int dmxq_unary_cast_table(int dtype, int kind, float param, const dmxq_float_fmt* cast_in, const dmxq_float_fmt* cast_out,
                                     void* table, void* stream) {
  if ((dtype != DMXQ_BF16 && dtype != DMXQ_F16) || kind < DMXQ_UNARY_GELU || kind > DMXQ_UNARY_SILU_EXPERIMENTAL || !table)
    return DMXQ_ERR_BAD_ARG;
  CastG gi, go;
  if (!castg_of(cast_in, &gi) || !castg_of(cast_out, &go)) return DMXQ_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == DMXQ_BF16) DMXQ_LAUNCH(unary_table_kernel<DMXQ_BF16>, dim3(256), dim3(256), 0, s, kind, param, gi, go, (uint16_t*)table);
  else DMXQ_LAUNCH(unary_table_kernel<DMXQ_F16>, dim3(256), dim3(256), 0, s, kind, param, gi, go, (uint16_t*)table);
  return launch_status();
}

extern "C" int dmxq_lut16_apply(const void* in, void* out, int64_t n, const void* table, void* stream) {
  if (n < 0) return DMXQ_ERR_BAD_ARG;
  if (n == 0) return DMXQ_OK;
  if (!in || !out || !table) return DMXQ_ERR_BAD_ARG;
  if (n % 8 != 0 || !aligned16(in) || !aligned16(out) || !aligned16(table)) return DMXQ_ERR_UNSUPPORTED;
  // 128 KiB of dynamic LDS is above a kernel's default limit: raised once per device (an idempotent call: a race between two
  // threads' first calls sets the same value twice)
  static bool lds_set[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return DMXQ_ERR_LAUNCH;
  if (!lds_set[dev]) {
    const void* ks[4] = {(const void*)lut16_apply_kernel<1>, (const void*)lut16_apply_kernel<2>, (const void*)lut16_apply_kernel<4>,
                         (const void*)lut16_apply_kernel<8>};
    for (const void* k : ks)
      if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 * 2) != hipSuccess) {
        (void)hipGetLastError();
        return DMXQ_ERR_UNSUPPORTED;
      }
    lds_set[dev] = true;
  }
  // vectors per lane: as many as still give every CU a workgroup (one resident workgroup per CU: 128 KiB of LDS); beyond 256 tiles of
  // 1024 x 8 the workgroups loop (the table copy is paid once per workgroup)
  const int64_t n_vec = n / 8;
  const int u = n_vec >= (int64_t)256 * kLutThreads * 8 ? 8 : (n_vec >= (int64_t)256 * kLutThreads * 4 ? 4 : (n_vec >= (int64_t)256 * kLutThreads * 2 ? 2 : 1));
  const int64_t tile = (int64_t)kLutThreads * u;
  int64_t grid = (n_vec + tile - 1) / tile;
  if (grid > 256) grid = 256;
  hipStream_t s = (hipStream_t)stream;
#define DMXQ_LUT(U_) DMXQ_LAUNCH(lut16_apply_kernel<U_>, dim3((unsigned)grid), dim3(kLutThreads), 65536 * 2, s, in, out, n_vec, (const uint16_t*)table)
  switch (u) { case 1: DMXQ_LUT(1); break; case 2: DMXQ_LUT(2); break; case 4: DMXQ_LUT(4); break; default: DMXQ_LUT(8); }
#undef DMXQ_LUT
  return launch_status();
}
