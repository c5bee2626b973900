// csrc/unary_ops.hpp -- the per-element functions of the approximator slot as OPs of the streaming skeleton (stream.hpp):
// GELU (erf / tanh), SILU, QUICK_GELU, EXP and the reference's `experimental.silu`.  Shared by elementwise.hip (dmxq_gelu),
// unary.hip (dmxq_unary) and act_cast.hip (dmxq_unary_cast: the same functions between a module's two casts).
#pragma once
#include <math.h>

#include "common.hpp"

namespace dmxq {

// torch.nn.functional.gelu, erf and tanh forms (approximator slot, see approx.hip).
// FAST (16-bit outputs only): the libm calls are replaced by short closed forms whose error (< 8e-7 absolute on the
// result for |x| <= 10, exact saturation beyond) is far inside the output format's half-ulp:
//   erf form : with z = |x| / sqrt 2, erfc(z) = t (a1 + t (a2 + t (a3 + t (a4 + t a5)))) exp(-z^2), t = 1 / (1 + p z)
//              (Abramowitz-Stegun 7.1.26, |error| < 1.5e-7), gelu = x >= 0 ? x - x erfc / 2 : x erfc / 2 -- the erfc form
//              keeps RELATIVE accuracy in the negative tail, where 1 + erf cancels;
//   tanh form: x (1 + tanh u) / 2 = x / (1 + exp(-2u)).
// TANH is a template parameter, not a kernel argument: as a run-time flag it was a scalar branch around EVERY element (16 per
// pair of vectors), which kept the compiler from interleaving the elements' rcp / exp chains
template <bool FAST, bool TANH>
struct GeluOp {
  static constexpr bool kHeavy = true;
  // tile geometry (stream.hpp), measured on 4096 x 4096 in round 3 after the arithmetic moved to the packed pipe: bf16 256 x 8 13.2 us vs
  // 256 x 2 13.9; the float32 module form (act_cast.hip, 4 elements per vector) 256 x 2 25.3 us vs 256 x 8 27.5
  static constexpr int kTileUnroll = FAST ? 8 : 2;
  static constexpr int kTileUnrollF32 = 2;
  // the fused module on 16-bit tensors (act_cast.hip: two packed range casts more per vector): 512 x 4 10.5 / 12.8 / 13.7 us on
  // 3072 / 3584 / 4096 x 4096 bf16, 256 x 8 12.0 / 14.4 / 15.2
  static constexpr int kCastUnroll = 4, kCastThreads = 512;
  __device__ __forceinline__ void apply_one(float x, float& y, int64_t) const {
    if (TANH) {
      const float k0 = 0.7978845608028654f, k1 = 0.044715f;
      const float u = k0 * (x + k1 * x * x * x);
      if (FAST) {  // x (1 + tanh u) / 2 = x / (1 + exp(-2u)): no cancellation in the negative tail
        y = x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u * -2.8853900817779268f));
      } else {
        y = 0.5f * x * (1.0f + tanhf(u));
      }
    } else if (FAST) {
      const float z = fabsf(x) * 0.7071067811865476f;
      const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
      float p = __builtin_fmaf(t, 1.061405429f, -1.453152027f);
      p = __builtin_fmaf(t, p, 1.421413741f);
      p = __builtin_fmaf(t, p, -0.284496736f);
      p = __builtin_fmaf(t, p, 0.254829592f);
      const float erfc = p * t * __builtin_amdgcn_exp2f(z * z * -1.4426950408889634f);
      const float g = 0.5f * x * erfc;
      y = x > 0.0f ? x - g : g;  // (x = -0: g = -0, as torch)
      if (!(x < INFINITY)) y = x;  // NaN, and +inf (inf - inf 0 otherwise); -inf keeps g = -inf 0 = NaN, as torch's 0.5 x (1 + erf)
    } else {
      y = 0.5f * x * (1.0f + erff(x * 0.7071067811865476f));
    }
  }
  // FAST, two elements per instruction through the packed fp32 pipe (v_pk_mul / v_pk_fma / v_pk_add_f32: these kernels are bound by
  // VALU issue, ~4 cycles per wave instruction, not by HBM -- profiles/r03_pmc_second_tier.txt): 11.5 instead of ~22 VALU per element.
  //   erf form : gelu = x Phi(x), Phi = x > 0 ? 1 - h : h with h = erfc(|x| / sqrt 2) / 2 -- the select sits on Phi, so +inf gives
  //              inf * 1, -inf gives -inf * 0 = NaN (as torch's 0.5 x (1 + erf)), -0 gives -0 * 0.5 and NaN propagates, with no
  //              special cases; relative accuracy in the negative tail as before (h is formed without cancellation);
  //   tanh form: x / (1 + exp(-2u)), u = k0 x (1 + k1 x^2).
  __device__ __forceinline__ f32x2 apply_pair(f32x2 x) const {
    if (TANH) {
      const f32x2 x2 = x * x;
      const f32x2 w = __builtin_elementwise_fma(x2, (f32x2){0.044715f, 0.044715f}, (f32x2){1.0f, 1.0f}) * x;   // x (1 + k1 x^2)
      const f32x2 a = w * (0.7978845608028654f * -2.8853900817779268f);                                          // -2 u log2 e
      f32x2 d = {__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
      d = d + 1.0f;
      const f32x2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
      return x * r;
    }
    const f32x2 ax = {__builtin_fabsf(x.x), __builtin_fabsf(x.y)};
    const f32x2 z = ax * 0.7071067811865476f;
    const f32x2 den = __builtin_elementwise_fma(z, (f32x2){0.3275911f, 0.3275911f}, (f32x2){1.0f, 1.0f});
    const f32x2 t = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
    f32x2 p = __builtin_elementwise_fma(t, (f32x2){1.061405429f, 1.061405429f}, (f32x2){-1.453152027f, -1.453152027f});
    p = __builtin_elementwise_fma(t, p, (f32x2){1.421413741f, 1.421413741f});
    p = __builtin_elementwise_fma(t, p, (f32x2){-0.284496736f, -0.284496736f});
    p = __builtin_elementwise_fma(t, p, (f32x2){0.254829592f, 0.254829592f});
    const f32x2 a = (z * z) * -1.4426950408889634f;
    const f32x2 e = {__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
    const f32x2 h = ((p * t) * e) * 0.5f;            // erfc / 2
    const f32x2 g = 1.0f - h;
    const f32x2 phi = {x.x > 0.0f ? g.x : h.x, x.y > 0.0f ? g.y : h.y};
    return x * phi;
  }
  template <int N>
  __device__ __forceinline__ void apply_vec(const float (&x)[N], float (&y)[N], int64_t e0) const {
    if constexpr (FAST && N % 2 == 0) {
#pragma unroll
      for (int k = 0; k < N; k += 2) {
        const f32x2 r = apply_pair((f32x2){x[k], x[k + 1]});
        y[k] = r.x;
        y[k + 1] = r.y;
      }
    } else {
#pragma unroll
      for (int k = 0; k < N; k++) apply_one(x[k], y[k], e0 + k);
    }
  }
};

template <int DT>
__device__ __forceinline__ float round_dt(float v) {  // RNE to DT and back (exact for fp32)
  if (DT == DMXQ_BF16) return (float)(__bf16)v;
  if (DT == DMXQ_F16) return (float)(_Float16)opaque(v);
  return v;
}

// FAST: 16-bit outputs -- v_exp_f32 + v_rcp_f32 (relative error ~2^-21, far inside the 2^-9 / 2^-12 half-ulp of the
// output format); fp32 outputs keep expf and the IEEE division.
// NOTAIL: the caller's output cast flushes everything below ~1e-30 to +0, which is where the far negative tail (t < -87) lands
// whatever its last bits: x * rcp(d) = -0 there and the per-element branch around the IEEE division (five scalar instructions
// and their wait states per element) goes away.
template <bool FAST, bool NOTAIL = false>
__device__ __forceinline__ float sigmoid_mul(float x, float t) {  // x * sigmoid(t)
  if (FAST) {
    const float d = 1.0f + __builtin_amdgcn_exp2f(t * -1.4426950408889634f);
    if (NOTAIL) return x * __builtin_amdgcn_rcpf(d);
    // v_rcp_f32 flushes a denormal RESULT to zero (1 / d for d > 2^126): the far negative tail, where the true value
    // x / d is still a normal number (silu(-88) = -5.3e-37), takes the IEEE division instead (rare: t < -87)
    return d > 8.5e37f ? x / d : x * __builtin_amdgcn_rcpf(d);
  }
  return x / (1.0f + expf(-t));
}

template <int KIND, int DTI, bool FAST, bool NOTAIL = false>
struct UnaryOp {
  static constexpr bool kHeavy = true;
  static constexpr int kTileUnroll = KIND == DMXQ_UNARY_SILU ? 8 : (KIND == DMXQ_UNARY_QUICK_GELU ? 2 : 4);  // stream.hpp
  static constexpr int kTileUnrollF32 = kTileUnroll;
  // common.hpp OpLoadPace (round 5): 32 idle issue cycles between a wave's loads on SiLU's 256 x 8 tiles -- 4096 x 4096 bf16 11.50 -> 10.42 us
  // (73 -> 80.5 %; pace 2 / 4 / 6: 11.09 / 10.42 / 10.48, library builds in one lease); QuickGELU 13.10 -> 12.96 (not taken), erf GELU nothing
  static constexpr int kLoadPace = KIND == DMXQ_UNARY_SILU ? 4 : 0;
  // the fused module on 16-bit tensors (act_cast.hip): silu 512 x 16 10.6 / 11.0 / 11.7 us on 3072 / 3584 / 4096 x 4096 bf16 (256 x 8: 9.3 / 12.8 / 13.3)
  static constexpr int kCastUnroll = KIND == DMXQ_UNARY_SILU ? 16 : kTileUnroll, kCastThreads = KIND == DMXQ_UNARY_SILU ? 512 : 256;
  float param;
  __device__ __forceinline__ void apply_one(float x, float& y, int64_t) const {
    if (KIND == DMXQ_UNARY_SILU) {
      y = sigmoid_mul<FAST, NOTAIL>(x, x);
    } else if (KIND == DMXQ_UNARY_EXP) {
      if (FAST) {
        // v_exp_f32 flushes denormal RESULTS: below 2^-126 the argument is raised by 64 and the result scaled back (exact)
        const float t = x * 1.4426950408889634f;
        const bool tiny = t < -126.0f;
        y = __builtin_amdgcn_exp2f(tiny ? t + 64.0f : t) * (tiny ? 5.421010862427522e-20f : 1.0f);
      } else {
        y = expf(x);
      }
    } else if (KIND == DMXQ_UNARY_QUICK_GELU) {
      const float t = round_dt<DTI>(1.702f * x);
      float s;
      if (FAST && DTI != DMXQ_F32) {
        const float d = 1.0f + __builtin_amdgcn_exp2f(t * -1.4426950408889634f);
        s = (!NOTAIL && d > 8.5e37f) ? 1.0f / d : __builtin_amdgcn_rcpf(d);  // (v_rcp_f32 flushes denormal results: see sigmoid_mul)
      } else {
        s = 1.0f / (1.0f + expf(-t));
      }
      s = round_dt<DTI>(s);
      y = x * s;
    } else {  // DMXQ_UNARY_SILU_EXPERIMENTAL: relu(half(x)) * scale, the product rounded to half by the store
      const float h = round_dt<DMXQ_F16>(x);
      const float r = h < 0.0f ? 0.0f : h;  // at::relu == clamp_min(0): NaN stays NaN, -0.0 stays -0.0 (max(-0, +0) keeps the first)
      y = r * param;
    }
  }
  template <int N>
  __device__ __forceinline__ void apply_vec(const float (&x)[N], float (&y)[N], int64_t e0) const {
    if constexpr (KIND == DMXQ_UNARY_SILU && FAST && NOTAIL && N % 2 == 0) {
      // x / (1 + exp(-x)) on pairs (packed fp32 pipe: GeluOp::apply_pair): 3.5 VALU per element
#pragma unroll
      for (int k = 0; k < N; k += 2) {
        const f32x2 v = {x[k], x[k + 1]};
        const f32x2 a = v * -1.4426950408889634f;
        f32x2 d = {__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
        d = d + 1.0f;
        const f32x2 r = v * (f32x2){__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
        y[k] = r.x;
        y[k + 1] = r.y;
      }
    } else {
#pragma unroll
      for (int k = 0; k < N; k++) apply_one(x[k], y[k], e0 + k);
    }
  }
};

}  // namespace dmxq
