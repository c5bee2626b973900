// csrc/act_cast.hip — an activation-function DmxModule in ONE pass (VERDICT r2 "missing" 2):
//     out = cast_out( f( cast_in(x) ) )        f in {GELU erf / tanh, SiLU, QuickGELU, Exp}
// Replaces the three launches of such a module (modeling/nn/core.py:228-264: CastTo on the input, the exact torch function --
// functional/approximate.py:300-304 with vsimd absent --, CastTo on the output; in BASIC mode both casts are FLOAT16,
// src/dmx/compressor/__init__.py:360-455): 4 B/element for a 16-bit tensor instead of 12.
//
// Value contract (the function itself is floating point: no bit-exact reference exists for it):
//   * the two casts are the library's bit-exact casts (dmxq_float_qdq arithmetic) with CastTo's `.to(dtype)` after each;
//   * f is evaluated in fp32 on the cast input and rounded ONCE to the tensor dtype, which is how torch evaluates these
//     functions on a 16-bit tensor (QUICK_GELU: in the tensor dtype, three roundings, like transformers' module);
//   * so out == cast_out(v) for a v within 1 ulp (of the tensor dtype) of the correctly rounded f(cast_in(x)): for 16-bit
//     tensors the fp32 evaluation is within 2^-20 relative of the truth, for float32 tensors within the ulps stated in
//     DESIGN.md §4.1.  Tests: tests/test_gpu_act_cast.py (float64 truth, the casts from the oracle).
// Two forms, like dmxq_binary_cast: 16-bit tensors whose casts are range-only (bf16 with >= 7, fp16 with >= 10 mantissa bits,
// subnormals flushed: FLOAT16 / BFLOAT16 of the BASIC rules) apply them on the packed words (stream.hpp raw hooks); float32
// tensors (any FloatingPoint format, nearest) per element with the magic-add cast of floatq.hpp.  Other combinations
// (a rounding cast on a 16-bit tensor) return DMXQ_ERR_UNSUPPORTED and the caller runs the three launches.
#include <math.h>

#include "floatq.hpp"
#include "stream.hpp"
#include "unary_ops.hpp"

namespace dmxq {

template <class BASE, int DT>
struct CastedOp {
  static constexpr bool kHeavy = true;
  static constexpr int kTileUnroll = DT == DMXQ_F32 ? BASE::kTileUnrollF32 : BASE::kCastUnroll;  // (stream.hpp: geometry for 20-32 MiB tensors)
  static constexpr int kTileThreads = DT == DMXQ_F32 ? 256 : BASE::kCastThreads;
  static constexpr bool kRawHooks = DT != DMXQ_F32;
  BASE base;
  Range16 ri, ro;  // 16-bit tensors
  CastG gi, go;    // float32 tensors
  __device__ __forceinline__ void raw_in(u32x4& r) const {
#pragma unroll
    for (int j = 0; j < 4; j++) r[j] = range16_word(r[j], ri);
  }
  template <class OV>
  __device__ __forceinline__ void raw_out(OV& o) const {
#pragma unroll
    for (int j = 0; j < OV::kWords; j++) o.w[j] = range16_word(o.w[j], ro);
  }
  template <int N>
  __device__ __forceinline__ void apply_vec(const float (&x)[N], float (&y)[N], int64_t e0) const {
    if (DT == DMXQ_F32) {
      float xc[N];
#pragma unroll
      for (int k = 0; k < N; k++) xc[k] = x[k];
      castg_vec<DT, N>(xc, gi);
      base.apply_vec(xc, y, e0);
      castg_vec<DT, N>(y, go);
    } else {
      // (a NaN result keeps the sign the hardware gives it: torch's CPU bf16 conversion makes every NaN positive, its GPU one does
      // not, and a cast without NaN codes then saturates to +-max accordingly -- the magnitude is the contract here)
      base.apply_vec(x, y, e0);
    }
  }
  __device__ __forceinline__ void apply_one(float, float&, int64_t) const {}  // (n % EPL == 0 is required: no scalar tail)
};

struct ActCasts { Range16 ri, ro; CastG gi, go; };

template <class BASE, int DT>
static int launch_casted(const void* in, void* out, int64_t n, const BASE& base, const ActCasts& c, hipStream_t s) {
  return launch_stream<DT, DT, CastedOp<BASE, DT>>(in, out, n, CastedOp<BASE, DT>{base, c.ri, c.ro, c.gi, c.go}, s);
}

// accurate libm forms only where an fp32 result can be seen: float32 tensors whose output cast keeps more than 16 mantissa bits
template <int KIND>
static int launch_unary_cast(const void* in, void* out, int dtype, int64_t n, float param, const ActCasts& c, bool fast32, bool notail, hipStream_t s) {
  if (notail && dtype == DMXQ_BF16) return launch_casted<UnaryOp<KIND, DMXQ_BF16, true, true>, DMXQ_BF16>(in, out, n, {param}, c, s);
  if (notail && dtype == DMXQ_F16) return launch_casted<UnaryOp<KIND, DMXQ_F16, true, true>, DMXQ_F16>(in, out, n, {param}, c, s);
  if (dtype == DMXQ_BF16) return launch_casted<UnaryOp<KIND, DMXQ_BF16, true>, DMXQ_BF16>(in, out, n, {param}, c, s);
  if (dtype == DMXQ_F16) return launch_casted<UnaryOp<KIND, DMXQ_F16, true>, DMXQ_F16>(in, out, n, {param}, c, s);
  if (fast32) return launch_casted<UnaryOp<KIND, DMXQ_F32, true>, DMXQ_F32>(in, out, n, {param}, c, s);
  return launch_casted<UnaryOp<KIND, DMXQ_F32, false>, DMXQ_F32>(in, out, n, {param}, c, s);
}
template <bool TANH>
static int launch_gelu_cast(const void* in, void* out, int dtype, int64_t n, const ActCasts& c, bool fast32, hipStream_t s) {
  if (dtype == DMXQ_BF16) return launch_casted<GeluOp<true, TANH>, DMXQ_BF16>(in, out, n, {}, c, s);
  if (dtype == DMXQ_F16) return launch_casted<GeluOp<true, TANH>, DMXQ_F16>(in, out, n, {}, c, s);
  if (fast32) return launch_casted<GeluOp<true, TANH>, DMXQ_F32>(in, out, n, {}, c, s);
  return launch_casted<GeluOp<false, TANH>, DMXQ_F32>(in, out, n, {}, c, s);
}

}  // namespace dmxq

using namespace dmxq;

extern "C" int dmxq_unary_cast(const void* in, void* out, int dtype, int64_t n, int kind, float param, const dmxq_float_fmt* cast_in,
                               const dmxq_float_fmt* cast_out, void* stream) {
  if (!valid_dtype(dtype) || n < 0 || kind < DMXQ_UNARY_GELU || kind > DMXQ_UNARY_EXP) return DMXQ_ERR_BAD_ARG;
  if (n == 0) return DMXQ_OK;
  if (!in || !out) return DMXQ_ERR_BAD_ARG;
  const int epl = dtype == DMXQ_F32 ? 4 : 8;
  if (n % epl != 0 || !aligned16(in) || !aligned16(out)) return DMXQ_ERR_UNSUPPORTED;
  ActCasts c{};
  bool fast32 = false;
  if (dtype == DMXQ_F32) {
    if (!castg_of(cast_in, &c.gi) || !castg_of(cast_out, &c.go)) return DMXQ_ERR_UNSUPPORTED;
    // v_exp / v_rcp forms (relative error ~2^-21) are invisible behind an output cast that keeps <= 16 mantissa bits
    fast32 = c.go.active && c.go.f.man <= 16;
  } else if (!range16_of(cast_in, dtype, &c.ri) || !range16_of(cast_out, dtype, &c.ro)) {
    return DMXQ_ERR_UNSUPPORTED;
  }
  hipStream_t s = (hipStream_t)stream;
  // the far tail of silu / quick_gelu (|result| < 1e-35) needs no exact last bits when the output cast flushes it: a 16-bit tensor
  // whose output cast has a smallest normal value of 2^-100 or more
  const bool notail = dtype != DMXQ_F32 && cast_out && cast_out->exp_bits != 0 && cast_out->flush_subnormal && -(cast_out->exp_bias - 1) >= -100;
  switch (kind) {
    case DMXQ_UNARY_GELU: return launch_gelu_cast<false>(in, out, dtype, n, c, fast32, s);
    case DMXQ_UNARY_GELU_TANH: return launch_gelu_cast<true>(in, out, dtype, n, c, fast32, s);
    case DMXQ_UNARY_SILU: return launch_unary_cast<DMXQ_UNARY_SILU>(in, out, dtype, n, param, c, fast32, notail, s);
    case DMXQ_UNARY_QUICK_GELU: return launch_unary_cast<DMXQ_UNARY_QUICK_GELU>(in, out, dtype, n, param, c, fast32, notail, s);
    default: return launch_unary_cast<DMXQ_UNARY_EXP>(in, out, dtype, n, param, c, fast32, false, s);
  }
}
