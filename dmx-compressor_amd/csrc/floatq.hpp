// csrc/floatq.hpp — branch-free nearest-even quantisation to a low-bit float format, shared by the elementwise float
// cast (elementwise.hip) and the MX block formats (blockfmt.hip).  Reference: quant_cpu.cpp:359-402, bit_helper.cpp:4-22.
#pragma once
#include <math.h>
#include <string.h>

#include "common.hpp"

namespace dmxq {

// Nearest-even without the two data-dependent branches (normal / subnormal of the simulated format; both run in
// practically every wave, ~30 VALU instructions per element together).  With E' = max(exponent(a), min_exp):
//   M = +-1.5 * 2^(23 + E' - man)   (fl(t + M) lies in M's binade, whose ulp is the quantum 2^(E' - man): the add rounds
//                                    to the nearest multiple, ties to the even one = the kept LSB of the bit pattern)
//   S = +-2^min_exp if exponent(a) < min_exp else 0      (the reference's subnormal shift; its add is the same fp32 add)
//   q = fl(fl(a + S) + M) - (M + S),  then the saturation as a clamp at +-max_val (rounding is monotone, so "rounded
//   exponent > max_e" <=> |q| > max_val)
// all constants carrying a's sign, which also reproduces the +0 the reference returns when a negative subnormal
// rounds to zero.  Valid for finite a with exponent <= 103 + man (M finite) and 1 <= man <= 20; the caller falls back to
// the bit-level form for a wave that holds anything else.
struct FloatFast {
  uint32_t min_exp_bits;  // (127 + min_exp) << 23
  uint32_t k1;            // ((23 - man) << 23) | 0x00400000
  uint32_t max_ok_bits;   // largest exponent field the fast form takes
  float max_val;          // saturation value, +inf when the format's exponent range is fp32's
  int usable;             // man <= 20 and min_exp in range
};
inline FloatFast make_float_fast(int man, int exp_bits, int bias) {
  FloatFast k{};
  const int min_exp = -(bias - 1);
  // man = 0: the reference's tie rule then looks at the exponent's lowest bit, not at a mantissa bit -> bit-level form
  k.usable = (man >= 1 && man <= 20 && min_exp >= -126 && min_exp <= 100) ? 1 : 0;
  k.min_exp_bits = (uint32_t)(127 + (k.usable ? min_exp : 0)) << 23;
  k.k1 = ((uint32_t)(23 - man) << 23) | 0x00400000u;
  k.max_ok_bits = (uint32_t)(103 + man + 127 > 254 ? 254 : 103 + man + 127) << 23;
  const int max_e = (1 << (exp_bits - 1)) + 127;
  if (max_e >= 255) {
    k.max_val = INFINITY;
  } else {
    uint32_t b = ((uint32_t)max_e << 23) | ((0x007FFFFFu >> (23 - man)) << (23 - man));
    float v;
    memcpy(&v, &b, 4);
    k.max_val = v;
  }
  return k;
}
__device__ __forceinline__ bool float_fast_ok(float a, const FloatFast& k) {
  const uint32_t eb = f2u(a) & 0x7F800000u;
  return eb <= k.max_ok_bits;  // finite, and M representable
}
__device__ __forceinline__ float float_q1_fast(float a, const FloatFast& k, bool flush, bool unsigned_abs) {
  const uint32_t t = f2u(a), sign = t & 0x80000000u, eb = t & 0x7F800000u;
  const bool sub = eb < k.min_exp_bits;
  const uint32_t e2 = sub ? k.min_exp_bits : eb;
  const float M = u2f((e2 + k.k1) | sign);
  const float S = u2f(sub ? (k.min_exp_bits | sign) : sign);  // +-2^min_exp, or +-0
  float q = ((a + S) + M) - (M + S);
  q = __builtin_amdgcn_fmed3f(q, -k.max_val, k.max_val);
  if (flush) q = sub ? 0.0f : q;
  return unsigned_abs ? fabsf(q) : q;
}


// ---------------------------------------------------------------------------------------------------------------------------
// The bit-level form (every rounding mode) and the per-element cast used by the fused modules (elementwise.hip, rope.hip)
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

// Formats that FLUSH their subnormals (FLOAT16, BFLOAT16 of the BASIC rules) on finite inputs, on the magnitude bits: round to
// `man` bits half-to-even (add half - 1 + kept LSB, mask: quant_cpu.cpp:211-237 -- the carry may ripple into the exponent),
// saturate at sign | max_e | max_man (a rounded pattern above it has an exponent above max_e, bit_helper.cpp:4-22), and +0 where
// the INPUT's exponent is below the smallest normal one (quant_cpu.cpp:372-376 decides on the input).  8 integer operations per
// element instead of the ~17 of the magic-add form; Inf / NaN inputs (and man = 0, whose tie rule reads the exponent's lowest
// bit) keep the bit-level form through the caller's wave-uniform fallback.
struct FlushFast { uint32_t half_m1, mask, lim_bits, min_bits; int sh, usable; };
inline FlushFast make_flush_fast(int man, int exp_bits, int bias, int flush) {
  FlushFast k{};
  const int min_exp = -(bias - 1), max_e = (1 << (exp_bits - 1)) + 127;
  k.usable = (flush && man >= 1 && man <= 22 && min_exp >= -126 && min_exp <= 127) ? 1 : 0;
  k.sh = 23 - man;
  k.mask = (1u << (23 - man)) - 1u;
  k.half_m1 = (1u << (22 - man)) - 1u;
  k.lim_bits = max_e >= 255 ? 0x7F800000u : (((uint32_t)max_e << 23) | ((0x007FFFFFu >> (23 - man)) << (23 - man)));  // max_e = 255: a finite input can at most round up to Inf
  k.min_bits = (uint32_t)(127 + (k.usable ? min_exp : 0)) << 23;
  return k;
}
__device__ __forceinline__ float float_q1_flush(float a, const FlushFast& k) {
  const uint32_t u = f2u(a), m = u & 0x7FFFFFFFu;
  uint32_t r = (m + k.half_m1 + ((m >> k.sh) & 1u)) & ~k.mask;
  r = r < k.lim_bits ? r : k.lim_bits;
  r |= u & 0x80000000u;
  return u2f(m < k.min_bits ? 0u : r);
}

struct CastG { FloatFmt f; FloatFast k; FlushFast ff; int active; };
inline bool castg_of(const dmxq_float_fmt* f, CastG* c) {  // false: not a format the kernels take
  if (!f || f->exp_bits == 0) { c->active = 0; c->f = FloatFmt{}; c->k = FloatFast{}; c->ff = FlushFast{}; return true; }
  if (f->exp_bits < 1 || f->exp_bits > 8 || f->man_bits < 0 || f->man_bits > 22) return false;
  c->f = FloatFmt{f->man_bits, f->exp_bits, f->exp_bias, f->flush_subnormal ? 1 : 0, 0, DMXQ_ROUND_NEAREST, 0ull};
  c->k = make_float_fast(f->man_bits, f->exp_bits, f->exp_bias);
  c->ff = make_flush_fast(f->man_bits, f->exp_bits, f->exp_bias, f->flush_subnormal ? 1 : 0);
  c->active = 1;
  return true;
}
template <int DT>
__device__ __forceinline__ float castg_dt(float x) {  // CastTo's `.to(physical dtype)` / torch's rounding of an op's fp32 result
  // c10::BFloat16 turns EVERY NaN into +0x7FC0 (c10/util/BFloat16.h round_to_nearest_even), c10::Half keeps the sign: it matters
  // here because a following cast without NaN codes saturates a NaN to sign | max_val
  if (DT == DMXQ_BF16) return x != x ? u2f(0x7FC00000u) : (float)(__bf16)x;
  if (DT == DMXQ_F16) return (float)(_Float16)opaque(x);
  return x;
}
template <int DT, int N>
__device__ __forceinline__ void castg_vec(float (&x)[N], const CastG& c) {
  if (!c.active) return;  // (wave-uniform, once per vector)
  if (c.ff.usable) {      // flush formats: the integer form, finite inputs
    bool fin = true;
    float q[N];
#pragma unroll
    for (int j = 0; j < N; j++) { fin = fin && (f2u(x[j]) & 0x7F800000u) != 0x7F800000u; q[j] = float_q1_flush(x[j], c.ff); }
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!fin) != 0ull, 0)) {
#pragma unroll
      for (int j = 0; j < N; j++)
        if ((f2u(x[j]) & 0x7F800000u) == 0x7F800000u) q[j] = float_q1<DMXQ_ROUND_NEAREST>(x[j], c.f, 0u);
    }
#pragma unroll
    for (int j = 0; j < N; j++) x[j] = castg_dt<DT>(q[j]);
    return;
  }
  bool ok = c.k.usable != 0;
  float q[N];
#pragma unroll
  for (int j = 0; j < N; j++) { ok = ok && float_fast_ok(x[j], c.k); q[j] = float_q1_fast(x[j], c.k, c.f.flush != 0, false); }
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(!ok) != 0ull, 0)) {
#pragma unroll
    for (int j = 0; j < N; j++)
      if (!c.k.usable || !float_fast_ok(x[j], c.k)) q[j] = float_q1<DMXQ_ROUND_NEAREST>(x[j], c.f, 0u);
  }
#pragma unroll
  for (int j = 0; j < N; j++) x[j] = castg_dt<DT>(q[j]);
}

}  // namespace dmxq
