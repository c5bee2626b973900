// csrc/bfp_math.hpp — per-block / per-element BFP arithmetic shared by bfp.hip and tools/tune_bfp.hip.
//
// Reference sequence (quant_cpu.cpp:239-275, oracle/oracle.c bfp_q1), per block with max|x| = m:
//   E = bits(m) & 0x7F800000 ; base = 6 * float(E)
//   t  = fl(x + base)                                   first rounding (fp32 RNE add); t in [4,8] * 2^e
//   t' = t's mantissa rounded to wl bits, on the bit pattern (nearest-even / down / up / stochastic)
//   q  = t' - base ; if exponent(q) > E: q = sign | E | top (wl-2) mantissa bits
//
// (1) For a normal, finite block maximum the clip is med3(q, -maxv, +maxv): t' stays inside [4,8]*2^e, so
//     |q| <= 2^(e+1) and "exponent field above E" <=> |q| == 2^(e+1) <=> |q| > maxv = 2^(e+1) - quantum; the clamp
//     returns the same bits.  It is NOT the same for a denormal maximum (E == 0: no rebase, wl kept bits, and only
//     a carry into exponent field 1 is clipped — non-monotonic) nor for poisoned blocks, so the literal path keeps
//     the reference's exponent-field compare, including the sign x86 gives inf - inf (negative quiet NaN).
// (2) Nearest-even on the bit pattern == one more fp32 add: with quantum = 2^(e+2-wl) (the value of t's wl-th
//     mantissa bit) and M = 1.5 * 2^(23) * quantum, fl(t + M) lies in M's binade, whose ulp IS the quantum, so the
//     hardware's RNE add rounds t to the nearest multiple of the quantum, ties to the even multiple — the same
//     parity as t's kept LSB because M/quantum = 1.5*2^23 is even.  K = M + base is exactly representable
//     (25-wl significant bits), and fl(t + M) - K = t' - base exactly.  So
//         q = (fl(fl(x + base) + M)) - K                       ["double" form: any input dtype, wl <= 20]
//     and when fl(x + base) cannot matter — 16-bit inputs with few enough significant bits that the first
//     rounding is either exact or far (>= 2x) from any tie of the second: bf16 (8 bits) wl <= 14, fp16 (11 bits)
//     wl <= 11 — the two roundings collapse into one:
//         q = fl(x + K) - K                                    ["single" form]
//     E == 0 (denormal block max): base = 0 and the kept bits are the top wl of the 23-bit field, i.e.
//     quantum = 2^(-126-wl); same formulas with that quantum.
//     Blocks whose M would overflow (finite maxima with biased exponent > 229 + wl) and blocks with a denormal
//     maximum (sign of a zero result, see fast_ok) take the literal bit path; the choice is made per wave (uniform
//     branch).  Inf/NaN maxima -- which the reference turns into an all-NaN block -- stay on the fast path: K = inf.
// (3) Asymmetric formats ("(_N)", format.py:349-372) relax only the negative clip by one code; closed form:
//         x <= -(2^(e+1) - quantum/2)  ->  y = -2^(e+1)        (tie goes to the even code -2^(wl-1)).
#pragma once
#include "common.hpp"

namespace dmxq {

struct BfpBlockParams {
  float base;     // 6 * 2^e
  float maxv;     // largest representable magnitude, 2^(e+1) - quantum
  float thr;      // asymmetric threshold  -(2^(e+1) - quantum/2)
  float neg_lim;  // -2^(e+1)
  float M;        // 1.5 * 2^23 * quantum   (fast path)
  float K;        // M + base               (fast path)
  uint32_t E;     // exponent field of the block maximum
};

// maxabs_bits: fp32 bit pattern of the block's max|x| (a NaN/Inf pattern gives E = 0xFF: base = inf, every
// element becomes NaN — the reference poisons such a block the same way, torch.max propagating NaN).
template <bool ASYM, bool FAST = false>
__device__ __forceinline__ BfpBlockParams bfp_block_params(uint32_t maxabs_bits, int wl) {
  BfpBlockParams p;
  const uint32_t E = maxabs_bits & 0x7F800000u;
  p.base = u2f(E) * 6.0f;
  const uint32_t max_man = (0x007FFFFFu >> (25 - wl)) << (25 - wl);
  // FAST, Inf/NaN maximum (E = 0xFF): base = K = inf, so every element becomes inf - inf = NaN as in the reference;
  // the clamp limits are made NaN too, so that med3 (min3 when an operand is NaN) cannot turn the NaN into a limit
  const uint32_t nan_lim = (FAST && E == 0x7F800000u) ? 0x00400000u : 0u;
  p.maxv = u2f(E | max_man | nan_lim);
  p.E = E;
  if (ASYM) {
    const uint32_t thr_man = (0x007FFFFFu >> (24 - wl)) << (24 - wl);
    // a block poisoned to NaN (max >= 2^126 or Inf/NaN: base overflows) stays NaN: make the compare always false
    p.thr = E >= 0x7E800000u ? u2f(0x7FC00000u) : u2f(0x80000000u | E | thr_man);
    p.neg_lim = nan_lim ? u2f(0xFFC00000u) : u2f(0x80000000u | (E + 0x00800000u));
  }
  if (FAST) {
    const uint32_t eb = E >> 23;
    uint32_t mexp = eb ? eb + 25u - (uint32_t)wl : 24u - (uint32_t)wl;  // biased exponent of M
    mexp = mexp < 254u ? mexp : 254u;                                    // (only E = 0xFF gets here with more)
    p.M = u2f((mexp << 23) | 0x00400000u);
    p.K = p.M + p.base;
  }
  return p;
}

// may this block take the magic-add path?  M must be representable, and:
// E == 0 with a non-zero (denormal) maximum: base = 0, so a negative x that rounds to zero keeps its sign in the
// reference (-0.0) while the magic add yields +0.0 -> literal path.  All-zero blocks stay fast.
// An Inf maximum (e.g. the -inf of an attention mask) is fine: the formulas produce the all-NaN block by themselves, see
// bfp_block_params.  A NaN maximum is NOT: the reference rounds the mantissa bits of a NaN ELEMENT like any other value, so what
// such an element becomes depends on its payload (0xFFFF0000 at wl = 4 carries into the exponent and comes out as -inf, the
// canonical 0x7FC00000 stays NaN) -- only the literal bit path reproduces that.
__device__ __forceinline__ bool bfp_fast_ok(uint32_t maxabs_bits, int wl) {
  const uint32_t eb = (maxabs_bits & 0x7F800000u) >> 23;
  return (eb + 25u - (uint32_t)wl <= 254u || maxabs_bits == 0x7F800000u) && (eb != 0u || maxabs_bits == 0u);
}

// literal bit path (every rounding mode)
template <int RND, bool ASYM>
__device__ __forceinline__ float bfp_q1(float x, const BfpBlockParams& p, int wl, int rounding, uint32_t rnd) {
  const float t = x + p.base;
  const uint32_t tb = round_bitwise<RND>(f2u(t), wl, rounding, rnd);
  float q = u2f(tb) - p.base;
  uint32_t qb = f2u(q);
  if (q != q) qb |= 0x80000000u;  // the reference runs on x86, where inf - inf is the NEGATIVE quiet NaN
  if ((qb & 0x7F800000u) > p.E) qb = (qb & 0x80000000u) | f2u(p.maxv);  // bit_helper.cpp:24-37, literally
  q = u2f(qb);
  if (ASYM) q = (x <= p.thr) ? p.neg_lim : q;
  return q;
}

// (5) the other rounding modes (down / up / stochastic act on t's bit pattern) keep the literal rounding and only
//     replace the exponent-field clip by the clamp of (1): valid for a normal block maximum whose base is finite
//     (biased exponent 1..252), and for an all-zero block unless the mode is "up" (which turns 0 into the smallest
//     kept code, 2^(-126-wl), and at wl = 2 the clamp limit is 0).
__device__ __forceinline__ bool bfp_bitfast_ok(uint32_t maxabs_bits, int rounding) {
  const uint32_t eb = (maxabs_bits & 0x7F800000u) >> 23;
  return (eb >= 1u && eb <= 252u) || (maxabs_bits == 0u && rounding != DMXQ_ROUND_UP);
}
template <int RND, bool ASYM>
__device__ __forceinline__ float bfp_q1_bitfast(float x, const BfpBlockParams& p, int wl, int rounding, uint32_t rnd) {
  const float t = x + p.base;
  const uint32_t tb = round_bitwise<RND>(f2u(t), wl, rounding, rnd);
  float q = __builtin_amdgcn_fmed3f(u2f(tb) - p.base, -p.maxv, p.maxv);
  if (ASYM) q = (x <= p.thr) ? p.neg_lim : q;
  return q;
}

// nearest-even fast path, see (2); only for blocks that pass bfp_fast_ok (normal finite maximum, or all zero)
template <bool SINGLE, bool ASYM>
__device__ __forceinline__ float bfp_q1_fast(float x, const BfpBlockParams& p) {
  if (SINGLE) {
    const float q = (x + p.K) - p.K;
    // single rounding: q == -2^(e+1) exactly when x <= thr (x < -(2^(wl-1) - 1/2) quanta rounds to the code
    // -2^(wl-1), and the tie itself goes to that even code), so the asymmetric format is just an asymmetric clamp
    return __builtin_amdgcn_fmed3f(q, ASYM ? p.neg_lim : -p.maxv, p.maxv);
  }
  const float t = x + p.base;
  float q = (t + p.M) - p.K;
  q = __builtin_amdgcn_fmed3f(q, -p.maxv, p.maxv);
  if (ASYM) q = (x <= p.thr) ? p.neg_lim : q;  // the first rounding may cross thr: compare the original x
  return q;
}

// may the two roundings be collapsed for this input dtype / precision?  (see (2))
template <int DT>
inline bool bfp_single_rounding_ok(int wl) {
  return (DT == DMXQ_BF16 && wl <= 14) || (DT == DMXQ_F16 && wl <= 11);
}

}  // namespace dmxq
