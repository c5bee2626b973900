/* oracle/oracle.c — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference's (d-matrix-ai/dmx-compressor v0.1.11) CPU algorithms for the
 * fake-quantisation hot path.  It exists so that the HIP kernels can be checked bit-for-bit on the GPU box,
 * where /root/reference does not exist.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load it; the product package (dmx-compressor_amd/) never does.
 *
 * PINNING: this file is validated bit-exactly against (a) the reference's own compiled C++ extension
 * (oracle/_ref/quant_cpu.so, built from the reference sources in place by oracle/Makefile), (b) the
 * reference's Python path (numerical/format.py, cast.py, sparse.py imported through oracle/ref_shim.py) by
 * oracle/gen_golden.py, and (c) the committed fixtures in tests/golden/ (which include the reference's own
 * known-answer values, tests/test_bfp.py:26-65 and tests/test_group_quant.py:49-63).
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/src/dmx/compressor/).  The code is a restatement, not a copy: whole-tensor, layout-explicit,
 * ATen-free, with 64-bit sizes.
 *
 * Build: gcc -O2 -fPIC -shared -fopenmp -ffp-contract=off -fno-fast-math  (see oracle/Makefile)
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* rounding codes shared with include/dmxq.h (order of the reference's `enum Mode`, quant/quant_cpu/quant_cpu.cpp:9-15) */
enum { R_UP = 0, R_DOWN = 1, R_NEAREST = 2, R_STOCHASTIC = 3 };

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* Counter-based random bits for the stochastic mode.  The reference uses an UNSEEDED global mt19937
 * (quant_cpu.cpp:32-34), so its stochastic results are not reproducible; parity there is statistical only.
 * The HIP kernels use the same stream, so kernel-vs-oracle is still bit-exact: a 32-bit avalanche hash (two odd
 * multiplies, three xor-shifts) of the low half of the linear element index, keyed by the seed and the high half of
 * the index -- 32-bit on purpose, a 64-bit multiply costs a GPU lane four quarter-rate instructions. */
static inline uint32_t rnd_bits(uint64_t seed, uint64_t idx) {
  const uint32_t key = (uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x9E3779B9u) ^ ((uint32_t)(idx >> 32) * 0x85EBCA6Bu);
  uint32_t x = (uint32_t)idx ^ key;
  x ^= x >> 16; x *= 0x7FEB352Du;
  x ^= x >> 15; x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}

/* The draws of the BFP cast's stochastic mode (round 5; csrc/common.hpp bfp_rnd is the same function): one hash per aligned group of 8
 * elements; per pair a Weyl step, a xor-shift, a 24-bit multiply (full rate on the GPU) and a xor-shift; the pair's word for the even
 * element, the word with its halves swapped for the odd one.  A function of (seed, linear element index) like rnd_bits; unbiased and
 * pairwise uncorrelated by measurement (tests/test_golden.py). */
static inline uint32_t bfp_rnd(uint64_t seed, uint64_t idx) {
  const uint32_t pair = ((uint32_t)idx & 7u) >> 1;
  uint32_t w = rnd_bits(seed, idx >> 3);
  if (pair != 0u) {   /* (the group's hash itself serves its first pair) */
    w += pair * 0x9E3779B9u;
    w ^= w >> 15;
    w = (w & 0xFFFFFFu) * 0xB5297Bu;
    w ^= w >> 12;
  }
  return ((uint32_t)idx & 1u) ? ((w >> 16) | (w << 16)) : w;
}

/* quant_cpu.cpp:211-237 round_bitwise: keep `man_bits` mantissa bits of an fp32 bit pattern.
 * nearest = round-half-to-even on the bit pattern; the carry may ripple into the exponent (intended).
 * Valid for 0 <= man_bits <= 22 (the reference shifts by a negative count at 23: undefined behaviour). */
static inline uint32_t round_bitwise(uint32_t target, int man_bits, int rounding, uint32_t rnd) {
  const uint32_t mask = (1u << (23 - man_bits)) - 1u;
  uint32_t add;
  if (rounding == R_STOCHASTIC) {
    add = rnd & mask;
  } else if (rounding == R_NEAREST) {
    add = 1u << (22 - man_bits);
    if ((target & mask) == add && ((target >> (23 - man_bits)) & 1u) == 0u) add = 0u; /* tie, kept LSB even */
  } else if (rounding == R_DOWN) {
    add = 0u;
  } else {
    add = 1u << (23 - man_bits);
  }
  return (target + add) & ~mask;
}

/* quant_cpu/bit_helper.cpp:4-22 clip_exponent: saturate to the largest finite value of the simulated format.
 * No exponent code is reserved for inf/nan: max stored exponent = 2^(exp_bits-1) + 127. */
static inline uint32_t clip_exponent(int exp_bits, int man_bits, uint32_t old_num, uint32_t q) {
  if (q == 0u) return q;
  const int qe = (int)((q & 0x7FFFFFFFu) >> 23);
  const int max_e = (1 << (exp_bits - 1)) + 127;
  if (qe > max_e) {
    const uint32_t max_man = (0x007FFFFFu >> (23 - man_bits)) << (23 - man_bits);
    q = (old_num & 0x80000000u) | ((uint32_t)max_e << 23) | max_man;
  }
  return q;
}

/* quant_cpu/bit_helper.cpp:24-37 clip_max_exponent */
static inline uint32_t clip_max_exponent(int man_bits, uint32_t max_exponent, uint32_t q) {
  const uint32_t qe = q & 0x7F800000u;
  if (qe > max_exponent) {
    const uint32_t max_man = (0x007FFFFFu >> (23 - man_bits)) << (23 - man_bits);
    q = (q & 0x80000000u) | max_exponent | max_man;
  }
  return q;
}

/* ------------------------------------------------------------------------------------------- low-bit float
 * quant_cpu.cpp:359-402 float_quantize (one element). */
static inline float float_q1(float a, int man, int exp_bits, int bias, int flush, int rounding, uint32_t rnd) {
  uint32_t target = f2u(a);
  const int target_exp = (int)((target & 0x7FFFFFFFu) >> 23) - 127;
  const int min_exp = -(bias - 1);
  if (target_exp < min_exp) {
    if (flush) return 0.0f;
    const uint32_t shift_bits = ((uint32_t)(127 + min_exp) << 23) | (target & 0x80000000u);
    const float shift = u2f(shift_bits);
    const float val = a + shift;
    const uint32_t qb = round_bitwise(f2u(val), man, rounding, rnd);
    return u2f(qb) - shift;
  }
  uint32_t qb = round_bitwise(target, man, rounding, rnd);
  qb = clip_exponent(exp_bits, man, target, qb);
  return u2f(qb);
}

/* returns 0 ok, 1 bad argument (man outside 0..22: the reference's behaviour at man=23 is undefined) */
int oracle_float_qdq(const float* in, float* out, int64_t n, int man, int exp_bits, int bias, int flush,
                     int rounding, uint64_t seed) {
  if (man < 0 || man > 22 || exp_bits < 1 || exp_bits > 8) return 1;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++)
    out[i] = float_q1(in[i], man, exp_bits, bias, flush, rounding, rounding == R_STOCHASTIC ? rnd_bits(seed, (uint64_t)i) : 0u);
  return 0;
}

/* ------------------------------------------------------------------------------------------- block floating point
 * quant_cpu.cpp:239-275 block_quantize_helper for one element, given the block's max|x| (symmetric=true path,
 * the only one the Python layer ever requests: numerical/format.py:332). */
static inline float bfp_q1n(float x, float maxabs, int wl, int rounding, uint32_t rnd, int native_asym) {
  uint32_t max_num = f2u(maxabs);
  /* quant_cpu.cpp:247-253, the NATIVE symmetric == false branch (never requested by the Python layer,
   * numerical/format.py:332 forces true): an element that equals -max, when the top 7 mantissa bits of the maximum are
   * all ones, is quantised with the NEXT exponent (per element: the loop variable max_num is local to the element). */
  if (native_asym && x == -maxabs && ((max_num >> 16) << 25) == 0xFE000000u) max_num = ((max_num >> 23) + 1u) << 23;
  const uint32_t max_exp = max_num & 0x7F800000u; /* (max_num << 1 >> 24 << 23) */
  const float base = u2f(max_exp) * 6.0f;
  const float t = x + base;
  const uint32_t qb = round_bitwise(f2u(t), wl, rounding, rnd);
  const float q = u2f(qb) - base;
  return u2f(clip_max_exponent(wl - 2, max_exp, f2u(q)));
}

static inline float bfp_q1(float x, float maxabs, int wl, int rounding, uint32_t rnd) {
  const uint32_t max_exp = f2u(maxabs) & 0x7F800000u; /* (max_num << 1 >> 24 << 23) */
  const float base = u2f(max_exp) * 6.0f;
  const float t = x + base;                           /* fp32 RNE add: deliberate first rounding */
  const uint32_t qb = round_bitwise(f2u(t), wl, rounding, rnd);
  const float q = u2f(qb) - base;
  return u2f(clip_max_exponent(wl - 2, max_exp, f2u(q)));
}

/* numerical/format.py:349-372 make_mantissa_asymmetric, restated literally for one block of `len` elements
 * (q = symmetric result, x = original).  The reference runs it on a whole [N,B] chunk; rows are independent,
 * and its "no edge -> return unchanged" early-out yields the same values as the rebuild (ldexp of the
 * integer mantissas reproduces q exactly). */
static void bfp_asym_block(float* q, const float* x, int64_t len, int n) {
  int max_e = -200;
  for (int64_t i = 0; i < len; i++) {
    int e; float m = frexpf(q[i], &e);
    if (e == 0 && m == 0.0f) e = -200;
    if (e > max_e) max_e = e;
  }
  const int scale_e = max_e - n + 1;
  const int edge = -((1 << (n - 1)) - 1);
  for (int64_t i = 0; i < len; i++) {
    int e; float m = frexpf(q[i], &e);
    if (e == 0 && m == 0.0f) e = -200;
    int im = (int)(m * powf(2.0f, (float)(e - scale_e)));
    if (im == edge) {
      const float old_err = q[i] - x[i];
      const float cand_err = old_err - powf(2.0f, (float)scale_e);
      if (fabsf(cand_err) <= fabsf(old_err)) im -= 1;
      q[i] = ldexpf((float)im, scale_e);
    }
  }
}

/* numerical/format.py:304-343 BlockFloatingPoint.cast on a contiguous fp32 [rows, L] matrix with blocks of B
 * along the last dim; ragged last block = torch.split semantics (:324-326).  B==1 is routed to float_quantize
 * with man = wl-2, exp=8, bias=127, no flush (:312-320).
 * returns 0 ok, 1 bad argument. */
int oracle_bfp_qdq(const float* in, float* out, int64_t rows, int64_t L, int64_t B, int wl, int rounding,
                   int symmetric, uint64_t seed) {
  if (B < 1 || wl < 2) return 1;
  if (B == 1) return oracle_float_qdq(in, out, rows * L, wl - 2, 8, 127, 0, rounding, seed);
  if (wl > 22) return 1; /* reference shifts by a negative count: undefined */
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < rows; r++) {
    const float* xi = in + r * L;
    float* yo = out + r * L;
    for (int64_t b0 = 0; b0 < L; b0 += B) {
      const int64_t len = (L - b0 < B) ? (L - b0) : B;
      float m = 0.0f;
      /* quant_cpu.cpp:277-297 get_max_entry: torch's abs().max() PROPAGATES NaN, so a NaN (or Inf) element
       * makes the block's exponent field 0xFF, base = inf and every element of the block NaN. */
      for (int64_t i = 0; i < len; i++) { float a = fabsf(xi[b0 + i]); if (a > m || isnan(a)) m = a; }
      for (int64_t i = 0; i < len; i++)
        yo[b0 + i] = bfp_q1n(xi[b0 + i], m, wl, rounding,
                             rounding == R_STOCHASTIC ? bfp_rnd(seed, (uint64_t)(r * L + b0 + i)) : 0u, symmetric == 2);
      /* NaN/Inf maximum, or a maximum >= 2^126 whose base 6*2^e overflows: the symmetric pass already turned the
       * whole block into NaN and the reference's post-pass is garbage-in/garbage-out there -> the block stays NaN */
      /* symmetric: 1 = symmetric, 0 = the Python layer's asymmetric post-pass (format.py:349-372), 2 = the native
       * symmetric == false branch of block_quantize_helper (no post-pass) */
      if (symmetric == 0 && isfinite(m) && isfinite(u2f(f2u(m) & 0x7F800000u) * 6.0f)) bfp_asym_block(yo + b0, xi + b0, len, wl);
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------- fixed point
 * quant_cpu/sim_helper.cpp:5-12 fixed_min_max */
static void fixed_min_max(int wl, int fl, int symmetric, float* t_min, float* t_max) {
  const int sigma = -fl;
  *t_min = (float)(-ldexp(1.0, wl - fl - 1));
  *t_max = (float)(-(double)*t_min - ldexp(1.0, sigma));
  if (symmetric) *t_min = (float)((double)*t_min + ldexp(1.0, sigma));
}

/* quant_cpu/sim_helper.cpp:14-21 round(a, r, sigma): `a + r` is a float add (r is a float parameter),
 * `- 0.5` promotes to double, nearbyint(double) rounds half-to-even, result narrows to float.
 * sim_helper.cpp:24-38 up_round / down_round use ceil / floor. */
static inline float fixed_q1(float a, int sigma, int rounding, float r) {
  a = ldexpf(a, -sigma);
  if (rounding == R_UP) a = ceilf(a);
  else if (rounding == R_DOWN) a = floorf(a);
  else a = (float)nearbyint((double)(float)(a + r) - 0.5);
  return ldexpf(a, sigma);
}

static inline float clampf(float a, float lo, float hi) { return a > hi ? hi : (a < lo ? lo : a); }

static inline float rnd_unit(uint64_t seed, uint64_t idx) { /* [0,1) with 24 random bits, like rand_like */
  return (float)(rnd_bits(seed, idx) >> 8) * (1.0f / 16777216.0f);
}

/* quant_cpu.cpp:148-167 (+ :127-146, 169-209 for the other modes), fused with the affine wrapper of
 * numerical/cast.py:278-296:  v = x/sc + zp ; q = fixed_point_quantize(v) ; y = (q - zp) * sc, all fp32.
 * x is viewed as [outer, C, inner] (C = size along ch_axis); channel c uses scale[c / group_size]
 * (repeat_interleave(sc, group_size)[:C], cast.py:281-292).  scale == NULL -> no affine (bare Format.cast).
 * Per-tensor: C = 1 (outer*inner = numel), group_size = 1, one scale.  Per-channel: group_size = 1. */
int oracle_fixed_qdq(const float* in, float* out, int64_t outer, int64_t C, int64_t inner, int wl, int fl,
                     int clamp, int symmetric, int rounding, const float* scale, const int64_t* zp,
                     int64_t group_size, uint64_t seed) {
  if (wl < 1 || group_size < 1) return 1;
  float t_min, t_max;
  fixed_min_max(wl, fl, symmetric, &t_min, &t_max);
  const int sigma = -fl;
  const int64_t n = outer * C * inner;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) {
    const int64_t c = (i / inner) % C;
    float v = in[i], sc = 1.0f, z = 0.0f;
    if (scale) {
      sc = scale[c / group_size];
      z = (float)zp[c / group_size];
      v = v / sc + z;
    }
    float r = 0.5f;
    if (rounding == R_STOCHASTIC) r = rnd_unit(seed, (uint64_t)i);
    float q = fixed_q1(v, sigma, rounding, r);
    if (clamp) q = clampf(q, t_min, t_max);
    if (scale) q = (q - z) * sc;
    out[i] = q;
  }
  return 0;
}

static void fixed_min_max(int wl, int fl, int symmetric, float* t_min, float* t_max);
static inline float fixed_q1(float a, int sigma, int rounding, float r);
static inline float clampf(float a, float lo, float hi);
/* ------------------------------------------------------------------------------------------- composite block formats
 * numerical/format.py:453-479 ScaledBlockFloatingPoint.cast on a contiguous fp32 [rows, L] matrix, blocks of B along
 * the last dim (ragged tail = torch.split).  Per block: s = max|x| / (2^(p-1)-1);  where s > 0:
 *   y = fixed_point_quantize(x / s, wl=p, fl=0, clamp, symmetric, nearest) * |float_quantize(s, man, exp, bias, flush)|
 * else y = x.  All fp32 (torch CPU): two IEEE divisions, the CPU fixed rounding, one product. */
int oracle_sbfp_qdq(const float* in, float* out, int64_t rows, int64_t L, int64_t B, int p, int clamp, int symmetric,
                    int man, int exp_bits, int bias, int flush) {
  if (B < 1 || p < 1 || man < 0 || man > 22) return 1;
  float t_min, t_max;
  fixed_min_max(p, 0, symmetric, &t_min, &t_max);
  const float man_scaling = (float)((1 << (p - 1)) - 1);
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < rows; r++)
    for (int64_t b0 = 0; b0 < L; b0 += B) {
      const int64_t len = (L - b0 < B) ? (L - b0) : B;
      const float* x = in + r * L + b0;
      float* y = out + r * L + b0;
      float m = 0.0f;
      for (int64_t i = 0; i < len; i++) { float a = fabsf(x[i]); if (a > m || isnan(a)) m = a; }
      const float s = m / man_scaling;
      if (s > 0.0f) {
        const float sc = fabsf(float_q1(s, man, exp_bits, bias, flush, R_NEAREST, 0u));
        for (int64_t i = 0; i < len; i++) {
          float q = fixed_q1(x[i] / s, 0, R_NEAREST, 0.5f);
          if (clamp) q = clampf(q, t_min, t_max);
          y[i] = q * sc;
        }
      } else {
        for (int64_t i = 0; i < len; i++) y[i] = x[i];
      }
    }
  return 0;
}

/* floor(log2f(m)) as the reference gets it from `torch.floor(torch.log2(chunk_max))` (numerical/format.py:551-553) for a
 * normal finite fp32 m, WITHOUT calling a libm.  log2(m) is irrational unless m is a power of two, and a float32 log2 within
 * an ulp of the truth (Sleef in torch, glibc here, ocml on the device) can only be pushed across an INTEGER when m lies just
 * below a power of two: m = 2^v (1 - j 2^-24).  Then log2 m = v - j 2^-24 / ln 2 (1 + ...), and its float32 rounding is v
 * itself -- so the floor comes out one too high -- exactly when that distance is under half the gap between v and the float32
 * below it:  gap = ulp(|v|), halved when v is a positive power of two.  With g = -log2(gap / 2) that is j <= jmax[g] =
 * floor(2^24 (1 - 2^(-2^-g))).  oracle/gen_golden_r3.py checks this rule against torch.log2 itself for EVERY exponent and every
 * j <= 256 (and 2 M random maxima) and records reference MXFP casts of such blocks (tests/golden/boundaries.npz). */
int oracle_floor_log2f(float m) {
  const uint32_t b = f2u(m) & 0x7FFFFFFFu;
  const int eb = (int)(b >> 23);
  if (eb < 1 || eb > 254) return (int)floorf(log2f(m)); /* zero / denormal / Inf / NaN: not the rule's domain */
  const uint32_t man = b & 0x007FFFFFu;
  int fl = eb - 127;
  if (man != 0u) {
    const int v = fl + 1;
    const uint32_t j = 0x00800000u - man; /* m = 2^v (1 - j 2^-24) */
    if (v != 0) {
      const uint32_t a = (uint32_t)(v < 0 ? -v : v);
      int c = 0;
      while ((a >> (c + 1)) != 0u) c++; /* floor(log2 |v|) */
      const int g = (v > 0 && (a & (a - 1u)) == 0u) ? 25 - c : 24 - c; /* 17 .. 25 */
      static const uint32_t jmax[9] = {88u, 44u, 22u, 11u, 5u, 2u, 1u, 0u, 0u}; /* g = 17 .. 25 */
      if (j <= jmax[g - 17]) fl += 1;
    }
  }
  return fl;
}

/* numerical/format.py:545-564 MXFP.cast (intended layout: blocks along the last dim; the reference's
 * cat(dim=block_dim) slip is not reproduced).  Per block: scale = 2^floor(log2(max|x|)) / 2^(2^(e-1));
 * y = float_quantize(x / scale, man, exp, bias = 2^(e-1)-1, no flush) * scale.
 * An all-zero block gives log2(0) = -inf, scale 0 and NaN in the reference; here it stays zero (documented). */
int oracle_mxfp_qdq(const float* in, float* out, int64_t rows, int64_t L, int64_t B, int man, int exp_bits) {
  if (B < 1 || man < 0 || man > 22 || exp_bits < 1 || exp_bits > 8) return 1;
  const int bias = (1 << (exp_bits - 1)) - 1;
  const float big = ldexpf(1.0f, 1 << (exp_bits - 1));
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < rows; r++)
    for (int64_t b0 = 0; b0 < L; b0 += B) {
      const int64_t len = (L - b0 < B) ? (L - b0) : B;
      const float* x = in + r * L + b0;
      float* y = out + r * L + b0;
      float m = 0.0f;
      for (int64_t i = 0; i < len; i++) { float a = fabsf(x[i]); if (a > m || isnan(a)) m = a; }
      if (m == 0.0f) { for (int64_t i = 0; i < len; i++) y[i] = x[i] * 0.0f; continue; }
      const float scale = powf(2.0f, (float)oracle_floor_log2f(m)) / big;
      for (int64_t i = 0; i < len; i++) y[i] = float_q1(x[i] / scale, man, exp_bits, bias, 0, R_NEAREST, 0u) * scale;
    }
  return 0;
}

/* ------------------------------------------------------------------------------------------- N:M mask
 * sparse.py:163-180 BlockTopK.forward on contiguous groups of M: idx = argsort(score)[:, :M-K]; mask =
 * ones.scatter_(idx, 0).  torch.argsort on CPU behaves as a STABLE ascending sort (pinned by the golden
 * vectors): ties keep index order, NaN sorts last.  sparse.py:300: y = x * mask (a multiply, so a masked
 * negative keeps its sign bit: -1 * 0 = -0.0).  Either of mask/y may be NULL. */
static inline int key_lt(float a, float b) { /* a sorts strictly before b */
  if (isnan(a)) return 0;
  if (isnan(b)) return 1;
  return a < b;
}
int oracle_nm_mask(const float* score, const float* x, float* mask, float* y, int64_t n_groups, int M, int K) {
  if (M < 1 || M > 64 || K < 1 || K > M) return 1;
#pragma omp parallel for schedule(static)
  for (int64_t g = 0; g < n_groups; g++) {
    const float* s = score + g * M;
    int order[64];
    for (int i = 0; i < M; i++) order[i] = i;
    for (int i = 1; i < M; i++) { /* stable insertion sort, ascending */
      int oi = order[i], j = i - 1;
      while (j >= 0 && key_lt(s[oi], s[order[j]])) { order[j + 1] = order[j]; j--; }
      order[j + 1] = oi;
    }
    float mk[64];
    for (int i = 0; i < M; i++) mk[i] = 1.0f;
    for (int i = 0; i < M - K; i++) mk[order[i]] = 0.0f;
    for (int i = 0; i < M; i++) {
      if (mask) mask[g * M + i] = mk[i];
      if (y) y[g * M + i] = x[g * M + i] * mk[i];
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------- reductions
 * numerical/observer.py:173-193 MinMaxObserver.forward (one call, no running state) on slabs: x viewed as
 * [outer, C, inner]; group g covers channels [g*group_size, min((g+1)*group_size, C)) (torch.split along
 * ch_axis, cast.py:200-204).  Writes min/max per group. */
int oracle_group_minmax(const float* in, int64_t outer, int64_t C, int64_t inner, int64_t group_size,
                        float* mn, float* mx) {
  if (group_size < 1) return 1;
  const int64_t G = (C + group_size - 1) / group_size;
  for (int64_t g = 0; g < G; g++) { mn[g] = INFINITY; mx[g] = -INFINITY; }
  for (int64_t o = 0; o < outer; o++)
    for (int64_t c = 0; c < C; c++) {
      const int64_t g = c / group_size;
      const float* p = in + (o * C + c) * inner;
      for (int64_t i = 0; i < inner; i++) {
        /* torch.amin / torch.amax (observer.py:181-182) propagate NaN: one NaN in a group makes both results NaN, for good */
        if (isnan(p[i]) || isnan(mn[g])) { mn[g] = NAN; mx[g] = NAN; continue; }
        if (p[i] < mn[g]) mn[g] = p[i];
        if (p[i] > mx[g]) mx[g] = p[i];
      }
    }
  return 0;
}

/* numerical/observer.py:59-115 DMXObserverBase._calculate_qparams for FixedPoint formats.
 * qmin/qmax from observer.py:13-21: symmetric format -> [-(2^(p-1)-1), 2^(p-1)-1], else [-2^(p-1), 2^(p-1)-1].
 * symmetric qscheme: scale = max(-min_neg, max_pos) / ((qmax-qmin)/2), clamped >= eps, zp = 0.
 * affine qscheme:    scale = (max_pos - min_neg) / (qmax-qmin), clamped >= eps,
 *                    zp = clamp(qmin - round(min_neg/scale), qmin, qmax)  (torch.round = half-to-even). */
int oracle_qparams(const float* mn, const float* mx, int64_t G, int qmin, int qmax, int symmetric_qscheme,
                   float* scale, int64_t* zp) {
  const float eps = 1.1920928955078125e-07f; /* torch.finfo(torch.float32).eps */
  for (int64_t g = 0; g < G; g++) {
    const float min_neg = mn[g] < 0.0f ? mn[g] : 0.0f;
    const float max_pos = mx[g] > 0.0f ? mx[g] : 0.0f;
    if (symmetric_qscheme) {
      float m = (-min_neg > max_pos) ? -min_neg : max_pos;
      float s = m / ((float)(qmax - qmin) / 2.0f);
      scale[g] = s > eps ? s : eps;
      zp[g] = 0;
    } else {
      float s = (max_pos - min_neg) / (float)(qmax - qmin);
      s = s > eps ? s : eps;
      float z = (float)qmin - nearbyintf(min_neg / s);
      z = z < (float)qmin ? (float)qmin : (z > (float)qmax ? (float)qmax : z);
      scale[g] = s;
      zp[g] = (int64_t)z;
    }
  }
  return 0;
}

/* numerical/smoothquant.py:285-299 _maxabs: per-channel max|x| of x viewed as [outer, C, inner]. */
int oracle_channel_maxabs(const float* in, int64_t outer, int64_t C, int64_t inner, float* out) {
  for (int64_t c = 0; c < C; c++) out[c] = 0.0f;
  for (int64_t o = 0; o < outer; o++)
    for (int64_t c = 0; c < C; c++) {
      const float* p = in + (o * C + c) * inner;
      for (int64_t i = 0; i < inner; i++) { float a = fabsf(p[i]); if (a > out[c] || isnan(a)) out[c] = a; }
    }
  return 0;
}

/* torch.histc as numerical/observer.py:470-472,489-491 calls it (ATen 2.10 CPU, aten/src/ATen/native/cpu/
 * HistogramKernel.cpp, the LINEAR_INTERPOLATION binning used by histc -- a third-party dependency of the reference,
 * restated from its behaviour and pinned against torch.histc itself in oracle/gen_golden.py): with lo < hi,
 * elements outside [lo, hi] (and NaN) are dropped, bin = (int64)((x - lo) * bins / (hi - lo)) in fp32, and the
 * right edge belongs to the last bin.  lo == hi means "use the data's own min/max" and, if those are equal too,
 * [lo - 1, hi + 1].  Counts are accumulated in fp32 like ATen's (+1.0f per element). */
int oracle_histc(const float* x, int64_t n, int64_t bins, float lo, float hi, float* hist) {
  if (bins <= 0) return 1;
  for (int64_t b = 0; b < bins; b++) hist[b] = 0.0f;
  if (n == 0) return 0;
  if (lo == hi) {
    lo = hi = x[0];
    for (int64_t i = 1; i < n; i++) { if (x[i] < lo) lo = x[i]; if (x[i] > hi) hi = x[i]; }
    if (lo == hi) { lo -= 1.0f; hi += 1.0f; }
  }
  if (!(lo < hi)) return 1;
  const float width = hi - lo, fb = (float)bins;
  for (int64_t i = 0; i < n; i++) {
    const float v = x[i];
    if (!(v >= lo && v <= hi)) continue;
    int64_t pos = (int64_t)((v - lo) * fb / width);
    if (pos >= bins) pos = bins - 1;
    hist[pos] += 1.0f;
  }
  return 0;
}

/* sparse.py:201-221 Bernoulli.forward with the counter-based stream of rnd_bits in place of torch.bernoulli's global
 * generator: mask = (u < score), u = (rnd >> 8) * 2^-24 in [0, 1). */
int oracle_bernoulli_mask(const float* score, float* mask, int64_t n, uint64_t seed) {
  for (int64_t i = 0; i < n; i++) {
    const float u = (float)(rnd_bits(seed, (uint64_t)i) >> 8) * (1.0f / 16777216.0f);
    mask[i] = u < score[i] ? 1.0f : 0.0f;
  }
  return 0;
}
