// csrc/elementwise.hip — per-element fused quantize->dequantize kernels for gfx950:
//   * low-bit floating point  (numerical/format.py:208-233 -> quant_cpu.cpp:359-402, bit_helper.cpp:4-22)
//   * fixed point + affine     (numerical/format.py:134-142 -> quant_cpu.cpp:127-209, sim_helper.cpp:5-38;
//                               affine wrapper numerical/cast.py:278-296)
//   * per-channel scaling      (numerical/smoothquant.py:255-283)
// All share one skeleton: each lane moves 16 B of input per step (global_load_dwordx4), UNROLL steps in flight,
// fp32 arithmetic, one RNE narrowing to the output dtype, 16-byte stores.  HBM-bound: 2+2 B/elem for 16-bit I/O.
#include "floatq.hpp"
#include "stream.hpp"
#include "lastdim.hpp"
#include "bfp_math.hpp"
#include "unary_ops.hpp"

namespace dmxq {

// ------------------------------------------------------------------------------------------------- float
// (FloatFmt, float_q1 and the per-element cast of the fused modules: floatq.hpp)
// ------------------------------------------------------------------------------------------------- fixed
struct FixedFmt {
  int sigma, clamp, rounding;
  float t_min, t_max;
  uint64_t seed;
};

// sim_helper.cpp:14-21 round(a, r, sigma): ldexp; a1 = (float)(a + r); nearbyint((double)a1 - 0.5) (half-even);
// narrow to float; ldexp.  The fp32 add comes first — that is what makes 0.5 + 2^-24 round to 0 — and the
// double subtraction is exact.  It is reproduced in fp32 only (no f64 VALU, half rate on gfx950):
//   |a1| <  2^23 : a1 - 0.5f is exactly representable, rintf of it is the same integer;
//   |a1| >= 2^23 : a1 is an integer, a1 - 0.5 is an exact tie between a1-1 and a1 -> the even one: a1 unless it
//                  is odd (only possible below 2^24, where the mantissa LSB is the units bit), then a1 - 1.
// sim_helper.cpp:24-38 for up (ceil) / down (floor).
__device__ __forceinline__ float rne_minus_half(float a1) {
  const float mag = fabsf(a1);
  const float small = rintf(a1 - 0.5f);
  const bool odd = (f2u(a1) & 1u) != 0u && mag < 16777216.0f;
  const float big = odd ? a1 - 1.0f : a1;
  return mag >= 8388608.0f ? big : small;
}
__device__ __forceinline__ float fixed_q1(float a, const FixedFmt& f, float r) {
  a = ldexpf(a, -f.sigma);
  if (f.rounding == DMXQ_ROUND_UP) a = ceilf(a);
  else if (f.rounding == DMXQ_ROUND_DOWN) a = floorf(a);
  else a = rne_minus_half(a + r);
  a = ldexpf(a, f.sigma);
  if (f.clamp) a = a > f.t_max ? f.t_max : (a < f.t_min ? f.t_min : a);
  return a;
}

__device__ __forceinline__ float rnd_unit(uint64_t seed, uint64_t idx) {
  return (float)(rnd_bits(seed, idx) >> 8) * (1.0f / 16777216.0f);
}

// ------------------------------------------------------------------------------------------------- ops
// (streaming skeleton, geometry and the channel walker: stream.hpp)
template <int RND>
struct FloatOp {
  static constexpr bool kHeavy = true;  // 13 (nearest) to ~25 VALU ops per element
  // 20-32 MiB tensors (stream.hpp): E4M3 on 3072 / 3584 / 4096 x 4096 bf16: 64 x 8 9.7 / 11.1 / 11.9 us, 256 x 8 9.5 / 12.4 / 12.8; any x 16 spills
  static constexpr int kTileUnroll = 8, kTileThreads = RND == DMXQ_ROUND_NEAREST ? 64 : 256;
  FloatFmt f;
  FloatFast k;
  FlushFast ff;
  __device__ __forceinline__ void apply_one(float x, float& y, int64_t e) const {
    const bool stoch = (RND == kRuntimeRounding) && f.rounding == DMXQ_ROUND_STOCHASTIC;
    y = float_q1<RND>(x, f, stoch ? rnd_bits(f.seed, (uint64_t)e) : 0u);
  }
  template <int N>
  __device__ __forceinline__ void apply_vec(const float (&x)[N], float (&y)[N], int64_t e0) const {
    if (RND == DMXQ_ROUND_NEAREST && ff.usable) {
      // formats that flush their subnormals (FLOAT16 / BFLOAT16 of the BASIC rules): 8 integer operations per element on the magnitude
      // bits (floatq.hpp float_q1_flush); Inf / NaN inputs redone with the bit-level form behind one cold wave-uniform branch
      bool fin = true;
#pragma unroll
      for (int j = 0; j < N; j++) {
        fin = fin && (f2u(x[j]) & 0x7F800000u) != 0x7F800000u;
        y[j] = float_q1_flush(x[j], ff);
      }
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(!fin) != 0ull, 0)) {
#pragma unroll
        for (int j = 0; j < N; j++)
          if ((f2u(x[j]) & 0x7F800000u) == 0x7F800000u) apply_one(x[j], y[j], e0 + j);
      }
      return;
    }
    if (RND == DMXQ_ROUND_NEAREST && k.usable) {
      // the branch-free form for every element, unconditionally; the (rare) elements it does not cover -- Inf, NaN, exponents
      // too large for the magic constant -- are redone with the bit-level form behind ONE cold wave-uniform branch, so that the
      // hot path is straight-line code the scheduler can interleave across the vectors of a tile
      bool ok = true;
#pragma unroll
      for (int j = 0; j < N; j++) {
        ok = ok && float_fast_ok(x[j], k);
        y[j] = float_q1_fast(x[j], k, f.flush != 0, f.unsigned_abs != 0);
      }
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(!ok) != 0ull, 0)) {
#pragma unroll
        for (int j = 0; j < N; j++)
          if (!float_fast_ok(x[j], k)) apply_one(x[j], y[j], e0 + j);
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < N; j++) apply_one(x[j], y[j], e0 + j);
  }
};

// How scale / zero_point are looked up for the N elements of one vector (chosen on the host, see pick_mode):
//   kNone    no affine
//   kTensor  one scale for everything (C == 1)
//   kUniform inner % N == 0: the whole vector lies in ONE channel -> one (32-bit when possible) division pair
//   kLast    inner == 1, group_size == 1, C % N == 0: channel = element index mod C, N consecutive scales
//   kWalk    anything else: ChanIter carries per element
typedef int64_t i64x2 __attribute__((ext_vector_type(2)));
enum ChanMode { kNone = 0, kTensor = 1, kUniform = 2, kLast = 3, kWalk = 4 };

static inline int pick_mode(int64_t C, int64_t inner, int64_t group_size, int epl, const void* scale, const void* zp) {
  if (C <= 1) return kTensor;
  if (inner % epl == 0) return kUniform;
  if (inner == 1 && group_size == 1 && C % epl == 0 && aligned16(scale) && aligned16(zp)) return kLast;
  return kWalk;
}

// SIMPLE: the integer formats of the alias tables (INT8 / INT4: fraction 0, clamped, nearest).  Then ldexp is the
// identity, and the |a1| >= 2^23 branch of rne_minus_half cannot influence the result (such values are clamped to
// t_min / t_max whatever they round to; +-inf likewise; NaN stays NaN through both forms), so the per-element work is
// add, sub, rndne and the clamp.
template <int MODE, bool SIMPLE = false>
struct FixedOp {
  static constexpr bool kHeavy = true;
  // 20-32 MiB tensors (stream.hpp), clamped integer formats, 3072 / 3584 / 4096 x 4096 bf16: no affine 64 x 16 8.5 / 9.9 / 11.1 us (256 x 8
  // 9.9 / 11.8 / 13.7); per-group scale 128 x 16 11.1 / 11.6 / 12.4 us (256 x 8 10.9 / 13.9 / 14.9; 512 x 16 12.2 / 12.9 / 13.7)
#ifndef DMXQ_EXP_FIXED_T
#define DMXQ_EXP_FIXED_T 128
#endif
  static constexpr int kTileUnroll = !SIMPLE ? 4 : 16, kTileThreads = !SIMPLE ? 256 : (MODE == kNone ? 64 : DMXQ_EXP_FIXED_T);
  static constexpr bool kWaitAll = SIMPLE;  // stream.hpp OpWaitAll: the whole tile's data before the first vector's arithmetic (+3.5 % without a scale)
  // common.hpp OpLoadPace (round 5): without a scale, 16 idle issue cycles between a wave's loads on the 64 x 16 tiles: 10.81 -> 10.26 us
  // (pace 4: 10.89, 6: 11.30); with a per-group scale every pace LOSES (11.18 -> 11.42 / 11.81 / 12.21)
  static constexpr int kLoadPace = (SIMPLE && MODE == kNone) ? 2 : 0;
  FixedFmt f;
  ChannelMap cm;
  const float* scale;
  const int64_t* zp;
  // FAST: the caller has checked recip_ok(sc) for this vector and SIMPLE holds (clamped integer format): the quotient
  // comes from div_for_clamped_int (common.hpp), whose final clamped integer equals the IEEE path's for every input
  template <bool FAST = false>
  __device__ __forceinline__ float q(float x, float sc, float z, int64_t e, float rs = 0.0f) const {
    if (MODE != kNone) x = (FAST ? div_for_clamped_int(x, Recip{sc, rs}) : x / sc) + z;  // IEEE division, as torch CPU (cast.py:293)
    float v;
    if (SIMPLE) {
      v = rintf((x + 0.5f) - 0.5f);
      v = v > f.t_max ? f.t_max : (v < f.t_min ? f.t_min : v);
    } else {
      const float r = (f.rounding == DMXQ_ROUND_STOCHASTIC) ? rnd_unit(f.seed, (uint64_t)e) : 0.5f;
      v = fixed_q1(x, f, r);
    }
    if (MODE != kNone) v = (v - z) * sc;
    return v;
  }
  __device__ __forceinline__ void apply_one(float x, float& y, int64_t e) const {
    float sc = 1.0f, z = 0.0f;
    if (MODE != kNone) {
      ChanIter it;
      it.start(cm, e);
      sc = scale[it.g];
      z = (float)zp[it.g];
    }
    y = q(x, sc, z, e);
  }
  // lastdim_kernel interface (per-channel along the contiguous dim): the N channels of a lane stay in registers, as PAIRS for the
  // packed fp32 pipe (scale, reciprocal, zero point: 3 N registers)
  template <int N> struct ChanParams { f32x2 d[N / 2], rs[N / 2], z[N / 2]; bool fast; };
  template <int N> struct RawParams { f32x4 sc[N / 4]; i64x2 zp[N / 2]; };
  // the table reads only (the kernel issues them, then its data loads, and only then make_params: nothing waits on a load before
  // the last one is in flight)
  template <int N>
  __device__ __forceinline__ RawParams<N> fetch_params(int64_t c0) const {
    static_assert(N % 4 == 0, "vectors of 4 or 8 channels");
    RawParams<N> r;
#pragma unroll
    for (int k = 0; k < N / 4; k++) r.sc[k] = *(const f32x4*)(scale + c0 + 4 * k);
#pragma unroll
    for (int k = 0; k < N / 2; k++) r.zp[k] = *(const i64x2*)(zp + c0 + 2 * k);
    return r;
  }
  // (float)(int64): one v_cvt when the value fits 32 bits (every real zero point), the long conversion otherwise
  static __device__ __forceinline__ float zp_float(int64_t v) {
    return (int64_t)(int32_t)v == v ? (float)(int32_t)v : (float)v;
  }
  template <int N>
  __device__ __forceinline__ ChanParams<N> make_params(const RawParams<N>& r) const {
    ChanParams<N> p;
    bool ok = SIMPLE;
#pragma unroll
    for (int k = 0; k < N / 2; k++) {
      const f32x4 t = r.sc[k / 2];
      p.d[k] = (k % 2 == 0) ? (f32x2){t.x, t.y} : (f32x2){t.z, t.w};
      p.z[k] = (f32x2){zp_float(r.zp[k].x), zp_float(r.zp[k].y)};
      // reciprocals of the lane's N channel scales, once per workgroup
      p.rs[k] = (f32x2){1.0f / p.d[k].x, 1.0f / p.d[k].y};
      ok = ok && recip_ok(p.d[k].x) && recip_ok(p.d[k].y);
    }
    p.fast = ok;  // per lane: every one of its N scales has an exact-enough reciprocal
    return p;
  }
  // (lastdim.hpp OpDeferredRedo is NOT taken here: with the flagged rows redone after the stores the hot path has no branch, the
  //  compiler overlaps more rows and the kernel needs 214-224 VGPRs instead of 166 -- two waves per SIMD instead of three, i.e. exactly
  //  as many workgroup slots as a 32 MiB tensor has workgroups, no slack for the dispatcher: 11.9 -> 12.8 us, x / s 12.0 -> 15.5.
  //  The redo stays in place behind one cold wave-uniform branch per row.)
  template <int N>
  __device__ __forceinline__ bool apply_chan_flag(const float (&x)[N], const ChanParams<N>& p, float (&y)[N], int64_t) const {
    // the arithmetic of common.hpp affine_int_pairs with a (scale, reciprocal, zero point) PAIR per instruction: ~7 VALU per element,
    // straight-line.  Lanes holding a scale outside the reciprocal's range, or an Inf / NaN quotient (it shows as a NaN in the SUM of
    // the vector's corrected quotients: affine_int_pairs), are flagged.
    f32x2 acc = {0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < N; k += 2) {
      const f32x2 d = p.d[k / 2], rs = p.rs[k / 2], z = p.z[k / 2];
      const f32x2 n2 = {x[k], x[k + 1]};
      const f32x2 q0 = n2 * rs;
      const f32x2 t = __builtin_elementwise_fma(d, q0, -n2);  // -(r): r = n - d q0, exact
      const f32x2 q = __builtin_elementwise_fma(-t, rs, q0);
      acc = k == 0 ? q : acc + q;
      f32x2 u = q + z;
      u = (u + 0.5f) - 0.5f;
      f32x2 v;
      v.x = __builtin_amdgcn_fmed3f(__builtin_rintf(u.x), f.t_min, f.t_max);
      v.y = __builtin_amdgcn_fmed3f(__builtin_rintf(u.y), f.t_min, f.t_max);
      const f32x2 o = (v - z) * d;
      y[k] = o.x;
      y[k + 1] = o.y;
    }
    return !p.fast || __builtin_amdgcn_classf(acc.x + acc.y, 0x001 | 0x002 | 0x004 | 0x200);
  }
  template <int N>
  __device__ __forceinline__ void apply_chan_exact(const float (&x)[N], const ChanParams<N>& p, float (&y)[N], int64_t e0) const {
#pragma unroll
    for (int k = 0; k < N; k++) y[k] = q(x[k], p.d[k / 2][k % 2], p.z[k / 2][k % 2], e0 + k);
  }
  template <int N>
  __device__ __forceinline__ void apply_chan(const float (&x)[N], const ChanParams<N>& p, float (&y)[N], int64_t e0) const {
    if (SIMPLE) {
      const bool redo = apply_chan_flag(x, p, y, e0);
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(redo) != 0ull, 0)) {
        if (redo) apply_chan_exact(x, p, y, e0);
      }
    } else {
      apply_chan_exact(x, p, y, e0);
    }
  }
  // the vector's (scale, zero point) when it has a single one: fetched ahead of the arithmetic (stream.hpp OpPrep)
  struct Prep { float sc, z, rs; int64_t zraw; };  // rs, and z from zraw: completed by tile_variant (kTileVariants == 3)
  __device__ __forceinline__ Prep prepare(int64_t e0) const {
    Prep p{1.0f, 0.0f, 1.0f, 0};
    if (MODE == kTensor) { p.sc = scale[0]; p.z = (float)zp[0]; }
    if (MODE == kUniform) {
      ChanIter it;
      it.start(cm, e0);
      p.sc = scale[it.g];
      p.z = (float)zp[it.g];
    }
    return p;
  }
  // stream.hpp OpTilePrep: a tile that lies inside one run of a group ((group_size or the last group's remainder) x inner
  // contiguous elements) has a single (scale, zero point)
  static constexpr bool kTilePrep = MODE == kTensor || MODE == kUniform;
  __device__ __forceinline__ bool tile_prepare(int64_t e0, int64_t len, Prep& p) const {
    int64_t g = 0;
    if (MODE == kUniform) {
      if (cm.run_align >= (uint32_t)len) {
        // equal runs, and the (power-of-two) tile divides them (ChannelMap): the group from the tile's first element by two scalar
        // multiply-highs -- round 5; the walker below costs ~30 vector instructions and a readfirstlane ahead of the tile's first load
        const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)e0);
        const uint32_t k = cm.f_run.div(e);
        g = k - cm.f_G.div(k) * cm.G;
      } else {
        ChanIter it;
        it.start(cm, uniform_i64(e0));
        const int64_t left = cm.C - it.g * cm.group_size;
        const int64_t run = (left < cm.group_size ? left : cm.group_size) * cm.inner;
        if (it.r * cm.inner + it.i + len > run) return false;
        g = uniform_i64(it.g);  // (the index arithmetic above ran on the vector unit: back to scalar registers, so that the two table
                                // reads are s_load and not two more vector loads queued in front of the tile's own)
      }
    }
    // The two table reads are REQUESTED here, ahead of the tile's loads, as scalar loads (load_uniform_const: through plain global
    // pointers the compiler issued vector loads and moved the zero point to scalar registers at once -- a full L2 round trip in
    // front of every tile's first load, 11.6 -> 14.0 us on 4096 x 4096 bf16); what consumes them (the int64 -> float conversion
    // included) runs behind the tile's loads, in tile_variant.
    p.sc = load_uniform_const(scale + g);
    if constexpr (kTileVariants == 3) p.zraw = load_uniform_const(zp + g);
    else p.z = (float)load_uniform_const(zp + g);
    return true;
  }
  // stream.hpp OpTileVariants (round 5): for a tile with ONE (scale, zero point) the choice between the reciprocal form and the IEEE
  // division is a scalar compare per TILE, and so is "zero point == 0" (every symmetric scheme): 2 = reciprocal form without the
  // + z / - z steps, 1 = reciprocal form, 0 = the general per-vector code.  Skipping + 0.0f / - 0.0f is exact: q + 0 differs from q
  // only for q = -0 (-> +0), and (u + 0.5f) - 0.5f maps both zeros to +0; v - (+0) == v for every v, -0 included.
  static constexpr int kTileVariants = (SIMPLE && (MODE == kTensor || MODE == kUniform)) ? 3 : 1;
  __device__ __forceinline__ int tile_variant(Prep& p) const {
    p.z = zp_float(p.zraw);
    const float sc = u2f((uint32_t)__builtin_amdgcn_readfirstlane((int)f2u(p.sc)));
    const float z = u2f((uint32_t)__builtin_amdgcn_readfirstlane((int)f2u(p.z)));
    p.rs = 1.0f / sc;  // ONE IEEE division per tile (inside apply_vec_tile it was redone per vector: the tile loop's scheduling fences pin it)
    return recip_ok(sc) ? (z == 0.0f ? 2 : 1) : 0;
  }
  template <int V, int N>
  __device__ __forceinline__ bool apply_vec_tile(const float (&x)[N], float (&y)[N], const Prep& pp) const {
    static_assert(N % 2 == 0, "pairs");
    return affine_int_pairs<N, V == 2>(x, y, pp.sc, pp.rs, pp.z, f.t_min, f.t_max);
  }
  template <int N>
  __device__ __forceinline__ void apply_vec_exact(const float (&x)[N], float (&y)[N], int64_t e0, const Prep& pp) const {
#pragma unroll
    for (int k = 0; k < N; k++) y[k] = q<true>(x[k], pp.sc, pp.z, e0 + k, pp.rs);
  }
  template <int N>
  __device__ __forceinline__ void apply_vec(const float (&x)[N], float (&y)[N], int64_t e0) const {
    apply_vec(x, y, e0, prepare(e0));
  }
  template <int N>
  __device__ __forceinline__ void apply_vec(const float (&x)[N], float (&y)[N], int64_t e0, const Prep& pp) const {
    if (MODE == kNone || MODE == kTensor || MODE == kUniform) {
      const float sc = pp.sc, z = pp.z;
      // one scale for the whole vector: its reciprocal once, then 5 VALU per quotient instead of ~11 (common.hpp)
      // (a two-sided wave-uniform branch here: measured 2 % faster on the same box than the unconditional reciprocal form
      // with a cold IEEE redo -- 15.0 vs 15.3 us, INT8 group_size 128 on 4096 x 4096 bf16 -- the opposite of the BFP kernels)
      if (SIMPLE && MODE != kNone && __builtin_amdgcn_ballot_w64(!recip_ok(sc)) == 0ull) {
        const float rs = 1.0f / sc;
        if constexpr (N % 2 == 0) {
          // pairs through the packed fp32 pipe (common.hpp affine_int_pairs); lanes holding an Inf / NaN quotient redo theirs
          const bool special = affine_int_pairs<N>(x, y, sc, rs, z, f.t_min, f.t_max);
          if (__builtin_expect(__builtin_amdgcn_ballot_w64(special) != 0ull, 0)) {
            if (special) {
#pragma unroll
              for (int k = 0; k < N; k++) y[k] = q<true>(x[k], sc, z, e0 + k, rs);
            }
          }
        } else {
#pragma unroll
          for (int k = 0; k < N; k++) y[k] = q<true>(x[k], sc, z, e0 + k, rs);
        }
      } else {
#pragma unroll
        for (int k = 0; k < N; k++) y[k] = q(x[k], sc, z, e0 + k);
      }
    } else if (MODE == kLast) {
      // c0 is a multiple of N and the tables are 16-byte aligned (pick_mode): N scales = N/4 and N zero points = N/2
      // 16-byte loads instead of 2N scalar ones
      const int64_t c0 = cm.small ? (int64_t)((uint32_t)e0 - cm.f_C.div((uint32_t)e0) * (uint32_t)cm.C) : e0 % cm.C;
      float sc[N], z[N];
#pragma unroll
      for (int k = 0; k < N; k += 4) {
        const f32x4 t = *(const f32x4*)(scale + c0 + k);
        sc[k] = t.x; sc[k + 1] = t.y; sc[k + 2] = t.z; sc[k + 3] = t.w;
      }
#pragma unroll
      for (int k = 0; k < N; k += 2) {
        const i64x2 t = *(const i64x2*)(zp + c0 + k);
        z[k] = (float)t.x; z[k + 1] = (float)t.y;
      }
#pragma unroll
      for (int k = 0; k < N; k++) y[k] = q(x[k], sc[k], z[k], e0 + k);
    } else {
      ChanIter it;
      it.start(cm, e0);
      float sc = scale[it.g], z = (float)zp[it.g];
#pragma unroll
      for (int k = 0; k < N; k++) {
        y[k] = q(x[k], sc, z, e0 + k);
        if (k + 1 < N && it.next(cm)) {  // group boundary inside the vector: reload
          sc = scale[it.g];
          z = (float)zp[it.g];
        }
      }
    }
  }
};

template <bool DIVIDE, int MODE>
struct ScaleOp {
  ChannelMap cm;
  const float* scale;
  // lastdim_kernel, x * s: 16 idle issue cycles between a lane's row loads (common.hpp OpLoadPace: 4096 x 4096 bf16 12.4 -> 11.2 us,
  // the same for 2 .. 12); x / s carries ~54 VALU per row behind its loads and LOSES with any pace (11.9 -> 13.7+)
  static constexpr int kLoadPace = DIVIDE ? 0 : 2;
  // ... and for the WIDENING launch (16-bit -> float32: 8-byte loads, lastdim.hpp IVB == 8) of the division too (round 6, same-lease A/B of
  // three builds, bf16 -> float32 x / s on 4096 x 4096: 19.67 -> 17.50 us with 2, 17.38 with 4 -- 63 -> 70 % -- while the same-width
  // division LOSES with either: 11.55 -> 13.75 / 14.07 us; profiles/r06_ab_stragglers.txt)
  static constexpr int kLoadPaceWide = 2;
  __device__ __forceinline__ void apply_one(float x, float& y, int64_t e) const {
    ChanIter it;
    it.start(cm, e);
    const float s = scale[it.g];
    y = DIVIDE ? x / s : x * s;
  }
  // (DIVIDE: the reciprocals of the lane's N scales too, once per workgroup; `fast`: every one of them has an exact-enough reciprocal)
  template <int N> struct ChanParams { float s[N]; float rs[DIVIDE ? N : 1]; bool fast; };
  template <int N> struct RawParams { f32x4 sc[N / 4]; };
  template <int N>
  __device__ __forceinline__ RawParams<N> fetch_params(int64_t c0) const {
    RawParams<N> r;
#pragma unroll
    for (int k = 0; k < N / 4; k++) r.sc[k] = *(const f32x4*)(scale + c0 + 4 * k);
    return r;
  }
  template <int N>
  __device__ __forceinline__ ChanParams<N> make_params(const RawParams<N>& r) const {
    ChanParams<N> p;
    p.fast = DIVIDE;
#pragma unroll
    for (int k = 0; k < N; k += 4) {
      const f32x4 t = r.sc[k / 4];
      p.s[k] = t.x; p.s[k + 1] = t.y; p.s[k + 2] = t.z; p.s[k + 3] = t.w;
    }
    if (DIVIDE) {
#pragma unroll
      for (int k = 0; k < N; k++) { p.rs[k] = 1.0f / p.s[k]; p.fast = p.fast && recip_ok(p.s[k]); }
    }
    return p;
  }
  // the quotient through the lane's reciprocals (common.hpp div_by_recip: 6 operations instead of an IEEE division's ~13); lanes with
  // a scale or an element outside its range redo theirs with the division behind one cold wave-uniform branch (in place: see FixedOp)
  template <int N>
  __device__ __forceinline__ bool apply_chan_flag(const float (&x)[N], const ChanParams<N>& p, float (&y)[N], int64_t) const {
    bool ok = p.fast;
#pragma unroll
    for (int k = 0; k < N; k++) { y[k] = div_by_recip(x[k], p.s[k], p.rs[k]); ok = ok && div_by_recip_ok(x[k]); }
    return !ok;
  }
  template <int N>
  __device__ __forceinline__ void apply_chan_exact(const float (&x)[N], const ChanParams<N>& p, float (&y)[N], int64_t) const {
#pragma unroll
    for (int k = 0; k < N; k++) y[k] = DIVIDE ? x[k] / p.s[k] : x[k] * p.s[k];
  }
  template <int N>
  __device__ __forceinline__ void apply_chan(const float (&x)[N], const ChanParams<N>& p, float (&y)[N], int64_t e0) const {
    if (DIVIDE) {
      const bool redo = apply_chan_flag(x, p, y, e0);
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(redo) != 0ull, 0)) {
        if (redo) apply_chan_exact(x, p, y, e0);
      }
    } else {
      apply_chan_exact(x, p, y, e0);
    }
  }
  struct Prep { float s; };
  __device__ __forceinline__ Prep prepare(int64_t e0) const {
    Prep p{1.0f};
    if (MODE == kTensor || MODE == kUniform) {
      ChanIter it;
      it.start(cm, e0);
      p.s = scale[it.g];
    }
    return p;
  }
  // stream.hpp OpTilePrep: a tile that lies inside one run of a group ((group_size or the last group's remainder) x inner
  // contiguous elements) has a single scale
  static constexpr bool kTilePrep = MODE == kTensor || MODE == kUniform;
  __device__ __forceinline__ bool tile_prepare(int64_t e0, int64_t len, Prep& p) const {
    int64_t g = 0;
    if (MODE == kUniform) {
      ChanIter it;
      it.start(cm, uniform_i64(e0));
      const int64_t left = cm.C - it.g * cm.group_size;
      const int64_t run = (left < cm.group_size ? left : cm.group_size) * cm.inner;
      if (it.r * cm.inner + it.i + len > run) return false;
      g = uniform_i64(it.g);
    }
    p.s = scale[g];
    return true;
  }
  template <int N>
  __device__ __forceinline__ void apply_vec(const float (&x)[N], float (&y)[N], int64_t e0) const {
    apply_vec(x, y, e0, prepare(e0));
  }
  template <int N>
  __device__ __forceinline__ void apply_vec(const float (&x)[N], float (&y)[N], int64_t e0, const Prep& pp) const {
    if (MODE == kTensor || MODE == kUniform) {
      const float s = pp.s;
#pragma unroll
      for (int k = 0; k < N; k++) y[k] = DIVIDE ? x[k] / s : x[k] * s;
    } else if (MODE == kLast) {
      const int64_t c0 = cm.small ? (int64_t)((uint32_t)e0 - cm.f_C.div((uint32_t)e0) * (uint32_t)cm.C) : e0 % cm.C;
#pragma unroll
      for (int k = 0; k < N; k += 4) {
        const f32x4 t = *(const f32x4*)(scale + c0 + k);  // 16-byte aligned: c0 % N == 0, aligned table (pick_mode)
        y[k] = DIVIDE ? x[k] / t.x : x[k] * t.x;
        y[k + 1] = DIVIDE ? x[k + 1] / t.y : x[k + 1] * t.y;
        y[k + 2] = DIVIDE ? x[k + 2] / t.z : x[k + 2] * t.z;
        y[k + 3] = DIVIDE ? x[k + 3] / t.w : x[k + 3] * t.w;
      }
    } else {
      ChanIter it;
      it.start(cm, e0);
      float s = scale[it.g];
#pragma unroll
      for (int k = 0; k < N; k++) {
        y[k] = DIVIDE ? x[k] / s : x[k] * s;
        if (k + 1 < N && it.next(cm)) s = scale[it.g];
      }
    }
  }
};

// Bernoulli supermask (sparse.py:201-221 torch.bernoulli(score)): mask = 1 with probability score, drawn from the
// counter-based stream of common.hpp (the reference draws from torch's global generator: statistical parity only)
struct BernoulliOp {
  uint64_t seed;
  __device__ __forceinline__ void apply_one(float p, float& y, int64_t e) const {
    y = rnd_unit(seed, (uint64_t)e) < p ? 1.0f : 0.0f;
  }
  template <int N>
  __device__ __forceinline__ void apply_vec(const float (&x)[N], float (&y)[N], int64_t e0) const {
#pragma unroll
    for (int k = 0; k < N; k++) apply_one(x[k], y[k], e0 + k);
  }
};

static inline ChannelMap make_channel_map(int64_t C, int64_t inner, int64_t group_size, int64_t n) {
  const int64_t c = C < 1 ? 1 : C, in = inner < 1 ? 1 : inner;
  const int small = (n < (int64_t)1 << 31 && c < (int64_t)1 << 31 && in < (int64_t)1 << 31) ? 1 : 0;
  ChannelMap m{c, in, group_size, small, make_fastdiv31(in), make_fastdiv31(c), make_fastdiv31(group_size), 0u, 1u, make_fastdiv31(1), make_fastdiv31(1)};
  const int64_t gs = group_size < c ? group_size : c;  // (one group when the group covers the dim)
  if (small && gs >= 1 && c % gs == 0 && gs * in < ((int64_t)1 << 31)) {
    const int64_t run = gs * in;
    m.run_align = (uint32_t)(run & -run);
    m.G = (uint32_t)(c / gs);
    m.f_run = make_fastdiv31(run);
    m.f_G = make_fastdiv31(c / gs);
  }
  return m;
}

}  // namespace dmxq

using namespace dmxq;

// This file is compiled THREE times (build.py: -DDMXQ_EW_PART=1 / 2 / 3), one object per group of entry points, so that the
// template cross products (7 dtype pairs x op flavours x tile geometries) of the float, fixed and scaling ops compile in
// parallel instead of forming one 3.5-minute translation unit.  Without the macro everything is compiled into one object.
#ifndef DMXQ_EW_PART
#define DMXQ_EW_PART 0
#endif
#define DMXQ_EW(P_) (DMXQ_EW_PART == 0 || DMXQ_EW_PART == (P_))

#if DMXQ_EW(1)
// ------------------------------------------------------------------------------------------------- FLOAT16-style casts of bf16 tensors
// The most frequent cast of the BASIC rule set is FP[1|5|10,15](FN) ("FLOAT16") applied to activations that ARE bf16
// (src/dmx/compressor/__init__.py:306-469: every output / elementwise cast).  A bf16 value has 7 mantissa bits, so keeping
// `man >= 7` of them with nearest rounding changes NOTHING in the format's normal range (round_bitwise adds less than the
// dropped field's weight to all-zero dropped bits); what is left of quant_cpu.cpp:359-402 + bit_helper.cpp:4-22 is the range
// handling, and that works on the raw 16-bit words, two per dword, without widening to fp32:
//   exponent below the format's smallest normal (flush_subnormal = 1) -> +0.0;
//   exponent above its largest                                        -> sign | max_val, which rounds (the `.to(bf16)` of
//     cast.py:306) to the bf16 word (max_e + 1) << 7 when man > 7 -- i.e. min_u16(|x| bits, limit) -- Inf and NaN included
//     (the reference reserves no Inf / NaN codes: they saturate too).
// 3.5 VALU operations per element instead of ~18, so the op streams like a copy; tile geometry by size as for BFP (rows_plan).
__device__ __forceinline__ uint32_t relu16_word(uint32_t w, uint32_t inf_m1_2 /* Inf bits - 1 in both halves */) {
  typedef short i16x2 __attribute__((ext_vector_type(2)));
  const u16x2 mag = __builtin_bit_cast(u16x2, w & 0x7FFF7FFFu);
  // a half is zeroed iff it is negative and 0 < mag <= Inf (not -0.0, not NaN):  t = mag - 1 wraps to 0xFFFF for mag = 0
  const u16x2 t = mag - (u16x2){1, 1};
  const u16x2 d = __builtin_elementwise_sub_sat(t, __builtin_bit_cast(u16x2, inf_m1_2));                 // 0 iff 1 <= mag <= Inf
  const u16x2 in_range = __builtin_elementwise_min(d, (u16x2){1, 1}) - (u16x2){1, 1};                    // 0xFFFF iff in range
  const u16x2 neg = __builtin_bit_cast(u16x2, __builtin_bit_cast(i16x2, w) >> (i16x2){15, 15});          // 0xFFFF iff sign set
  return w & ~__builtin_bit_cast(uint32_t, neg & in_range);
}
// RELU: cast_out(relu(cast_in(w))) with r = the input cast's range and ro the output cast's (dmxq_relu_cast); else the cast alone
struct ReluExtra { Range16 ro; uint32_t inf2; };
// RELU: 0 = the cast alone, 1 = cast_out(relu(cast_in(w))), 2 = relu(cast_in(w)) (same format on both sides: no second cast)
template <int RELU>
__device__ __forceinline__ uint32_t range_or_relu(uint32_t w, const Range16& r, const ReluExtra& x) {
  const uint32_t c = range16_word(w, r);
  if (RELU == 0) return c;
  const uint32_t q = relu16_word(c, x.inf2);
  return RELU == 1 ? range16_word(q, x.ro) : q;
}
// Paced load issue on the 512 x 16 tiles (common.hpp pace_issue: 24 idle issue cycles between a wave's loads): the BASIC activation cast
// 11.35 -> 10.45 us on 4096 x 4096 bf16 (74 -> 80 % of the roofline), the ReLU module 10.98 -> 10.45; pace 1: 11.1, 2 / 3 / 4: 10.47 / 10.45 /
// 10.52 (same-lease A/B of five builds, profiles/r05_tune_pace.txt section 6)
constexpr int kRangePace = 3;
template <int T, int U, int RELU = 0>
__global__ __launch_bounds__(T) void float_range_bf16_kernel(const void* __restrict__ in, void* __restrict__ out, int64_t n_vec, Range16 r,
                                                            ReluExtra rx) {
  constexpr int64_t TILE = (int64_t)T * U;
  const int64_t n_tiles = (n_vec + TILE - 1) / TILE;
  const uint32_t lane = threadIdx.x * 16u;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    // workgroup-uniform tile bases + 32-bit lane offsets, no predicates on full tiles (the schedule of bfp_rows.hpp)
    const char* src = (const char*)in + tile * (TILE * 16);
    char* dst = (char*)out + tile * (TILE * 16);
    if ((tile + 1) * TILE <= n_vec) {
      u32x4 raw[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        raw[u] = load_raw16<true>(src + u * (T * 16), lane);
        if (U >= 16 && u + 1 < U) pace_issue<kRangePace>();
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < U; u++) {
#pragma unroll
        for (int j = 0; j < 4; j++) raw[u][j] = range_or_relu<RELU>(raw[u][j], r, rx);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < U; u++) __builtin_nontemporal_store(raw[u], (u32x4*)(dst + u * (T * 16) + lane));
    } else {
      for (int u = 0; u < U; u++) {
        const int64_t v = tile * TILE + (int64_t)u * T + threadIdx.x;
        if (v < n_vec) {
          u32x4 w = load_raw16<true>(src + u * (T * 16), lane);
#pragma unroll
          for (int j = 0; j < 4; j++) w[j] = range_or_relu<RELU>(w[j], r, rx);
          __builtin_nontemporal_store(w, (u32x4*)(dst + u * (T * 16) + lane));
        }
      }
    }
  }
}

template <int RELU = 0>
static int launch_float_range_bf16(const void* in, void* out, int64_t n_vec, const Range16& r, hipStream_t s, const ReluExtra& rx = ReluExtra{}) {
  const RowsPlan pl = rows_plan(n_vec, true);
  const unsigned grid = (unsigned)(pl.tiles < (1 << 20) ? pl.tiles : (1 << 20));
#define DMXQ_RG(T_, U_) DMXQ_LAUNCH((float_range_bf16_kernel<T_, U_, RELU>), dim3(grid), dim3(T_), 0, s, in, out, n_vec, r, rx)
  switch (pl.id) {
    case 0: DMXQ_RG(512, 1); break;
    case 1: DMXQ_RG(128, 2); break;
    case 2: DMXQ_RG(512, 4); break;
    case 3: DMXQ_RG(512, 16); break;  // (the 16-20 MiB class of rows_plan: the copy-like range kernel keeps one geometry up to 32 MiB)
    case 4: DMXQ_RG(512, 16); break;
    default: DMXQ_RG(512, 2); break;
  }
#undef DMXQ_RG
  return launch_status();
}

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
  // 16-bit -> the same 16-bit dtype, nearest, at least the dtype's own mantissa bits kept, subnormals flushed, signed: the packed
  // range-only kernel (common.hpp range16_of decides and builds the two thresholds)
  if (dtype_in == dtype_out && dtype_in != DMXQ_F32 && rounding == DMXQ_ROUND_NEAREST && !unsigned_abs && n % 8 == 0 && aligned16(in) &&
      aligned16(out)) {
    const dmxq_float_fmt ff{man_bits, exp_bits, exp_bias, flush_subnormal ? 1 : 0};
    Range16 r;
    if (range16_of(&ff, dtype_in, &r)) return launch_float_range_bf16(in, out, n / 8, r, s);
  }
  // (the integer flush form does not apply the final abs of sign-less formats: those keep the magic-add form)
  const FlushFast ff = make_flush_fast(f.man, f.exp_bits, f.bias, f.flush && !f.unsigned_abs);
  if (rounding == DMXQ_ROUND_NEAREST) return dispatch_stream(in, out, dtype_in, dtype_out, n, FloatOp<DMXQ_ROUND_NEAREST>{f, make_float_fast(f.man, f.exp_bits, f.bias), ff}, s);
  return dispatch_stream(in, out, dtype_in, dtype_out, n, FloatOp<kRuntimeRounding>{f, make_float_fast(f.man, f.exp_bits, f.bias), FlushFast{}}, s);
}


// ---------------------------------------------------------------------------------------------------------------------------
// Binary module in one pass (dmxq_binary_cast): out = cast_out(cast_a(a) (+|*) cast_b(b)) on bf16 tensors whose casts are
// range-only (the FLOAT16-style formats of the BASIC rules: range16_word above).  Replaces the four launches of a ResAdd / Mul
// DmxModule (modeling/nn/core.py:228-264: two input casts, the torch op, the output cast): 6 B/element instead of 18.
// The op itself is torch's: fp32 arithmetic on the widened operands, one RNE rounding to bf16.
// The BFP cast of the module that consumes the result (dmxq_binary_cast_bfp / dmxq_relu_cast_bfp), on one lane-vector of the module's
// result in place: blocks of lpb adjacent lanes (DPP maximum; EVERY lane of the wave must call this), the arithmetic of bfp_math.hpp --
// symmetric, nearest, the fast block path with the literal one behind a wave vote.
template <int EPL>
__device__ __forceinline__ void bfp_cast_lane_vector(float (&c)[EPL], int lpb, int wl) {
  uint32_t mb = 0u;
#pragma unroll
  for (int j = 0; j < EPL; j++) mb = max(mb, f2u(c[j]) & 0x7FFFFFFFu);
  mb = group_max_u32(mb, lpb);
  const bool fast_ok = bfp_fast_ok(mb, wl);
  float q[EPL];
  {
    const BfpBlockParams bp = bfp_block_params<false, true>(mb, wl);
#pragma unroll
    for (int j = 0; j < EPL; j++) q[j] = bfp_q1_fast<false, false>(c[j], bp);
  }
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(!fast_ok) != 0ull, 0)) {
    if (!fast_ok) {
      const BfpBlockParams bp = bfp_block_params<false, false>(mb, wl);
#pragma unroll
      for (int j = 0; j < EPL; j++) q[j] = bfp_q1<DMXQ_ROUND_NEAREST, false>(c[j], bp, wl, DMXQ_ROUND_NEAREST, 0u);
    }
  }
#pragma unroll
  for (int j = 0; j < EPL; j++) c[j] = q[j];
}

// OP 0 add, 1 mul, 2 relu (b unused; only with BFPOUT: the plain ReLU module is float_range_bf16_kernel<., ., 1>)
struct BinArgs { const void* a; const void* b; void* out; int64_t n_vec; Range16 ra, rb, ro; int bfp_lpb = 0, bfp_wl = 0; };
template <int DT, int OP, int T, int U, bool BFPOUT = false>
__global__ __launch_bounds__(T) void binary_range_bf16_kernel(const BinArgs g) {
  constexpr int64_t TILE = (int64_t)T * U;
  const int64_t base = (int64_t)blockIdx.x * TILE + threadIdx.x;
  u32x4 ra[U], rb[OP == 2 ? 1 : U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    const int64_t v = base + u * T < g.n_vec ? base + u * T : g.n_vec - 1;  // clamped: unconditional loads
    ra[u] = load_raw16<true>(g.a, v * 16);
    if (OP != 2) rb[u] = load_raw16<true>(g.b, v * 16);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int u = 0; u < U; u++) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint32_t wa = range16_word(ra[u][j], g.ra), wb = OP == 2 ? 0u : range16_word(rb[OP == 2 ? 0 : u][j], g.rb);
      float a0, a1, b0, b1;
      if (DT == DMXQ_BF16) { a0 = u2f(wa << 16); a1 = u2f(wa & 0xFFFF0000u); b0 = u2f(wb << 16); b1 = u2f(wb & 0xFFFF0000u); }
      else { a0 = half_lo(wa); a1 = half_hi(wa); b0 = half_lo(wb); b1 = half_hi(wb); }
      float c0, c1;
      if (OP == 2) {  // clamp_min(x, 0): -0.0 and NaN pass with their bits
        ra[u][j] = range16_word((a0 < 0.0f ? 0u : (wa & 0xFFFFu)) | (a1 < 0.0f ? 0u : (wa & 0xFFFF0000u)), g.ro);
      } else {
        c0 = OP == 0 ? a0 + b0 : a0 * b0, c1 = OP == 0 ? a1 + b1 : a1 * b1;
        if (DT == DMXQ_BF16) { c0 = c0 != c0 ? u2f(0x7FC00000u) : c0; c1 = c1 != c1 ? u2f(0x7FC00000u) : c1; }  // c10::BFloat16: every NaN -> +0x7FC0
        ra[u][j] = range16_word(pack2<DT>(c0, c1), g.ro);
      }
    }
    if constexpr (BFPOUT) {
      float c[8];
      widen<DT, 8>(ra[u], c);
      bfp_cast_lane_vector<8>(c, __builtin_amdgcn_readfirstlane(g.bfp_lpb), __builtin_amdgcn_readfirstlane(g.bfp_wl));
      const OutVec<DT, 8> o = pack_vec<DT, 8>(c);
#pragma unroll
      for (int j = 0; j < 4; j++) ra[u][j] = o.w[j];
    }
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int u = 0; u < U; u++) {
    const int64_t v = base + u * T;
    if (v < g.n_vec) __builtin_nontemporal_store(ra[u], (u32x4*)((char*)g.out + v * 16));
  }
}

// The general form of the fused modules: ANY nearest-rounding FloatingPoint cast (incl. ones that round: FLOAT16 on a float32
// tensor, 8-bit floats on a bf16 one) on ANY tensor dtype.  Per element: widen, cast_a / cast_b (magic-add form of floatq.hpp with
// the bit-level form for what it does not cover), CastTo's `.to(dtype)`, the op in fp32 rounded once to the dtype, cast_out and
// its `.to(dtype)`.  OP 0 add, 1 mul, 2 relu (b unused).  3 casts ~ 50 VALU per element: still under the memory time of a float32
// tensor (12 B/element), about level with it for 16-bit ones.
struct GenArgs { const void* a; const void* b; void* out; int64_t n_vec; CastG ca, cb, co; int bfp_lpb = 0, bfp_wl = 0; };
// BFPOUT (dmxq_binary_cast_bfp / dmxq_relu_cast_bfp): the input cast of the module that consumes the result -- BFP blocks of lpb adjacent
// lane-vectors along the contiguous rows -- applied to the module's result before it is stored: every lane of the kernel already runs
// unconditionally (clamped loads, predicated stores), so the DPP block maximum needs nothing else; arithmetic of bfp_math.hpp.
template <int DT, int OP, int T, int U, bool BFPOUT = false>
__global__ __launch_bounds__(T) void fused_cast_generic_kernel(const GenArgs g) {
  constexpr int EPL = 16 / Elem<DT>::bytes;
  const int64_t base = (int64_t)blockIdx.x * ((int64_t)T * U) + threadIdx.x;
  u32x4 ra[U], rb[OP == 2 ? 1 : U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    const int64_t v = base + u * T < g.n_vec ? base + u * T : g.n_vec - 1;  // clamped: unconditional loads
    ra[u] = load_raw16<true>(g.a, v * 16);
    if (OP != 2) rb[u] = load_raw16<true>(g.b, v * 16);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int u = 0; u < U; u++) {
    float x[EPL], y[EPL], c[EPL];
    widen<DT, EPL>(ra[u], x);
    castg_vec<DT, EPL>(x, g.ca);
    if (OP != 2) {
      widen<DT, EPL>(rb[OP == 2 ? 0 : u], y);
      castg_vec<DT, EPL>(y, g.cb);
    }
#pragma unroll
    for (int j = 0; j < EPL; j++)
      c[j] = castg_dt<DT>(OP == 0 ? x[j] + y[j] : (OP == 1 ? x[j] * y[j] : (x[j] < 0.0f ? 0.0f : x[j])));  // clamp_min(x, 0): -0.0 and NaN pass
    castg_vec<DT, EPL>(c, g.co);
    if constexpr (BFPOUT) {
#pragma unroll
      for (int j = 0; j < EPL; j++) c[j] = castg_dt<DT>(c[j]);  // the module's result, in its dtype
      bfp_cast_lane_vector<EPL>(c, __builtin_amdgcn_readfirstlane(g.bfp_lpb), __builtin_amdgcn_readfirstlane(g.bfp_wl));
    }
    const OutVec<DT, EPL> o = pack_vec<DT, EPL>(c);
#pragma unroll
    for (int j = 0; j < 4; j++) ra[u][j] = o.w[j];
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int u = 0; u < U; u++)
    if (base + u * T < g.n_vec) __builtin_nontemporal_store(ra[u], (u32x4*)((char*)g.out + (base + u * T) * 16));
}
// bfp_block > 0: rows of `row_len` elements in whole blocks of bfp_block (the consumer's BFP input cast, symmetric, nearest)
template <int OP>
static int launch_fused_generic(const void* a, const void* b, void* out, int dtype, int64_t n, const dmxq_float_fmt* fa, const dmxq_float_fmt* fb,
                                const dmxq_float_fmt* fo, hipStream_t s, int64_t row_len = 0, int64_t bfp_block = 0, int bfp_precision = 0) {
  const int epl = dtype == DMXQ_F32 ? 4 : 8;
  GenArgs g{a, b, out, n / epl, {}, {}, {}};
  if (n % epl != 0 || !aligned16(a) || (OP != 2 && !aligned16(b)) || !aligned16(out) || !castg_of(fa, &g.ca) || !castg_of(fb, &g.cb) || !castg_of(fo, &g.co))
    return DMXQ_ERR_UNSUPPORTED;
  constexpr int T = 256, U = 2;
  const int64_t tiles = (g.n_vec + T * U - 1) / (T * U);
  if (tiles > 0x7FFFFFFF) return DMXQ_ERR_UNSUPPORTED;
  if (bfp_block > 0) {
    const int64_t lpb = bfp_block / epl;
    if (row_len < 1 || row_len % bfp_block != 0 || n % row_len != 0 || bfp_block % epl != 0 || lpb < 1 || lpb > 64 || (lpb & (lpb - 1)) != 0 ||
        bfp_precision < 2 || bfp_precision > 20)
      return DMXQ_ERR_UNSUPPORTED;
    g.bfp_lpb = (int)lpb;
    g.bfp_wl = bfp_precision;
    if (dtype == DMXQ_F32) DMXQ_LAUNCH((fused_cast_generic_kernel<DMXQ_F32, OP, T, U, true>), dim3((unsigned)tiles), dim3(T), 0, s, g);
    else if (dtype == DMXQ_F16) DMXQ_LAUNCH((fused_cast_generic_kernel<DMXQ_F16, OP, T, U, true>), dim3((unsigned)tiles), dim3(T), 0, s, g);
    else DMXQ_LAUNCH((fused_cast_generic_kernel<DMXQ_BF16, OP, T, U, true>), dim3((unsigned)tiles), dim3(T), 0, s, g);
    return launch_status();
  }
  if (dtype == DMXQ_F32) DMXQ_LAUNCH((fused_cast_generic_kernel<DMXQ_F32, OP, T, U>), dim3((unsigned)tiles), dim3(T), 0, s, g);
  else if (dtype == DMXQ_F16) DMXQ_LAUNCH((fused_cast_generic_kernel<DMXQ_F16, OP, T, U>), dim3((unsigned)tiles), dim3(T), 0, s, g);
  else DMXQ_LAUNCH((fused_cast_generic_kernel<DMXQ_BF16, OP, T, U>), dim3((unsigned)tiles), dim3(T), 0, s, g);
  return launch_status();
}

extern "C" int dmxq_binary_cast(const void* a, const void* b, void* out, int dtype, int64_t n, int op, const dmxq_float_fmt* cast_a,
                                const dmxq_float_fmt* cast_b, const dmxq_float_fmt* cast_out, void* stream) {
  if (!valid_dtype(dtype) || n < 0 || (op != DMXQ_BINARY_ADD && op != DMXQ_BINARY_MUL)) return DMXQ_ERR_BAD_ARG;
  if (n == 0) return DMXQ_OK;
  if (!a || !b || !out) return DMXQ_ERR_BAD_ARG;
  BinArgs g{a, b, out, n / 8, {}, {}, {}};
  if (n % 8 != 0 || !aligned16(a) || !aligned16(b) || !aligned16(out) || !range16_of(cast_a, dtype, &g.ra) ||
      !range16_of(cast_b, dtype, &g.rb) || !range16_of(cast_out, dtype, &g.ro))  // not range-only: the general form
    return op == DMXQ_BINARY_ADD ? launch_fused_generic<0>(a, b, out, dtype, n, cast_a, cast_b, cast_out, (hipStream_t)stream)
                                 : launch_fused_generic<1>(a, b, out, dtype, n, cast_a, cast_b, cast_out, (hipStream_t)stream);
  // measured: 256x2 17.9 us, 256x4 18.3, 512x8 19.3 (three streams: small tiles interleave best); round 5, with paced loads (common.hpp
  // pace_issue): 256x2 17.45, 256x8 pace 0 / 2 / 4 18.0 / 17.8 / 17.9, 512x8 pace 2 18.1, 256x4 pace 3 17.5 -- no deep tile catches up
  constexpr int T = 256, U = 2;
  const int64_t tiles = (g.n_vec + T * U - 1) / (T * U);
  if (tiles > 0x7FFFFFFF) return DMXQ_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == DMXQ_BF16) {
    if (op == DMXQ_BINARY_ADD) DMXQ_LAUNCH((binary_range_bf16_kernel<DMXQ_BF16, 0, T, U>), dim3((unsigned)tiles), dim3(T), 0, s, g);
    else DMXQ_LAUNCH((binary_range_bf16_kernel<DMXQ_BF16, 1, T, U>), dim3((unsigned)tiles), dim3(T), 0, s, g);
  } else {
    if (op == DMXQ_BINARY_ADD) DMXQ_LAUNCH((binary_range_bf16_kernel<DMXQ_F16, 0, T, U>), dim3((unsigned)tiles), dim3(T), 0, s, g);
    else DMXQ_LAUNCH((binary_range_bf16_kernel<DMXQ_F16, 1, T, U>), dim3((unsigned)tiles), dim3(T), 0, s, g);
  }
  return launch_status();
}


// A ReLU DmxModule in one pass (dmxq_relu_cast): out = cast_out(relu(cast_in(x))) on 16-bit tensors with range-only casts, all on
// the packed words.  relu = at::clamp_min(x, 0): negative values (not -0.0, not NaN) become +0.0, everything else passes.
extern "C" int dmxq_relu_cast(const void* in, void* out, int dtype, int64_t n, const dmxq_float_fmt* cast_in, const dmxq_float_fmt* cast_out,
                              void* stream) {
  if (!valid_dtype(dtype) || n < 0) return DMXQ_ERR_BAD_ARG;
  if (n == 0) return DMXQ_OK;
  if (!in || !out) return DMXQ_ERR_BAD_ARG;
  Range16 ri;
  ReluExtra rx{{}, dtype == DMXQ_BF16 ? 0x7F7F7F7Fu : 0x7BFF7BFFu};
  if (n % 8 != 0 || !aligned16(in) || !aligned16(out) || !range16_of(cast_in, dtype, &ri) || !range16_of(cast_out, dtype, &rx.ro))
    return launch_fused_generic<2>(in, nullptr, out, dtype, n, cast_in, nullptr, cast_out, (hipStream_t)stream);  // the general form
  // the same format on both sides (the BASIC rules): relu of an already cast value needs no second cast -- an in-range value stays
  // in range, zeros and the clamped NaN / Inf pass relu unchanged or become +0
  if (rx.ro.limit2 == ri.limit2 && rx.ro.minb2 == ri.minb2) return launch_float_range_bf16<2>(in, out, n / 8, ri, (hipStream_t)stream, rx);
  return launch_float_range_bf16<1>(in, out, n / 8, ri, (hipStream_t)stream, rx);  // the tile plans of the plain range cast
}

// 16-bit tensors whose casts are all range-only: the packed-word kernel with the BFP epilogue; kNotRangeOnly = take the general form
constexpr int kNotRangeOnly = -1000;
static int launch_range_bfp(int op, const void* a, const void* b, void* out, int dtype, int64_t n, const dmxq_float_fmt* cast_a, const dmxq_float_fmt* cast_b,
                            const dmxq_float_fmt* cast_out, int64_t row_len, int64_t block_size, int precision, hipStream_t s) {
  BinArgs g{a, b, out, n / 8, {}, {}, {}};
  if (n % 8 != 0 || !aligned16(a) || !aligned16(b) || !aligned16(out) || !range16_of(cast_a, dtype, &g.ra) || !range16_of(cast_b, dtype, &g.rb) ||
      !range16_of(cast_out, dtype, &g.ro))
    return kNotRangeOnly;
  const int64_t lpb = block_size / 8;
  if (row_len < 1 || row_len % block_size != 0 || n % row_len != 0 || block_size % 8 != 0 || lpb < 1 || lpb > 64 || (lpb & (lpb - 1)) != 0 ||
      precision < 2 || precision > 20)
    return DMXQ_ERR_UNSUPPORTED;
  g.bfp_lpb = (int)lpb;
  g.bfp_wl = precision;
  constexpr int T = 256, U = 2;
  const int64_t tiles = (g.n_vec + T * U - 1) / (T * U);
  if (tiles > 0x7FFFFFFF) return DMXQ_ERR_UNSUPPORTED;
#define DMXQ_RB(DT_, OP_) DMXQ_LAUNCH((binary_range_bf16_kernel<DT_, OP_, T, U, true>), dim3((unsigned)tiles), dim3(T), 0, s, g)
  if (dtype == DMXQ_BF16) { if (op == 0) DMXQ_RB(DMXQ_BF16, 0); else if (op == 1) DMXQ_RB(DMXQ_BF16, 1); else DMXQ_RB(DMXQ_BF16, 2); }
  else { if (op == 0) DMXQ_RB(DMXQ_F16, 0); else if (op == 1) DMXQ_RB(DMXQ_F16, 1); else DMXQ_RB(DMXQ_F16, 2); }
#undef DMXQ_RB
  return launch_status();
}

// The same modules followed by the BFP input cast of the ONE module that consumes the result (Mul -> the down projection, ReLU -> fc2:
// `input_casts` of modeling/nn/core.py:228-264; blocks of block_size along rows of row_len elements, symmetric, nearest) in one launch:
// out = BFP_QDQ(module(x)), bit-identical to dmxq_binary_cast / dmxq_relu_cast followed by dmxq_bfp_qdq.  16-bit tensors with range-only casts on the packed words, anything else in the general form.
extern "C" int dmxq_binary_cast_bfp(const void* a, const void* b, void* out, int dtype, int64_t n, int op, const dmxq_float_fmt* cast_a,
                                    const dmxq_float_fmt* cast_b, const dmxq_float_fmt* cast_out, int64_t row_len, int64_t block_size,
                                    int precision, void* stream) {
  if (!valid_dtype(dtype) || n < 0 || (op != DMXQ_BINARY_ADD && op != DMXQ_BINARY_MUL) || block_size < 1) return DMXQ_ERR_BAD_ARG;
  if (n == 0) return DMXQ_OK;
  if (!a || !b || !out) return DMXQ_ERR_BAD_ARG;
  if (dtype != DMXQ_F32) {
    const int rc = launch_range_bfp(op == DMXQ_BINARY_ADD ? 0 : 1, a, b, out, dtype, n, cast_a, cast_b, cast_out, row_len, block_size, precision, (hipStream_t)stream);
    if (rc != kNotRangeOnly) return rc;
  }
  return op == DMXQ_BINARY_ADD ? launch_fused_generic<0>(a, b, out, dtype, n, cast_a, cast_b, cast_out, (hipStream_t)stream, row_len, block_size, precision)
                               : launch_fused_generic<1>(a, b, out, dtype, n, cast_a, cast_b, cast_out, (hipStream_t)stream, row_len, block_size, precision);
}
extern "C" int dmxq_relu_cast_bfp(const void* in, void* out, int dtype, int64_t n, const dmxq_float_fmt* cast_in, const dmxq_float_fmt* cast_out,
                                  int64_t row_len, int64_t block_size, int precision, void* stream) {
  if (!valid_dtype(dtype) || n < 0 || block_size < 1) return DMXQ_ERR_BAD_ARG;
  if (n == 0) return DMXQ_OK;
  if (!in || !out) return DMXQ_ERR_BAD_ARG;
  if (dtype != DMXQ_F32) {
    const int rc = launch_range_bfp(2, in, in, out, dtype, n, cast_in, nullptr, cast_out, row_len, block_size, precision, (hipStream_t)stream);
    if (rc != kNotRangeOnly) return rc;
  }
  return launch_fused_generic<2>(in, nullptr, out, dtype, n, cast_in, nullptr, cast_out, (hipStream_t)stream, row_len, block_size, precision);
}

#endif  // part 1a
#if DMXQ_EW(2)
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
  const ChannelMap cm = make_channel_map(C, inner, group_size, n);
  hipStream_t s = (hipStream_t)stream;
  const bool simple = fraction == 0 && clamp && rounding == DMXQ_ROUND_NEAREST && precision <= 22;  // |t| <= 2^21
  const int mode = scale ? pick_mode(C, inner, group_size, dtype_in == DMXQ_F32 ? 4 : 8, scale, zero_point) : kNone;
  const float* sc_ = scale;
  const int64_t* zp_ = zero_point;
  if (mode == kLast) {  // per-channel along the contiguous dim: parameters in registers (lastdim_kernel)
    const int rc = simple ? launch_lastdim(in, out, dtype_in, dtype_out, outer, C, FixedOp<kLast, true>{f, cm, sc_, zp_}, s)
                          : launch_lastdim(in, out, dtype_in, dtype_out, outer, C, FixedOp<kLast, false>{f, cm, sc_, zp_}, s);
    if (rc != DMXQ_ERR_UNSUPPORTED) return rc;
  }
#define DMXQ_FIX(M_, S_) return dispatch_stream(in, out, dtype_in, dtype_out, n, FixedOp<M_, S_>{f, cm, sc_, zp_}, s)
  if (simple) {
    switch (mode) {
      case kNone: DMXQ_FIX(kNone, true);
      case kTensor: DMXQ_FIX(kTensor, true);
      case kUniform: DMXQ_FIX(kUniform, true);
      default: break;  // the rarer lookup modes share the general build
    }
  }
  switch (mode) {
    case kNone: DMXQ_FIX(kNone, false);
    case kTensor: DMXQ_FIX(kTensor, false);
    case kUniform: DMXQ_FIX(kUniform, false);
    case kLast: DMXQ_FIX(kLast, false);
    default: DMXQ_FIX(kWalk, false);
  }
#undef DMXQ_FIX
}

#endif  // part 2
#if DMXQ_EW(3)
extern "C" int dmxq_scale_channels(const void* in, void* out, int dtype_in, int dtype_out, int64_t outer, int64_t C,
                                   int64_t inner, const float* scale, int divide, void* stream) {
  if (!valid_dtype(dtype_in) || !valid_dtype(dtype_out) || outer < 0 || C < 0 || inner < 0) return DMXQ_ERR_BAD_ARG;
  const int64_t n = outer * C * inner;
  if (n == 0) return DMXQ_OK;
  if (!in || !out || !scale) return DMXQ_ERR_BAD_ARG;
  const ChannelMap cm = make_channel_map(C, inner, 1, n);
  hipStream_t s = (hipStream_t)stream;
  const int mode = pick_mode(C, inner, 1, dtype_in == DMXQ_F32 ? 4 : 8, scale, nullptr);
  if (mode == kLast) {
    const int rc = divide ? launch_lastdim(in, out, dtype_in, dtype_out, outer, C, ScaleOp<true, kLast>{cm, scale}, s)
                          : launch_lastdim(in, out, dtype_in, dtype_out, outer, C, ScaleOp<false, kLast>{cm, scale}, s);
    if (rc != DMXQ_ERR_UNSUPPORTED) return rc;
  }
#define DMXQ_SCALE(D_, M_) return dispatch_stream(in, out, dtype_in, dtype_out, n, ScaleOp<D_, M_>{cm, scale}, s)
  if (divide) {
    if (mode == kTensor || mode == kUniform) DMXQ_SCALE(true, kUniform);
    if (mode == kLast) DMXQ_SCALE(true, kLast);
    DMXQ_SCALE(true, kWalk);
  }
  if (mode == kTensor || mode == kUniform) DMXQ_SCALE(false, kUniform);
  if (mode == kLast) DMXQ_SCALE(false, kLast);
  DMXQ_SCALE(false, kWalk);
#undef DMXQ_SCALE
}

#endif  // part 3
#if DMXQ_EW(1)
extern "C" int dmxq_gelu(const void* in, void* out, int dtype_in, int dtype_out, int64_t n, int tanh_form,
                         void* stream) {
  if (!valid_dtype(dtype_in) || !valid_dtype(dtype_out) || n < 0) return DMXQ_ERR_BAD_ARG;
  if (n == 0) return DMXQ_OK;
  if (!in || !out) return DMXQ_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (dtype_out != DMXQ_F32)
    return tanh_form ? dispatch_stream(in, out, dtype_in, dtype_out, n, GeluOp<true, true>{}, s) : dispatch_stream(in, out, dtype_in, dtype_out, n, GeluOp<true, false>{}, s);
  return tanh_form ? dispatch_stream(in, out, dtype_in, dtype_out, n, GeluOp<false, true>{}, s) : dispatch_stream(in, out, dtype_in, dtype_out, n, GeluOp<false, false>{}, s);
}

extern "C" int dmxq_bernoulli_mask(const void* score, void* mask_out, int dtype_score, int dtype_mask, int64_t n,
                                   uint64_t seed, void* stream) {
  if (!valid_dtype(dtype_score) || !valid_dtype(dtype_mask) || n < 0) return DMXQ_ERR_BAD_ARG;
  if (n == 0) return DMXQ_OK;
  if (!score || !mask_out) return DMXQ_ERR_BAD_ARG;
  return dispatch_stream(score, mask_out, dtype_score, dtype_mask, n, BernoulliOp{seed}, (hipStream_t)stream);
}

// Test hook (not part of include/dmxq.h): out[r, c] = div_for_clamped_int(n[r, c], 1 / d[c]) -- lets the test suite
// check the "equals the IEEE quotient inside the stated operand range" claim of common.hpp directly, bit for bit.
namespace dmxq {
__global__ void div_selftest_kernel(const float* n, const float* d, float* out, int64_t rows, int64_t cols) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < rows * cols) out[i] = div_for_clamped_int(n[i], make_recip(d[i % cols]));
}
}  // namespace dmxq
extern "C" int dmxq_internal_div_selftest(const float* n, const float* d, float* out, int64_t rows, int64_t cols, void* stream) {
  if (!n || !d || !out || rows < 0 || cols < 1) return DMXQ_ERR_BAD_ARG;
  if (rows == 0) return DMXQ_OK;
  DMXQ_LAUNCH(dmxq::div_selftest_kernel, dim3((unsigned)((rows * cols + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, d, out, rows, cols);
  return launch_status();
}

extern "C" const char* dmxq_status_string(int status) {
  switch (status) {
    case DMXQ_OK: return "ok";
    case DMXQ_ERR_BAD_ARG: return "bad argument";
    case DMXQ_ERR_UNSUPPORTED: return "unsupported parameter (undefined behaviour in the reference)";
    case DMXQ_ERR_LAUNCH: return "HIP kernel launch failed";
    case DMXQ_ERR_PENDING: return "a pre-existing HIP error was pending on this thread before the call: launches not verified";
  }
  return "unknown status";
}

extern "C" int dmxq_abi_version(void) { return 4; }  // 4: + dmxq_float_qdq_multi, dmxq_fixed_float_qdq_multi (round 5, additive); 3: + dmxq_weight_hypernet_multi, dmxq_unary_cast_table, dmxq_lut16_apply
#endif  // part 1b
