// csrc/blockfmt.hip — composite block formats as single kernels:
//   SBFP  numerical/format.py:453-479  per block: s = max|x| / (2^(p-1)-1); fixed(x/s) * |float(s)|
//   MXFP  numerical/format.py:545-564  per block: scale = 2^floor(log2 max|x|) / 2^(2^(e-1)); float(x/scale) * scale
// The reference composes each from 6+ ATen passes per CHUNK inside a Python loop; here: one read, one write.
//   rows path       inner == 1, L % B == 0, B = 2^k covering whole 16-byte vectors: an op of the streaming skeleton
//                   (stream.hpp), one vector per lane, block max across B/EPL lanes by DPP (as bfp_rows.hpp)
//   generic kernel  any [outer, L, inner], ragged tails: one lane per block, two strided passes
#include <math.h>

#include <cmath>

#include "floatq.hpp"
#include "stream.hpp"

namespace dmxq {

// per-element pieces shared with elementwise.hip (kept local: tiny, and it avoids a header for three functions)
__device__ __forceinline__ uint32_t rne_bits(uint32_t t, int man_bits) {  // quant_cpu.cpp:211-237, nearest
  const int sh = 23 - man_bits;
  const uint32_t mask = (1u << sh) - 1u;
  return (t + (mask >> 1) + ((t >> sh) & 1u)) & ~mask;
}
__device__ __forceinline__ float float_q_nearest(float a, int man, int exp_bits, int bias, int flush) {  // quant_cpu.cpp:359-402
  const uint32_t target = f2u(a);
  const int target_exp = (int)((target & 0x7FFFFFFFu) >> 23) - 127;
  const int min_exp = -(bias - 1);
  if (target_exp < min_exp) {
    if (flush) return 0.0f;
    const float shift = u2f(((uint32_t)(127 + min_exp) << 23) | (target & 0x80000000u));
    return u2f(rne_bits(f2u(a + shift), man)) - shift;
  }
  uint32_t qb = rne_bits(target, man);
  const int max_e = (1 << (exp_bits - 1)) + 127;
  if (qb != 0u && (int)((qb & 0x7FFFFFFFu) >> 23) > max_e)
    qb = (target & 0x80000000u) | ((uint32_t)max_e << 23) | ((0x007FFFFFu >> (23 - man)) << (23 - man));
  return u2f(qb);
}
__device__ __forceinline__ float fixed_rne(float a) {  // sim_helper.cpp:14-21 with sigma = 0 (see elementwise.hip)
  const float a1 = a + 0.5f;
  const float mag = fabsf(a1);
  const bool odd = (f2u(a1) & 1u) != 0u && mag < 16777216.0f;
  return mag >= 8388608.0f ? (odd ? a1 - 1.0f : a1) : rintf(a1 - 0.5f);
}

struct SbfpFmt { static constexpr int kThreads = 128, kUnroll = 8, kPace = 0; int p, clamp; float t_min, t_max, man_scaling; int man, exp_bits, bias, flush; };
struct MxfpFmt { static constexpr int kThreads = 256, kUnroll = 8, kPace = 6; int man, exp_bits, bias; float big; FloatFast fast; int big_log2, exact_exponent, xdomain; };

struct SbfpBlock {
  static constexpr bool kHasXDomain = false;
  float s, sc, rs;
  bool fast;  // clamped codes and a block scale whose reciprocal carries the exact-quotient argument of common.hpp
  __device__ __forceinline__ void setup(uint32_t maxbits, const SbfpFmt& f) {
    s = u2f(maxbits) / f.man_scaling;
    sc = fabsf(float_q_nearest(s, f.man, f.exp_bits, f.bias, f.flush));
    rs = 1.0f / s;
    fast = f.clamp != 0 && recip_ok(s);
  }
  __device__ __forceinline__ float apply(float x, const SbfpFmt& f) const {
    if (!(s > 0.0f)) return x;  // zero (or NaN) block: passed through (format.py:467-474 torch.where)
    float q = fixed_rne(x / s);
    if (f.clamp) q = q > f.t_max ? f.t_max : (q < f.t_min ? f.t_min : q);
    return q * sc;
  }
  // The codes are CLAMPED integers (|code| <= 2^(p-1) - 1), so the quotient x / s may come from div_for_clamped_int (3 FMA-class
  // operations instead of the ~14 of an IEEE division; the clamped code is the same for every x) and fixed_rne's |a| >= 2^23 branch
  // cannot matter (such values clamp whatever they round to): straight-line code for every lane, and the lanes whose block has
  // no such reciprocal (zero / NaN / denormal-range block maximum, unclamped format) redo theirs behind one cold branch.
  template <int N>
  __device__ __forceinline__ void apply_vec(const float (&x)[N], float (&y)[N], const SbfpFmt& f) const {
    // pairs through the packed fp32 pipe (the arithmetic of common.hpp affine_int_pairs without a zero point, the dequantisation by the
    // QUANTISED scale): bit-identical to the per-element form for every finite quotient; a lane holding an Inf / NaN quotient (where the
    // correction step and v_med3 are wrong: NaN must stay NaN) joins the cold redo
    bool redo = !fast;
    if constexpr (N % 2 == 0) {
#pragma unroll
      for (int k = 0; k < N; k += 2) {
        const f32x2 n2 = {x[k], x[k + 1]};
        const f32x2 q0 = n2 * rs;
        const f32x2 t = __builtin_elementwise_fma((f32x2){s, s}, q0, -n2);  // -(r): r = n - s q0, exact
        const f32x2 q = __builtin_elementwise_fma(-t, (f32x2){rs, rs}, q0);
        redo = redo || __builtin_amdgcn_classf(q0.x, 0x001 | 0x002 | 0x004 | 0x200) || __builtin_amdgcn_classf(q0.y, 0x001 | 0x002 | 0x004 | 0x200);
        const f32x2 u = (q + 0.5f) - 0.5f;
        f32x2 v;
        v.x = __builtin_amdgcn_fmed3f(__builtin_rintf(u.x), f.t_min, f.t_max);
        v.y = __builtin_amdgcn_fmed3f(__builtin_rintf(u.y), f.t_min, f.t_max);
        const f32x2 o = v * sc;
        y[k] = o.x;
        y[k + 1] = o.y;
      }
    } else {
      const Recip rc{s, rs};
#pragma unroll
      for (int k = 0; k < N; k++) {
        float q = rintf((div_for_clamped_int(x[k], rc) + 0.5f) - 0.5f);
        q = q > f.t_max ? f.t_max : (q < f.t_min ? f.t_min : q);
        y[k] = q * sc;
      }
    }
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(redo) != 0ull, 0)) {
      if (redo) {
#pragma unroll
        for (int k = 0; k < N; k++) y[k] = apply(x[k], f);
      }
    }
  }
};
struct MxfpBlock {
  static constexpr bool kHasXDomain = true;
  float scale, inv;
  bool zero, pow2;  // pow2: the scale is a normal power of two whose reciprocal is representable -> x / scale == x * inv exactly
  // The x-domain form (round 3).  When the block scale is a normal power of two 2^(se - 127), dividing by it, casting to the element
  // format and multiplying back commute with fp32 rounding (no intermediate leaves the normal range -- conditions below), so the element
  // cast of floatq.hpp float_q1_fast can run on x ITSELF with its three constants moved by the scale's exponent:
  //     threshold exponent   tbits = (127 + min_exp + se - 127) << 23        (below it the quantum stays 2^(min_exp - man) * scale)
  //     magic constant       M     = 2^(max(exp x, threshold) + 23 - man) * 1.5, signed
  //     saturation           maxv  = max_val * scale
  // 12 VALU per element instead of ~20 (no x * inv, no per-element range check -- every element is <= the block maximum --, no zero
  // select, no * scale), and the block set-up shrinks to ~10 operations.  Blocks it does not cover (zero / Inf / NaN maxima, scales below
  // 2^(bias - 127) or maxima from 2^105 up, float32 maxima within 88 ulps below a power of two -- the log2 rounding rule of setup()) make
  // the WAVE take the general path: one wave-uniform branch per vector.
  uint32_t tbits, k1;
  float maxv;
  __device__ __forceinline__ bool try_fast(uint32_t maxbits, const MxfpFmt& f) {
    const int eb = (int)(maxbits >> 23), se = eb - f.big_log2;
    const bool near_pow2 = !f.exact_exponent && (maxbits & 0x007FFFFFu) > 0x007FFFA7u;
    const bool ok = f.xdomain && !near_pow2 && se >= f.bias && eb <= 231;
    if (__builtin_amdgcn_ballot_w64(!ok) != 0ull) return false;
    tbits = (uint32_t)(se - (f.bias - 1)) << 23;  // 127 + min_exp + (se - 127), min_exp = -(bias - 1)
    k1 = f.fast.k1;
    maxv = u2f((uint32_t)((int)f2u(f.fast.max_val) + ((se - 127) << 23)));
    return true;
  }
  template <int N>
  __device__ __forceinline__ void apply_vec_fast(const float (&x)[N], float (&y)[N]) const {
#pragma unroll
    for (int k = 0; k < N; k++) {
      const uint32_t t = f2u(x[k]), sign = t & 0x80000000u, eb = t & 0x7F800000u;
      const uint32_t e2 = max(eb, tbits);
      const float M = u2f(e2 + k1 + sign);
      const float S = u2f(eb < tbits ? (tbits | sign) : sign);  // +-(threshold), or +-0
      const float q = ((x[k] + S) + M) - (M + S);
      y[k] = __builtin_amdgcn_fmed3f(q, -maxv, maxv);
    }
  }
  __device__ __forceinline__ void setup(uint32_t maxbits, const MxfpFmt& f) {
    const float m = u2f(maxbits);
    zero = m == 0.0f;
    // the reference evaluates 2^floor(log2 m) / 2^(2^(e-1)) in fp32 (format.py:551-555).  No libm: a float32 log2 within an ulp
    // of the truth crosses an integer only for m = 2^v (1 - j 2^-24) with j <= jmax(v) (the rule and its proof sketch are in
    // oracle/oracle.c oracle_floor_log2f; checked against torch.log2 for every exponent, fixtures tests/golden/boundaries.npz);
    // a maximum with at most 11 significant bits (bf16 / fp16 inputs) has j >= 2^13 and never does.
    int eb = (int)(maxbits >> 23);
    if (eb >= 1 && eb <= 254) {
      const uint32_t man = maxbits & 0x007FFFFFu;
      const int v = eb - 126;  // floor(log2 m) + 1
      if (!f.exact_exponent && man != 0u && v != 0) {
        const uint32_t a = (uint32_t)(v < 0 ? -v : v), j = 0x00800000u - man;
        const int c = 31 - __builtin_clz(a);
        const int g = (v > 0 && (a & (a - 1u)) == 0u) ? 25 - c : 24 - c;  // 17 .. 25
        // jmax = floor(2^24 (1 - 2^(-2^-g))): 88 44 22 11 | 5 2 1 | 0 0, as bytes of two constants
        const uint32_t jmax = g <= 20 ? ((0x0B162C58u >> (8 * (g - 17))) & 0xFFu) : (g <= 23 ? ((0x00010205u >> (8 * (g - 21))) & 0xFFu) : 0u);
        if (j <= jmax) eb += 1;
      }
      const int se = eb - f.big_log2;
      if (eb == 255) scale = INFINITY;                      // 2^128: the reference's fp32 power overflows too
      else if (se >= 1) scale = u2f((uint32_t)se << 23);
      else scale = ldexpf(1.0f, se - 127);                  // a denormal (or zero) scale, exact
    } else {
      scale = exp2f(floorf(log2f(m))) / f.big;              // zero (see `zero`), denormal, Inf, NaN maxima
    }
    const uint32_t sb = f2u(scale);
    pow2 = (sb & 0x007FFFFFu) == 0u && (sb >> 23) >= 1u && (sb >> 23) <= 253u;
    inv = u2f((254u - (sb >> 23)) << 23);
  }
  __device__ __forceinline__ float apply(float x, const MxfpFmt& f) const {
    if (zero) return x * 0.0f;
    return float_q_nearest(x / scale, f.man, f.exp_bits, f.bias, 0) * scale;
  }
  // a whole lane-vector; the branch-free element form when the wave's blocks and values allow it (floatq.hpp)
  template <int N>
  __device__ __forceinline__ void apply_vec(const float (&x)[N], float (&y)[N], const MxfpFmt& f) const {
    if (f.fast.usable) {
      bool ok = pow2 || zero;
      float v[N];
#pragma unroll
      for (int k = 0; k < N; k++) { v[k] = x[k] * inv; ok = ok && float_fast_ok(v[k], f.fast); }
      if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) {
#pragma unroll
        for (int k = 0; k < N; k++) y[k] = zero ? x[k] * 0.0f : float_q1_fast(v[k], f.fast, false, false) * scale;
        return;
      }
    }
#pragma unroll
    for (int k = 0; k < N; k++) y[k] = apply(x[k], f);
  }
};

// rows path as an op of the streaming skeleton (stream.hpp: contiguous tiles, all loads of a tile in flight before the
// arithmetic, store burst): a block is lpb adjacent lanes of one wave, its max an integer max over |x| bit patterns
// reduced with DPP moves.  A block never straddles the skeleton's partial-tile predicate (n_vec % lpb == 0, lpb | 256).
template <class FMT, class BLK>
struct BlockOp {
  static constexpr bool kHeavy = true;
  static constexpr bool kFixedVector = true;  // a block is lpb adjacent lanes of 16-byte vectors: no 8-byte-vector form for widening outputs
  static constexpr int kLoadPace = FMT::kPace;  // common.hpp OpLoadPace: MXFP 256 x 8 tiles 11.75 -> 11.5 us with 4-6 (2: 12.2); SBFP: nothing (11.6 at 0 / 2 / 4 / 6)
  static constexpr int kTileUnroll = FMT::kUnroll, kTileThreads = FMT::kThreads;  // stream.hpp, 20-32 MiB tensors: MXFP 256 x 8 (12.3 vs 13.5 us for 256 x 2, with the x-domain element cast); SBFP 128 x 8 since round 5 (re-measured with the round-3 arithmetic in place: 12.3 us against 13.9 for the 128 x 4 chosen before it, profiles/r05_tune_stream_waitall.txt "block")
  FMT f;
  int lpb;
  __device__ __forceinline__ void apply_one(float x, float& y, int64_t) const { y = x; }  // (no scalar tail: n % B == 0)
  template <int N>
  __device__ __forceinline__ void apply_vec(const float (&x)[N], float (&y)[N], int64_t) const {
    uint32_t mb = 0u;
#pragma unroll
    for (int k = 0; k < N; k++) mb = max(mb, f2u(x[k]) & 0x7FFFFFFFu);
    BLK b;
    const uint32_t gm = group_max_u32(mb, lpb);
    if constexpr (BLK::kHasXDomain) {
      if (b.try_fast(gm, f)) {  // wave-uniform
        b.apply_vec_fast(x, y);
        return;
      }
    }
    b.setup(gm, f);
    b.apply_vec(x, y, f);
  }
};

template <int DTI, int DTO, class FMT, class BLK>
__global__ __launch_bounds__(kThreads) void blockfmt_generic_kernel(const void* __restrict__ in, void* __restrict__ out,
                                                                   int64_t outer, int64_t L, int64_t inner, int64_t B,
                                                                   FMT f) {
  const int64_t nblk = (L + B - 1) / B;
  const int64_t total = outer * nblk * inner;
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x; t < total; t += stride) {
    const int64_t j = t % inner, k = (t / inner) % nblk, o = t / (inner * nblk);
    const int64_t l0 = k * B, len = (L - l0 < B) ? (L - l0) : B;
    const int64_t e0 = (o * L + l0) * inner + j;
    uint32_t mb = 0u;
    for (int64_t i = 0; i < len; i++) mb = max(mb, f2u(load1<DTI>(in, e0 + i * inner)) & 0x7FFFFFFFu);
    BLK b;
    b.setup(mb, f);
    for (int64_t i = 0; i < len; i++) store1<DTO>(out, e0 + i * inner, b.apply(load1<DTI>(in, e0 + i * inner), f));
  }
}

template <int DTI, int DTO, class FMT, class BLK>
static int launch_blockfmt(const void* in, void* out, int64_t outer, int64_t L, int64_t inner, int64_t B, const FMT& f,
                           hipStream_t s) {
  constexpr int EPL = 16 / Elem<DTI>::bytes;
  const int64_t n = outer * L * inner;
  const bool pow2 = (B & (B - 1)) == 0;
  if (inner == 1 && L % B == 0 && pow2 && B >= EPL && B <= 64 * EPL) {  // (any base alignment: stream.hpp)
    return launch_stream<DTI, DTO>(in, out, n, BlockOp<FMT, BLK>{f, (int)(B / EPL)}, s);
  } else {
    const int64_t nblk = (L + B - 1) / B;
    DMXQ_LAUNCH((blockfmt_generic_kernel<DTI, DTO, FMT, BLK>), dim3(grid_for(outer * nblk * inner)), dim3(kThreads), 0,
                       s, in, out, outer, L, inner, B, f);
  }
  return launch_status();
}

template <class FMT, class BLK>
static int dispatch_blockfmt(const void* in, void* out, int dti, int dto, int64_t outer, int64_t L, int64_t inner,
                             int64_t B, const FMT& f, hipStream_t s) {
#define DMXQ_DT(I_, O_) \
  if (dti == I_ && dto == O_) return launch_blockfmt<I_, O_, FMT, BLK>(in, out, outer, L, inner, B, f, s);
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

extern "C" int dmxq_sbfp_qdq(const void* in, void* out, int dtype_in, int dtype_out, int64_t outer, int64_t L,
                             int64_t inner, int64_t block_size, int precision, int clamp, int symmetric,
                             int scaler_man_bits, int scaler_exp_bits, int scaler_exp_bias, int scaler_flush_subnormal,
                             void* stream) {
  if (!valid_dtype(dtype_in) || !valid_dtype(dtype_out) || outer < 0 || L < 0 || inner < 0 || block_size < 1) return DMXQ_ERR_BAD_ARG;
  if (precision < 2 || precision > 24 || scaler_exp_bits < 1 || scaler_exp_bits > 8 || scaler_man_bits < 0) return DMXQ_ERR_BAD_ARG;
  if (scaler_man_bits > 22) return DMXQ_ERR_UNSUPPORTED;
  if (outer * L * inner == 0) return DMXQ_OK;
  if (!in || !out) return DMXQ_ERR_BAD_ARG;
  float t_min = (float)(-ldexp(1.0, precision - 1));            // sim_helper.cpp:5-12 with fl = 0
  const float t_max = (float)(-(double)t_min - 1.0);
  if (symmetric) t_min = (float)((double)t_min + 1.0);
  const SbfpFmt f{precision, clamp ? 1 : 0, t_min, t_max, (float)((1 << (precision - 1)) - 1), scaler_man_bits,
                  scaler_exp_bits, scaler_exp_bias, scaler_flush_subnormal ? 1 : 0};
  return dispatch_blockfmt<SbfpFmt, SbfpBlock>(in, out, dtype_in, dtype_out, outer, L, inner, block_size, f, (hipStream_t)stream);
}

extern "C" int dmxq_mxfp_qdq(const void* in, void* out, int dtype_in, int dtype_out, int64_t outer, int64_t L,
                             int64_t inner, int64_t block_size, int man_bits, int exp_bits, void* stream) {
  if (!valid_dtype(dtype_in) || !valid_dtype(dtype_out) || outer < 0 || L < 0 || inner < 0 || block_size < 1) return DMXQ_ERR_BAD_ARG;
  if (exp_bits < 1 || exp_bits > 8 || man_bits < 0) return DMXQ_ERR_BAD_ARG;
  if (man_bits > 22) return DMXQ_ERR_UNSUPPORTED;
  if (outer * L * inner == 0) return DMXQ_OK;
  if (!in || !out) return DMXQ_ERR_BAD_ARG;
  MxfpFmt f{man_bits, exp_bits, (1 << (exp_bits - 1)) - 1, (float)ldexp(1.0, 1 << (exp_bits - 1)),
            make_float_fast(man_bits, exp_bits, (1 << (exp_bits - 1)) - 1), 1 << (exp_bits - 1),
            dtype_in != DMXQ_F32 ? 1 : 0, 0};
  f.xdomain = (f.fast.usable && std::isfinite(f.fast.max_val) && f.bias >= 1) ? 1 : 0;  // MxfpBlock::try_fast
  return dispatch_blockfmt<MxfpFmt, MxfpBlock>(in, out, dtype_in, dtype_out, outer, L, inner, block_size, f, (hipStream_t)stream);
}
