// csrc/bfp.hip — block-floating-point fused quantize->dequantize for gfx950.
//
// Replaces the reference's BlockFloatingPoint.cast (numerical/format.py:304-343: .float() -> transpose ->
// split into ceil(L/B) chunks -> per chunk block_quantize -> cat -> transpose back) and its native leaf
// (quant_cpu.cpp:239-311; CUDA twin quant_cuda/quant.cu:14-112 + block_kernel.cu) with ONE launch per tensor
// that reads every element once and writes it once.
//
// Per block (exactly the reference arithmetic, see oracle/oracle.c bfp_q1):
//   m = max|x| ; E = bits(m) & 0x7F800000 ; base = 6 * float(E)
//   t = x + base                     (fp32 RNE add — the reference's deliberate first rounding)
//   t' = round t's mantissa to `wl` bits (mode: nearest-even / down / up / stochastic) on the bit pattern
//   q = t' - base ; clip: exponent(q) > E  ->  sign | E | top (wl-2) mantissa bits
// The clip is written as med3(q, -maxv, +maxv): |q| <= 2^(e+1) always holds (t' stays inside [4,8]*2^e), so
// "exponent field above E" <=> |q| == 2^(e+1) <=> |q| > maxv, and the clamp returns the same bits.
// Asymmetric formats ("(_N)", format.py:349-372) relax only the negative clip by one code; in closed form:
//   x <= -(2^(e+1) - quantum/2)  ->  y = -2^(e+1)     (tie goes to the even code -2^(wl-1)).
//
// Kernels:
//   bfp_rows_kernel   inner == 1, L % B == 0, B a power of two that one lane group covers: the tensor is a
//                     flat stream of blocks; each lane owns 16 B of input, a block spans B/EPL adjacent
//                     lanes and the block max is reduced with DPP moves (no LDS, no second read).
//   bfp_cols_kernel   inner > 1 (block_dim = -2 / conv dim 1), inner % VEC == 0: lanes run along the
//                     contiguous inner dim, each lane keeps its B x VEC column tile in registers.
//   bfp_lds_rows_kernel inner == 1, any L / B (ragged tails, odd row pitch): a workgroup stages a span of whole
//                     blocks in LDS with coalesced loads, lanes then own blocks inside LDS.
//   bfp_generic_kernel  everything else: one lane per block, strided two-pass (correct for any layout).
#include "common.hpp"

namespace dmxq {

struct BfpBlockParams {
  float base;    // 6 * 2^e
  float maxv;    // largest representable magnitude, 2^(e+1) - quantum
  float thr;     // asymmetric threshold  -(2^(e+1) - quantum/2)
  float neg_lim; // -2^(e+1)
};

template <bool ASYM>
__device__ __forceinline__ BfpBlockParams bfp_block_params(float maxabs, int wl) {
  BfpBlockParams p;
  const uint32_t E = f2u(maxabs) & 0x7F800000u;
  p.base = u2f(E) * 6.0f;
  const uint32_t max_man = (0x007FFFFFu >> (25 - wl)) << (25 - wl);
  p.maxv = u2f(E | max_man);
  if (ASYM) {
    const uint32_t thr_man = (0x007FFFFFu >> (24 - wl)) << (24 - wl);
    p.thr = u2f(0x80000000u | E | thr_man);
    p.neg_lim = u2f(0x80000000u | (E + 0x00800000u));
  }
  return p;
}

template <int RND, bool ASYM>
__device__ __forceinline__ float bfp_q1(float x, const BfpBlockParams& p, int wl, int rounding, uint32_t rnd) {
  const float t = x + p.base;
  const uint32_t tb = round_bitwise<RND>(f2u(t), wl, rounding, rnd);
  float q = u2f(tb) - p.base;
  q = __builtin_amdgcn_fmed3f(q, -p.maxv, p.maxv);
  if (ASYM) q = (x <= p.thr) ? p.neg_lim : q;
  return q;
}

// ---------------------------------------------------------------------------------------------------------
// Row blocks, flat stream.  n_vec = number of 16-byte input vectors (= numel / EPL).
template <int DTI, int DTO, int RND, bool ASYM, int UNROLL, bool NT>
__global__ __launch_bounds__(kThreads) void bfp_rows_kernel(const void* __restrict__ in, void* __restrict__ out,
                                                           int64_t n_vec, int lpb /*lanes per block*/, int wl,
                                                           int rounding, uint64_t seed) {
  const bool stoch = (RND == kRuntimeRounding) && rounding == DMXQ_ROUND_STOCHASTIC;
  constexpr int EPL = 16 / Elem<DTI>::bytes;
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  int64_t v = (int64_t)blockIdx.x * kThreads + threadIdx.x;

  // main body: UNROLL independent 16-byte loads in flight per lane before any arithmetic
  for (; v + (UNROLL - 1) * stride < n_vec; v += UNROLL * stride) {
    float x[UNROLL][EPL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) load_vec<DTI, EPL>(in, (v + u * stride) * EPL, x[u]);
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      float m = 0.0f;
#pragma unroll
      for (int k = 0; k < EPL; k++) m = fmaxf(m, fabsf(x[u][k]));
      m = group_max(m, lpb);
      const BfpBlockParams p = bfp_block_params<ASYM>(m, wl);
      float y[EPL];
      const int64_t e0 = (v + u * stride) * EPL;
#pragma unroll
      for (int k = 0; k < EPL; k++)
        y[k] = bfp_q1<RND, ASYM>(x[u][k], p, wl, rounding, stoch ? rnd_bits(seed, (uint64_t)(e0 + k)) : 0u);
      store_vec<DTO, EPL, NT>(out, e0, y);
    }
  }
  for (; v < n_vec; v += stride) {
    float x[EPL];
    load_vec<DTI, EPL>(in, v * EPL, x);
    float m = 0.0f;
#pragma unroll
    for (int k = 0; k < EPL; k++) m = fmaxf(m, fabsf(x[k]));
    m = group_max(m, lpb);
    const BfpBlockParams p = bfp_block_params<ASYM>(m, wl);
    float y[EPL];
#pragma unroll
    for (int k = 0; k < EPL; k++)
      y[k] = bfp_q1<RND, ASYM>(x[k], p, wl, rounding, stoch ? rnd_bits(seed, (uint64_t)(v * EPL + k)) : 0u);
    store_vec<DTO, EPL, NT>(out, v * EPL, y);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Generic fallback: one lane per block; two strided passes.  Correct for every (outer, L, inner, B) incl.
// ragged tails; coalesced across lanes when inner > 1.
template <int DTI, int DTO, int RND, bool ASYM>
__global__ __launch_bounds__(kThreads) void bfp_generic_kernel(const void* __restrict__ in, void* __restrict__ out,
                                                              int64_t outer, int64_t L, int64_t inner, int64_t B,
                                                              int wl, int rounding, uint64_t seed) {
  const bool stoch = (RND == kRuntimeRounding) && rounding == DMXQ_ROUND_STOCHASTIC;
  const int64_t nblk = (L + B - 1) / B;
  const int64_t total = outer * nblk * inner;
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x; t < total; t += stride) {
    const int64_t j = t % inner;
    const int64_t k = (t / inner) % nblk;
    const int64_t o = t / (inner * nblk);
    const int64_t l0 = k * B;
    const int64_t len = (L - l0 < B) ? (L - l0) : B;
    const int64_t e0 = (o * L + l0) * inner + j;
    float m = 0.0f;
    for (int64_t i = 0; i < len; i++) m = fmaxf(m, fabsf(load1<DTI>(in, e0 + i * inner)));
    const BfpBlockParams p = bfp_block_params<ASYM>(m, wl);
    for (int64_t i = 0; i < len; i++) {
      const int64_t e = e0 + i * inner;
      // the oracle numbers random draws by the element's position in the transposed [rows, L] matrix
      const uint64_t ridx = (uint64_t)(((o * inner + j) * L) + l0 + i);
      store1<DTO>(out, e, bfp_q1<RND, ASYM>(load1<DTI>(in, e), p, wl, rounding, stoch ? rnd_bits(seed, ridx) : 0u));
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// host-side dispatch
template <int DTI, int DTO, int RND, bool ASYM>
static int launch_bfp(const void* in, void* out, int64_t outer, int64_t L, int64_t inner, int64_t B, int wl,
                      int rounding, uint64_t seed, hipStream_t s) {
  constexpr int EPL = 16 / Elem<DTI>::bytes;
  const int64_t n = outer * L * inner;
  const bool pow2 = (B & (B - 1)) == 0;
  // (stochastic draws are numbered by flat element index in the rows kernel, which equals the oracle's
  //  numbering because inner == 1 on this path)
  if (inner == 1 && L % B == 0 && pow2 && B >= EPL && B <= 64 * EPL && aligned16(in) && aligned16(out)) {
    constexpr int UNROLL = 4;
    const int64_t n_vec = n / EPL;
    const int grid = grid_for((n_vec + UNROLL - 1) / UNROLL);
    hipLaunchKernelGGL((bfp_rows_kernel<DTI, DTO, RND, ASYM, UNROLL, false>), dim3(grid), dim3(kThreads), 0, s, in,
                       out, n_vec, (int)(B / EPL), wl, rounding, seed);
    return launch_status();
  }
  const int64_t nblk = (L + B - 1) / B;
  const int grid = grid_for(outer * nblk * inner);
  hipLaunchKernelGGL((bfp_generic_kernel<DTI, DTO, RND, ASYM>), dim3(grid), dim3(kThreads), 0, s, in, out, outer, L,
                     inner, B, wl, rounding, seed);
  return launch_status();
}

template <int DTI, int DTO>
static int dispatch_mode(const void* in, void* out, int64_t outer, int64_t L, int64_t inner, int64_t B, int wl,
                         int rounding, bool asym, uint64_t seed, hipStream_t s) {
  if (rounding == DMXQ_ROUND_NEAREST)
    return asym ? launch_bfp<DTI, DTO, DMXQ_ROUND_NEAREST, true>(in, out, outer, L, inner, B, wl, rounding, seed, s)
                : launch_bfp<DTI, DTO, DMXQ_ROUND_NEAREST, false>(in, out, outer, L, inner, B, wl, rounding, seed, s);
  return asym ? launch_bfp<DTI, DTO, kRuntimeRounding, true>(in, out, outer, L, inner, B, wl, rounding, seed, s)
              : launch_bfp<DTI, DTO, kRuntimeRounding, false>(in, out, outer, L, inner, B, wl, rounding, seed, s);
}

}  // namespace dmxq

extern "C" int dmxq_float_qdq(const void* in, void* out, int dtype_in, int dtype_out, int64_t n, int man_bits,
                              int exp_bits, int exp_bias, int flush_subnormal, int unsigned_abs, int rounding,
                              uint64_t seed, void* stream);

extern "C" int dmxq_bfp_qdq(const void* in, void* out, int dtype_in, int dtype_out, int64_t outer, int64_t L,
                            int64_t inner, int64_t block_size, int precision, int rounding, int symmetric,
                            uint64_t seed, void* stream) {
  using namespace dmxq;
  if (!valid_dtype(dtype_in) || !valid_dtype(dtype_out) || !valid_rounding(rounding)) return DMXQ_ERR_BAD_ARG;
  if (outer < 0 || L < 0 || inner < 0 || block_size < 1 || precision < 2) return DMXQ_ERR_BAD_ARG;
  const int64_t n = outer * L * inner;
  if (n == 0) return DMXQ_OK;
  if (!in || !out) return DMXQ_ERR_BAD_ARG;
  if (block_size == 1)  // numerical/format.py:312-320: BFP with block size 1 borrows float_quantize
    return dmxq_float_qdq(in, out, dtype_in, dtype_out, n, precision - 2, 8, 127, 0, 0, rounding, seed, stream);
  if (precision > 22) return DMXQ_ERR_UNSUPPORTED;  // reference shifts by a negative count (UB) beyond this
  hipStream_t s = (hipStream_t)stream;
  const bool asym = !symmetric;
#define DMXQ_DT(I_, O_)                                                                                        \
  if (dtype_in == I_ && dtype_out == O_)                                                                       \
    return dispatch_mode<I_, O_>(in, out, outer, L, inner, block_size, precision, rounding, asym, seed, s);
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
