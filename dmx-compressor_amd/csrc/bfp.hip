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
//   bfp_urows_kernel  inner == 1, ragged rows (L % B != 0) and / or rows and bases at any element alignment: rows read
//                     directly with unaligned 16-byte accesses, vectors numbered in a virtual space padded to whole
//                     blocks (bfp_urows.hip).
//   bfp_generic_kernel  everything else: one lane per block, strided two-pass (correct for any layout).
#include <stdio.h>

#include "bfp_rows.hpp"

// This file is compiled THREE times (build.py: -DDMXQ_EW_PART=1 / 2 / 3): the flat-stream kernel has ~9 tile shapes x 5 block-size cases per
// (dtype pair, symmetry, rounding path) -- one translation unit took 5-7 minutes, the critical path of the whole build.
//   part 1: the C entry points, 16-bit -> same 16-bit, the multi-tensor kernels;  part 2: float32 inputs;  part 3: 16-bit -> float32.
#ifndef DMXQ_EW_PART
#define DMXQ_EW_PART 0
#endif
#define DMXQ_BP(P_) (DMXQ_EW_PART == 0 || DMXQ_EW_PART == (P_))

namespace dmxq {

constexpr int kRowsMaxGrid = 1 << 20;

// (tile geometry: rows_plan, common.hpp)

// ---------------------------------------------------------------------------------------------------------
// Generic fallback: one lane per block; two strided passes.  Correct for every (outer, L, inner, B) incl.
// ragged tails; coalesced across lanes when inner > 1.
// NATIVE: the reference's native symmetric == false branch (quant_cpu.cpp:247-253; the Python layer never requests it,
// format.py:332): an element equal to -max, when the top 7 mantissa bits of the block maximum are all ones, is
// quantised with the next exponent -- per element, no post-pass.  Only reachable through the S1 seam (quant.py).
template <int DTI, int DTO, int RND, bool ASYM, bool NATIVE = false>
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
    uint32_t mb = 0u;
    for (int64_t i = 0; i < len; i++) mb = max(mb, f2u(load1<DTI>(in, e0 + i * inner)) & 0x7FFFFFFFu);
    const BfpBlockParams p = bfp_block_params<ASYM>(mb, wl);
    for (int64_t i = 0; i < len; i++) {
      const int64_t e = e0 + i * inner;
      // the oracle numbers random draws by the element's position in the transposed [rows, L] matrix
      const uint64_t ridx = (uint64_t)(((o * inner + j) * L) + l0 + i);
      const float x = load1<DTI>(in, e);
      if (NATIVE && x == -u2f(mb) && ((mb >> 16) & 0x7Fu) == 0x7Fu) {
        const BfpBlockParams pe = bfp_block_params<false>(((mb >> 23) + 1u) << 23, wl);
        store1<DTO>(out, e, bfp_q1<RND, false>(x, pe, wl, rounding, bfp_rnd_if(stoch, seed, ridx)));
      } else {
        store1<DTO>(out, e, bfp_q1<RND, ASYM>(x, p, wl, rounding, bfp_rnd_if(stoch, seed, ridx)));
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// host-side dispatch
template <int DTI, int DTO, int RND, bool ASYM>
static int launch_bfp(const void* in, void* out, int64_t outer, int64_t L, int64_t inner, int64_t B, int wl,
                      int rounding, uint64_t seed, hipStream_t s) {
  // lane-vector: 16 B of input, or 8 B when the output is wider than the input (bfp_rows.hpp IVB)
  constexpr int IVB = (Elem<DTO>::bytes > Elem<DTI>::bytes) ? 8 : 16;
  constexpr int EPL = IVB / Elem<DTI>::bytes;
  const int64_t n = outer * L * inner;
  const bool pow2 = (B & (B - 1)) == 0;
  // (stochastic draws are numbered by flat element index in the rows kernel, which equals the oracle's
  //  numbering because inner == 1 on this path)
  if (inner == 1 && L % B == 0 && pow2 && B >= EPL && B <= 64 * EPL && aligned16(in) && aligned16(out)) {
    // Geometry: rows_plan() above; non-temporal loads and stores, every load of a tile in flight before the arithmetic
    // starts, stores in one burst.
    constexpr int MODE = kRowsNtLoad | kRowsNtStore;
    const int64_t n_vec = n / EPL;
    const int lpb = (int)(B / EPL);
    // nearest-even: the magic-add path (bfp_math.hpp (2)); single rounding where the input dtype allows it
    const int fast = (RND == DMXQ_ROUND_NEAREST && wl <= 20) ? (bfp_single_rounding_ok<DTI>(wl) ? 2 : 1) : 0;
#define DMXQ_ROWS_GP(T_, U_, F_, G_, P_)                                                                         \
  do {                                                                                                           \
    const int64_t tiles = (n_vec + (int64_t)(T_) * (U_) - 1) / ((int64_t)(T_) * (U_));                           \
    const int grid = (int)(tiles < kRowsMaxGrid ? tiles : kRowsMaxGrid);                                         \
    DMXQ_LAUNCH((bfp_rows_kernel<DTI, DTO, RND, ASYM, U_, MODE, T_, F_, G_, IVB, 0, P_>), dim3(grid), dim3(T_), 0, s, in, \
                       out, n_vec, lpb, wl, rounding, seed);                                                     \
  } while (0)
#define DMXQ_ROWS_G(T_, U_, F_, G_) DMXQ_ROWS_GP(T_, U_, F_, G_, 0)
#define DMXQ_ROWS(T_, U_, F_) DMXQ_ROWS_G(T_, U_, F_, U_)
#define DMXQ_ROWS_GEOM(F_)                                                                         \
  do {                                                                                             \
    /* the exact-depth one-round plans (rows_plan): symmetric same-dtype builds; 17 / 18 vectors only where they fit 256 VGPRs */ \
    constexpr int kDepth = ((F_) != 4 && DTO == DTI && !ASYM) ? ((F_) == 2 ? 20 : 16) : 0;         \
    /* the one-round plans up to 512 x 16: not the any-rounding and stochastic builds (1 KiB of scratch per lane at 16 vectors) */ \
    constexpr bool kBig = (F_) != 4 || RND == DMXQ_ROUND_DOWN || RND == DMXQ_ROUND_UP;             \
    const RowsPlan pl = rows_plan(n_vec, kBig, kDepth);                                            \
    if constexpr (kDepth >= 16) {                                                                  \
      if (pl.id == 111) { DMXQ_ROWS(512, 11, F_); break; }                                         \
      if (pl.id == 112) { DMXQ_ROWS(512, 12, F_); break; }                                         \
      if (pl.id == 113) { DMXQ_ROWS(512, 13, F_); break; }                                         \
      if (pl.id == 114) { DMXQ_ROWS(512, 14, F_); break; }                                         \
      if (pl.id == 115) { DMXQ_ROWS(512, 15, F_); break; }                                         \
      if (pl.id == 116) { DMXQ_ROWS(512, 16, F_); break; }                                         \
    }                                                                                              \
    if constexpr (kDepth >= 18) {                                                                  \
      if (pl.id == 117) { DMXQ_ROWS(512, 17, F_); break; }                                         \
      if (pl.id == 118) { DMXQ_ROWS(512, 18, F_); break; }                                         \
      /* 19 / 20 vectors: the compact kernel (results in place of the raw vectors), bfp_rows.hpp */ \
      if (pl.id == 119) { DMXQ_LAUNCH((bfp_rows_compact_kernel<DTI, 19, 512, 19>), dim3((unsigned)pl.tiles), dim3(512), 0, s, in, out, n_vec, lpb, wl); break; } \
      if (pl.id == 120) { DMXQ_LAUNCH((bfp_rows_compact_kernel<DTI, 20, 512, 10>), dim3((unsigned)pl.tiles), dim3(512), 0, s, in, out, n_vec, lpb, wl); break; } \
    }                                                                                              \
    if constexpr (kBig) {                                                                          \
      /* paced loads (common.hpp pace_issue, 16 idle issue cycles between a wave's loads): 2560 x 4096 bf16 8.15 -> 7.04 us (64 -> 74.5 %), */ \
      /* 2048 rows 6.31 -> 6.11; the one-round depths 11 .. 20 and 512 x 2 gain nothing (profiles/r05_tune_pace.txt section 10) */ \
      if (pl.id == 2) { DMXQ_ROWS_GP(512, 4, F_, 4, 2); break; }                                   \
      if (pl.id == 3) { DMXQ_ROWS_GP(128, 8, F_, 8, 2); break; }                                   \
      if (pl.id == 4) { DMXQ_ROWS(512, 16, F_); break; }                                           \
    }                                                                                              \
    if (pl.id == 0) DMXQ_ROWS(512, 1, F_);                                                         \
    else if (pl.id == 1) DMXQ_ROWS(128, 2, F_);                                                    \
    else DMXQ_ROWS(512, 2, F_);                                                                    \
  } while (0)
    // instantiate only what can run (see bfp_cols.hip): literal path for the runtime-rounding build, magic-add for
    // nearest-even; nearest with wl > 20 is routed to the runtime-rounding build by dispatch_mode
    constexpr bool in16 = Elem<DTI>::bytes == 2;
    if constexpr (RND != DMXQ_ROUND_NEAREST) {
      // literal rounding on the bit pattern + clamp (bfp_math.hpp (5)).  RND = down / up / stochastic as a COMPILE-TIME mode (round 5;
      // symmetric formats): round_bitwise folds to 1-2 operations per element and the draw is computed without the run-time mode's
      // branch and scheduling fence -- the any-rounding build paid a chain of scalar compares and selects per element (50-57 % of
      // the roofline on 4096 x 4096 bf16) and spilled on deep tiles
      DMXQ_ROWS_GEOM(4);
    } else {
      if (in16 && fast == 2) {
        if constexpr (in16) DMXQ_ROWS_GEOM(2);
      } else {
        DMXQ_ROWS_GEOM(1);
      }
    }
#undef DMXQ_ROWS_GEOM
#undef DMXQ_ROWS
#undef DMXQ_ROWS_G
#undef DMXQ_ROWS_GP
    return launch_status();
  }
  if constexpr (RND != DMXQ_ROUND_NEAREST && RND != kRuntimeRounding) {
    // (the compile-time modes exist for the row kernel only: everything else shares the any-rounding build)
    return launch_bfp<DTI, DTO, kRuntimeRounding, ASYM>(in, out, outer, L, inner, B, wl, rounding, seed, s);
  } else {
    const int64_t nblk = (L + B - 1) / B;
    const int grid = grid_for(outer * nblk * inner);
    DMXQ_LAUNCH((bfp_generic_kernel<DTI, DTO, RND, ASYM>), dim3(grid), dim3(kThreads), 0, s, in, out, outer, L,
                       inner, B, wl, rounding, seed);
    return launch_status();
  }
}

template <int DTI, int DTO>
static int dispatch_mode(const void* in, void* out, int64_t outer, int64_t L, int64_t inner, int64_t B, int wl,
                         int rounding, bool asym, uint64_t seed, hipStream_t s) {
  if (rounding == DMXQ_ROUND_NEAREST && wl <= 20)
    return asym ? launch_bfp<DTI, DTO, DMXQ_ROUND_NEAREST, true>(in, out, outer, L, inner, B, wl, rounding, seed, s)
                : launch_bfp<DTI, DTO, DMXQ_ROUND_NEAREST, false>(in, out, outer, L, inner, B, wl, rounding, seed, s);
  if (!asym && wl <= 20) {
    if (rounding == DMXQ_ROUND_DOWN) return launch_bfp<DTI, DTO, DMXQ_ROUND_DOWN, false>(in, out, outer, L, inner, B, wl, rounding, seed, s);
    if (rounding == DMXQ_ROUND_UP) return launch_bfp<DTI, DTO, DMXQ_ROUND_UP, false>(in, out, outer, L, inner, B, wl, rounding, seed, s);
    if (rounding == DMXQ_ROUND_STOCHASTIC) return launch_bfp<DTI, DTO, DMXQ_ROUND_STOCHASTIC, false>(in, out, outer, L, inner, B, wl, rounding, seed, s);
  }
  return asym ? launch_bfp<DTI, DTO, kRuntimeRounding, true>(in, out, outer, L, inner, B, wl, rounding, seed, s)
              : launch_bfp<DTI, DTO, kRuntimeRounding, false>(in, out, outer, L, inner, B, wl, rounding, seed, s);
}

}  // namespace dmxq

// Which strided-block tensors take the LDS slab kernel (bfp_slab.hip) instead of the register-tiled column kernel (bfp_cols.hip).
// DMXQ_SLAB=0 / 1 overrides (A/B runs): never / whenever the kernel applies.
static bool slab_preferred(int dtype_in, int64_t L, int64_t inner, int64_t B) {
  static const int forced = [] { const char* e = getenv("DMXQ_SLAB"); return e ? atoi(e) : -1; }();
  if (forced >= 0) return forced != 0;
  (void)dtype_in; (void)L;
  // rows that are NOT whole 128-byte lines (14 x 14, 28 x 28 maps: the column kernel's row pieces share their first and last line with the
  // neighbouring piece, 50 % of the roofline) and slabs of at least 16 KiB (smaller tiles are all overhead: [8, 12, 1500, 64] along the
  // sequence 38 % here, 54 % there).  Whole-line rows stay with the column kernel: on 134 MB of attention operands blocked along the
  // sequence the slab kernel is ahead for rows of 256 bytes ([8, 32, 2048, 128] B = 64 / 128: 76 / 71 % against 73 / 62 %) and behind for
  // 384 / 512 bytes, but at the sizes a decoder layer has it is BEHIND ([4, 32, 512, 128]: 7.3 vs 6.3 us inside a graph) and a configured
  // Llama layer's graph went 617-621 -> 632-634 us with that routing (profiles/r06_slab_ab6*.txt, r06_slab_small_operands.txt).
  return (inner * 2) % 128 != 0 && B * inner * 2 >= 16 * 1024;
}

extern "C" int dmxq_float_qdq(const void* in, void* out, int dtype_in, int dtype_out, int64_t n, int man_bits,
                              int exp_bits, int exp_bias, int flush_subnormal, int unsigned_abs, int rounding,
                              uint64_t seed, void* stream);
// bfp_urows.hip: direct (unaligned 16-byte access) kernel for ragged / unaligned rows
extern "C" int dmxq_internal_bfp_urows(const void* in, void* out, int dtype_in, int dtype_out, int64_t rows, int64_t L,
                                       int64_t B, int wl, int rounding, int symmetric, uint64_t seed, void* stream);
// bfp_cols.hip: register-tiled kernel for blocks along a non-contiguous dimension; DMXQ_ERR_UNSUPPORTED = not applicable
extern "C" int dmxq_internal_bfp_cols(const void* in, void* out, int dtype_in, int dtype_out, int64_t outer, int64_t L,
                                      int64_t inner, int64_t B, int wl, int rounding, int symmetric, uint64_t seed,
                                      void* stream);

extern "C" int dmxq_internal_bfp_smallinner(const void* in, void* out, int dtype_in, int dtype_out, int64_t outer, int64_t L,
                                            int64_t inner, int64_t B, int wl, int rounding, int symmetric, void* stream);
// bfp_slab.hip (round 6): [B rows x inner] slabs / column tiles through the LDS for feature-map sized inner extents
extern "C" int dmxq_internal_bfp_slab(const void* in, void* out, int dtype_in, int dtype_out, int64_t outer, int64_t L, int64_t inner,
                                      int64_t B, int wl, int rounding, int symmetric, void* stream);

// parts 2 / 3: the flat-stream / generic dispatch of the other dtype pairs
extern "C" int dmxq_internal_bfp_flat_f32(const void* in, void* out, int dtype_out, int64_t outer, int64_t L, int64_t inner, int64_t block_size,
                                          int precision, int rounding, int asym, uint64_t seed, void* stream);
extern "C" int dmxq_internal_bfp_flat_widen(const void* in, void* out, int dtype_in, int64_t outer, int64_t L, int64_t inner, int64_t block_size,
                                            int precision, int rounding, int asym, uint64_t seed, void* stream);
#if DMXQ_BP(2)
extern "C" int dmxq_internal_bfp_flat_f32(const void* in, void* out, int dtype_out, int64_t outer, int64_t L, int64_t inner, int64_t block_size,
                                          int precision, int rounding, int asym, uint64_t seed, void* stream) {
  using namespace dmxq;
  hipStream_t s = (hipStream_t)stream;
  if (dtype_out == DMXQ_F32) return dispatch_mode<DMXQ_F32, DMXQ_F32>(in, out, outer, L, inner, block_size, precision, rounding, asym != 0, seed, s);
  if (dtype_out == DMXQ_BF16) return dispatch_mode<DMXQ_F32, DMXQ_BF16>(in, out, outer, L, inner, block_size, precision, rounding, asym != 0, seed, s);
  if (dtype_out == DMXQ_F16) return dispatch_mode<DMXQ_F32, DMXQ_F16>(in, out, outer, L, inner, block_size, precision, rounding, asym != 0, seed, s);
  return DMXQ_ERR_BAD_ARG;
}
#endif
#if DMXQ_BP(3)
extern "C" int dmxq_internal_bfp_flat_widen(const void* in, void* out, int dtype_in, int64_t outer, int64_t L, int64_t inner, int64_t block_size,
                                            int precision, int rounding, int asym, uint64_t seed, void* stream) {
  using namespace dmxq;
  hipStream_t s = (hipStream_t)stream;
  if (dtype_in == DMXQ_BF16) return dispatch_mode<DMXQ_BF16, DMXQ_F32>(in, out, outer, L, inner, block_size, precision, rounding, asym != 0, seed, s);
  if (dtype_in == DMXQ_F16) return dispatch_mode<DMXQ_F16, DMXQ_F32>(in, out, outer, L, inner, block_size, precision, rounding, asym != 0, seed, s);
  return DMXQ_ERR_BAD_ARG;
}
#endif

#if DMXQ_BP(1)
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
  if (symmetric == DMXQ_BFP_ASYM_NATIVE) {  // the pybind seam's symmetric = false (float32 only, like the seam)
    if (dtype_in != DMXQ_F32 || dtype_out != DMXQ_F32) return DMXQ_ERR_UNSUPPORTED;
    const int64_t nblk = (L + block_size - 1) / block_size;
    DMXQ_LAUNCH((bfp_generic_kernel<DMXQ_F32, DMXQ_F32, kRuntimeRounding, false, true>), dim3(grid_for(outer * nblk * inner)),
                dim3(kThreads), 0, (hipStream_t)stream, in, out, outer, L, inner, block_size, precision, rounding, seed);
    return launch_status();
  }
  if (inner == 1) {
    // flat-stream kernel (launch_bfp) when rows are whole blocks and 16-byte aligned; otherwise LDS re-alignment
    const bool widening = dtype_in != DMXQ_F32 && dtype_out == DMXQ_F32;  // lane-vector of 8 B, see launch_bfp
    const int epl = (dtype_in == DMXQ_F32 || widening) ? 4 : 8;
    const bool pow2 = (block_size & (block_size - 1)) == 0;
    const bool rows_ok = L % block_size == 0 && pow2 && block_size >= epl && block_size <= 64 * epl && aligned16(in) && aligned16(out);
    if (!rows_ok) {
      const int ru = dmxq_internal_bfp_urows(in, out, dtype_in, dtype_out, outer, L, block_size, precision, rounding,
                                             symmetric, seed, stream);
      if (ru != DMXQ_ERR_UNSUPPORTED) return ru;
    }
  }
  if (inner > 1 && inner < 64) {  // a few elements between the members of a block: sub-slabs through the LDS (bfp_smallinner.hip)
    const int rc = dmxq_internal_bfp_smallinner(in, out, dtype_in, dtype_out, outer, L, inner, block_size, precision, rounding, symmetric,
                                                stream);
    if (rc != DMXQ_ERR_UNSUPPORTED) return rc;
  }
  if (inner >= 64 && slab_preferred(dtype_in, L, inner, block_size)) {
    const int rc = dmxq_internal_bfp_slab(in, out, dtype_in, dtype_out, outer, L, inner, block_size, precision, rounding, symmetric, stream);
    if (rc != DMXQ_ERR_UNSUPPORTED) return rc;
  }
  if (inner > 1) {
    const int rc = dmxq_internal_bfp_cols(in, out, dtype_in, dtype_out, outer, L, inner, block_size, precision, rounding,
                                          symmetric, seed, stream);
    if (rc != DMXQ_ERR_UNSUPPORTED) return rc;
  }
  hipStream_t s = (hipStream_t)stream;
  const bool asym = !symmetric;
#define DMXQ_DT(I_, O_)                                                                                        \
  if (dtype_in == I_ && dtype_out == O_)                                                                       \
    return dispatch_mode<I_, O_>(in, out, outer, L, inner, block_size, precision, rounding, asym, seed, s);
  DMXQ_DT(DMXQ_BF16, DMXQ_BF16)
  DMXQ_DT(DMXQ_F16, DMXQ_F16)
#undef DMXQ_DT
  if (dtype_in == DMXQ_F32) return dmxq_internal_bfp_flat_f32(in, out, dtype_out, outer, L, inner, block_size, precision, rounding, asym ? 1 : 0, seed, stream);
  if (dtype_out == DMXQ_F32) return dmxq_internal_bfp_flat_widen(in, out, dtype_in, outer, L, inner, block_size, precision, rounding, asym ? 1 : 0, seed, stream);
  return DMXQ_ERR_BAD_ARG;
}

// ---------------------------------------------------------------------------------------------------------
// Multi-tensor entry point: the same result as one dmxq_bfp_qdq call per tensor, in as few launches as possible.
namespace dmxq {
template <int DTI, int DTO>
static int launch_multi(const MultiArgs& a, int64_t total_tiles, bool asym, int wl, hipStream_t s) {
  constexpr int IVB = (Elem<DTO>::bytes > Elem<DTI>::bytes) ? 8 : 16;
  constexpr bool in16 = Elem<DTI>::bytes == 2;
  const bool single = in16 && bfp_single_rounding_ok<DTI>(wl);
  uint32_t e[kMultiPre];
  int64_t total_vec = 0;
  for (int i = 0; i < a.n; i++) total_vec += a.d[i].n_vec;
  for (int i = 0; i < kMultiPre; i++) e[i] = i + 1 < a.n ? (uint32_t)a.d[i + 1].tile0 : 0xFFFFFFFFu;
  const bool small_set = total_vec * IVB <= ((int64_t)32 << 20);   // (plain stores: the results are consumed at once and fit the Infinity Cache)
#define DMXQ_MULTI(A_, F_)                                                                                                                    \
  do {                                                                                                                                        \
    if (small_set)                                                                                                                            \
      DMXQ_LAUNCH((bfp_rows_multi_kernel<DTI, DTO, A_, kMultiUnroll, kMultiThreads, F_, IVB, kRowsNtLoad>), dim3((unsigned)total_tiles),       \
                  dim3(kMultiThreads), 0, s, e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7], e[8], e[9], a);                                   \
    else                                                                                                                                      \
      DMXQ_LAUNCH((bfp_rows_multi_kernel<DTI, DTO, A_, kMultiUnroll, kMultiThreads, F_, IVB>), dim3((unsigned)total_tiles), dim3(kMultiThreads), \
                  0, s, e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7], e[8], e[9], a);                                                        \
  } while (0)
  if (single) {
    if constexpr (in16) { if (asym) DMXQ_MULTI(true, 2); else DMXQ_MULTI(false, 2); }
  } else {
    if (asym) DMXQ_MULTI(true, 1); else DMXQ_MULTI(false, 1);
  }
#undef DMXQ_MULTI
  return launch_status();
}
}  // namespace dmxq

extern "C" int dmxq_bfp_qdq_multi(const dmxq_tensor_desc* tensors, int64_t n_tensors, int dtype_in, int dtype_out,
                                  int64_t block_size, int precision, int rounding, int symmetric, uint64_t seed,
                                  void* stream) {
  using namespace dmxq;
  if (n_tensors < 0 || (n_tensors > 0 && !tensors)) return DMXQ_ERR_BAD_ARG;
  if (!valid_dtype(dtype_in) || !valid_dtype(dtype_out) || !valid_rounding(rounding) || block_size < 1 || precision < 2)
    return DMXQ_ERR_BAD_ARG;
  for (int64_t i = 0; i < n_tensors; i++) {
    const dmxq_tensor_desc& t = tensors[i];
    if (t.outer < 0 || t.L < 0 || t.inner < 0) return DMXQ_ERR_BAD_ARG;
    if (t.outer * t.L * t.inner > 0 && (!t.in || !t.out)) return DMXQ_ERR_BAD_ARG;
  }
  if (block_size > 1 && precision > 22) return DMXQ_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const bool widening = dtype_in != DMXQ_F32 && dtype_out == DMXQ_F32;
  const int epl = (dtype_in == DMXQ_F32 || widening) ? 4 : 8;
  const bool pow2 = (block_size & (block_size - 1)) == 0;
  const bool fmt_ok = rounding == DMXQ_ROUND_NEAREST && precision <= 20 && block_size > 1 && pow2 && block_size >= epl &&
                      block_size <= 64 * epl;
  constexpr int64_t TILE = (int64_t)kMultiThreads * kMultiUnroll;
  MultiArgs a;
  a.n = 0; a.lpb = (int)(block_size / epl); a.wl = precision;
  int64_t tiles = 0;
  int rc = DMXQ_OK;
  auto flush = [&]() {
    if (a.n == 0) return;
    int r = DMXQ_ERR_BAD_ARG;
#define DMXQ_DT(I_, O_) if (dtype_in == I_ && dtype_out == O_) r = launch_multi<I_, O_>(a, tiles, !symmetric, precision, s);
    DMXQ_DT(DMXQ_BF16, DMXQ_BF16) DMXQ_DT(DMXQ_F16, DMXQ_F16) DMXQ_DT(DMXQ_F32, DMXQ_F32) DMXQ_DT(DMXQ_BF16, DMXQ_F32)
    DMXQ_DT(DMXQ_F16, DMXQ_F32) DMXQ_DT(DMXQ_F32, DMXQ_BF16) DMXQ_DT(DMXQ_F32, DMXQ_F16)
#undef DMXQ_DT
    if (r != DMXQ_OK) rc = r;
    a.n = 0; tiles = 0;
  };
  for (int64_t i = 0; i < n_tensors && rc == DMXQ_OK; i++) {
    const dmxq_tensor_desc& t = tensors[i];
    const int64_t n = t.outer * t.L * t.inner;
    if (n == 0) continue;
    const bool batch = fmt_ok && t.inner == 1 && t.L % block_size == 0 && aligned16(t.in) && aligned16(t.out) &&
                       (n / epl + TILE - 1) / TILE < ((int64_t)1 << 30);
    if (!batch) {  // ragged / strided / unaligned tensors and the other rounding modes: their own launch
      const int r = dmxq_bfp_qdq(t.in, t.out, dtype_in, dtype_out, t.outer, t.L, t.inner, block_size, precision, rounding,
                                 symmetric, seed + (uint64_t)i, stream);
      if (r != DMXQ_OK) rc = r;
      continue;
    }
    const int64_t nt = (n / epl + TILE - 1) / TILE;
    if (a.n == kMultiMax || tiles + nt >= ((int64_t)1 << 31)) flush();
    a.d[a.n] = MultiDesc{t.in, t.out, n / epl, tiles};
    a.n++;
    tiles += nt;
  }
  flush();
  return rc;
}

// What would dmxq_bfp_qdq launch for these arguments?  (bench.py reports it next to the roofline numbers.)
extern "C" int dmxq_bfp_qdq_describe(int dtype_in, int dtype_out, int64_t outer, int64_t L, int64_t inner,
                                     int64_t block_size, int precision, int rounding, int symmetric, int aligned,
                                     char* buf, int64_t buf_len) {
  using namespace dmxq;
  if (!buf || buf_len < 16 || !valid_dtype(dtype_in) || !valid_dtype(dtype_out) || !valid_rounding(rounding)) return DMXQ_ERR_BAD_ARG;
  if (outer < 0 || L < 0 || inner < 0 || block_size < 1 || precision < 2) return DMXQ_ERR_BAD_ARG;
  static const char* dn[] = {"f32", "f16", "bf16"};
  static const char* rn[] = {"up", "down", "nearest", "stochastic"};
  const int64_t n = outer * L * inner;
  const bool widening = dtype_in != DMXQ_F32 && dtype_out == DMXQ_F32;
  const int epl = (dtype_in == DMXQ_F32 || widening) ? 4 : 8;
  const bool pow2 = (block_size & (block_size - 1)) == 0;
  if (n == 0) { snprintf(buf, (size_t)buf_len, "none (empty tensor)"); return DMXQ_OK; }
  if (block_size == 1) { snprintf(buf, (size_t)buf_len, "dmxq::stream_kernel<FloatOp> (block size 1: float_quantize detour)"); return DMXQ_OK; }
  if (precision > 22) return DMXQ_ERR_UNSUPPORTED;
  if (inner == 1 && L % block_size == 0 && pow2 && block_size >= epl && block_size <= 64 * epl && aligned) {
    const bool nearest = rounding == DMXQ_ROUND_NEAREST && precision <= 20;
    const bool single = nearest && ((dtype_in == DMXQ_BF16 && precision <= 14) || (dtype_in == DMXQ_F16 && precision <= 11));
    // (round 5: down / up of symmetric formats are compile-time builds on the one-round plans; stochastic and asymmetric non-nearest stay on 512 x 2)
    const bool big = nearest || (symmetric && precision <= 20 && (rounding == DMXQ_ROUND_DOWN || rounding == DMXQ_ROUND_UP));
    const RowsPlan pl = rows_plan(n / epl, big, (nearest && symmetric && dtype_in == dtype_out) ? (single ? 20 : 16) : 0);
    const int64_t grid = pl.tiles < kRowsMaxGrid ? pl.tiles : kRowsMaxGrid;
    snprintf(buf, (size_t)buf_len, "dmxq::bfp_rows_kernel<%s,%s,%s,%s,%s> tile %dx%d vectors, grid %lld, nt loads+stores",
             dn[dtype_in], dn[dtype_out], nearest ? "nearest" : rn[rounding], symmetric ? "sym" : "asym",
             !nearest ? "literal+clamp" : (single ? "magic-add single rounding" : "magic-add double rounding"),
             pl.threads, pl.unroll, (long long)grid);
    return DMXQ_OK;
  }
  snprintf(buf, (size_t)buf_len, "%s", inner == 1 ? "dmxq::bfp_urows_kernel (ragged / unaligned rows)"
                                                  : "dmxq::bfp_slab_kernel (rows that are not whole lines, through the LDS), bfp_smallinner_kernel, bfp_cols_kernel or "
                                                    "bfp_generic_kernel (blocks along a strided dim)");
  return DMXQ_OK;
}
#endif  // DMXQ_BP(1)
