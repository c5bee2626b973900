// csrc/bfp_cols.hip — BFP Q->DQ for blocks that run along a NON-contiguous dimension (inner > 1):
// block_dim = -2 (attention K/V multipliers, modeling/nn/torch_modules.py:197-204), conv activations/weights
// blocked along dim 1 (:582-585), i.e. the [outer, L, inner] view with inner > 1.
//
// The reference transposes the tensor so that the block dimension becomes last, runs its row algorithm and
// transposes back (numerical/format.py:322-341).  Here nothing is transposed: lanes run along the CONTIGUOUS
// inner dimension (16 B per lane, so a wave reads whole 1 KiB / 512 B / 256 B row segments), and each lane walks
// down the B rows of its block keeping the column tile in REGISTERS: one pass, every element read once and
// written once, all B row-loads of a lane in flight at once.
//   RPL = rows per lane (<= 32: 128 data VGPRs), RS = row split: the B = RPL*RS rows of a block are shared by RS lane groups of 64/RS
//   lanes (B = 64 -> two half-waves of 32 rows each, or eight groups of 8 lanes with 8 rows each: launch_cols); per-column maxima are
//   combined across the groups with lane permutes.  Every COLUMN is its own block, so a lane carries EPL block maxima.
// Arithmetic is bfp_math.hpp (magic-add nearest-even with the literal path as wave-uniform fallback).
#include "bfp_math.hpp"

namespace dmxq {

template <int RS>
__device__ __forceinline__ uint32_t split_max_u32(uint32_t m) {
  if (RS >= 2) m = max(m, (uint32_t)__shfl_xor((int)m, 32));
  if (RS >= 4) m = max(m, (uint32_t)__shfl_xor((int)m, 16));
  if (RS >= 8) m = max(m, (uint32_t)__shfl_xor((int)m, 8));
  if (RS >= 16) m = max(m, (uint32_t)__shfl_xor((int)m, 4));
  return m;
}

// per-element abs bit patterns (as fp32 bits) of one raw vector, max-accumulated into mb[EPL]
template <int DTI, int EPL>
__device__ __forceinline__ void accumulate_absmax(const u32x4& v, uint32_t (&mb)[EPL]) {
  if (DTI == DMXQ_F32) {
#pragma unroll
    for (int j = 0; j < 4; j++) mb[j] = max(mb[j], v[j] & 0x7FFFFFFFu);
  } else if (DTI == DMXQ_BF16) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      mb[2 * j] = max(mb[2 * j], (v[j] << 16) & 0x7FFF0000u);
      mb[2 * j + 1] = max(mb[2 * j + 1], v[j] & 0x7FFF0000u);
    }
  } else {  // fp16: half bit patterns order like magnitudes; widened after the reduction
#pragma unroll
    for (int j = 0; j < 4; j++) {
      mb[2 * j] = max(mb[2 * j], v[j] & 0x7FFFu);
      mb[2 * j + 1] = max(mb[2 * j + 1], (v[j] >> 16) & 0x7FFFu);
    }
  }
}

struct ColsIdx { FastDiv31 f_cvec, f_nblk, f_ctiles; int small; };  // host-made magic numbers for the unit decomposition

// UNAL: rows (inner elements) that are not whole aligned 16-byte vectors -- 14x14 or 7x7 feature maps, views that start
// mid-allocation.  Accesses are 16 bytes at element alignment (common.hpp load_raw16 / store_out UNAL), and the last,
// partial vector of a row is replaced by the vector that ENDS at the row end: every column is its own block, so the
// overlap with the previous lane just computes (and stores) those columns twice, with identical results.
template <int DTI, int DTO, int RND, bool ASYM, int RPL, int RS, int FAST, bool UNAL = false>
__global__ __launch_bounds__(kThreads) void bfp_cols_kernel(const void* __restrict__ in, void* __restrict__ out,
                                                           int64_t outer, int64_t L, int64_t inner, int wl,
                                                           int rounding, uint64_t seed, const ColsIdx ix) {
  constexpr int EPL = 16 / Elem<DTI>::bytes;
  constexpr int B = RPL * RS;
  constexpr int LPR = kWave / RS;  // lanes per row segment
  constexpr int OVB = EPL * Elem<DTO>::bytes;
  constexpr bool kFast = FAST != 0 && RND == DMXQ_ROUND_NEAREST;
  const bool stoch = (RND == kRuntimeRounding) && rounding == DMXQ_ROUND_STOCHASTIC;
  const int lane = threadIdx.x & (kWave - 1);
  const int grp = lane / LPR, lig = lane % LPR;  // row group, lane in group
  const int64_t nblk = (L + B - 1) / B;
  const int64_t cfull = inner / EPL;                      // whole 16-byte vectors per row
  const int64_t cvec = UNAL ? (inner + EPL - 1) / EPL : cfull;
  // one unit = one wave's [B rows x LPR lane-vectors].  Rows of at least LPR vectors: the LPR lanes take adjacent vectors
  // of ONE (outer, block) pair (tiles along the row).  Shorter rows (7x7 maps: 7 vectors): the lanes run over the
  // flattened (outer, block, vector) space instead, so that a wave is not mostly idle.
  const bool packed = cvec * 2 <= LPR;  // (at 25 of 32 lanes busy, 14x14 maps, the row tiles are still 25 % faster)
  const int64_t ctiles = (cvec + LPR - 1) / LPR;
  const int64_t total_cv = outer * nblk * cvec;
  const int64_t units = packed ? (total_cv + LPR - 1) / LPR : outer * nblk * ctiles;
  const int64_t wave_id = (int64_t)blockIdx.x * (kThreads / kWave) + threadIdx.x / kWave;
  const int64_t n_waves = (int64_t)gridDim.x * (kThreads / kWave);
  for (int64_t unit = wave_id; unit < units; unit += n_waves) {
    int64_t o, blk, cv;
    bool col_ok;
    // (unit -> (outer, block, column tile): magic-number divisions when everything fits 31 bits -- three 64-bit divisions here were ~450
    //  VALU operations in front of a unit's first load, as many as its 8 rows x 8 elements of arithmetic: tools/isa_prologue.py)
    if (packed) {
      const int64_t g = unit * LPR + lig;
      col_ok = g < total_cv;
      const int64_t gc = col_ok ? g : total_cv - 1;
      if (ix.small) {
        const uint32_t ob = ix.f_cvec.div((uint32_t)gc), o32 = ix.f_nblk.div(ob);
        cv = (uint32_t)gc - ob * (uint32_t)cvec;
        o = o32;
        blk = ob - o32 * (uint32_t)nblk;
      } else {
        const int64_t ob = gc / cvec;
        cv = gc - ob * cvec;
        o = ob / nblk;
        blk = ob - o * nblk;
      }
    } else {
      int64_t ct;
      if (ix.small) {
        const uint32_t q = ix.f_ctiles.div((uint32_t)unit), o32 = ix.f_nblk.div(q);
        ct = (uint32_t)unit - q * (uint32_t)ctiles;
        o = o32;
        blk = q - o32 * (uint32_t)nblk;
      } else {
        ct = unit % ctiles;
        blk = (unit / ctiles) % nblk;
        o = unit / (ctiles * nblk);
      }
      cv = ct * LPR + lig;                                // this lane's column vector
      col_ok = cv < cvec;
    }
    const int64_t row0 = blk * B + grp * RPL;             // first row of this lane's share of the block
    const int64_t ce = (UNAL && cv >= cfull) ? inner - EPL : cv * EPL;  // first column of this lane's vector
    const int64_t base_e = (o * L + row0) * inner + ce;
    u32x4 raw[RPL];
    uint32_t mb[EPL];
#pragma unroll
    for (int k = 0; k < EPL; k++) mb[k] = 0u;
#pragma unroll
    for (int r = 0; r < RPL; r++) {                        // all row loads in flight
      const bool ok = col_ok && row0 + r < L;              // ragged last block: rows beyond L count as absent
      raw[r] = ok ? load_raw16<true, int64_t, UNAL>(in, (base_e + r * inner) * Elem<DTI>::bytes) : u32x4{0u, 0u, 0u, 0u};
      // (blocks of 64+ rows: 16 idle issue cycles between a lane's row loads -- rows are `inner` elements apart, the pattern of lastdim_kernel;
      //  [4096, 4096] bf16 along dim -2, B = 64: 14.09 -> 13.32 us, pace 1 / 4: 13.43 / 13.83; B = 16 (RS = 2): 11.58 -> 11.68: stays unpaced)
      if (r + 1 < RPL) pace_issue<(RPL * RS >= 64) ? 2 : 0>();
    }
#pragma unroll
    for (int r = 0; r < RPL; r++) accumulate_absmax<DTI, EPL>(raw[r], mb);
    uint32_t mfin[EPL];
    bool all_fast = kFast;
#pragma unroll
    for (int k = 0; k < EPL; k++) {
      uint32_t m = split_max_u32<RS>(mb[k]);
      if (DTI == DMXQ_F16) m = f2u((float)__builtin_bit_cast(_Float16, (uint16_t)m));
      if (kFast) all_fast = all_fast && bfp_fast_ok(m, wl);
      mfin[k] = m;
    }
    char* const obase = (char*)out + base_e * Elem<DTO>::bytes;
    const int64_t ostride = inner * Elem<DTO>::bytes;
    if (kFast && __builtin_amdgcn_ballot_w64(!all_fast) == 0ull) {  // wave-uniform
      BfpBlockParams p[EPL];
#pragma unroll
      for (int k = 0; k < EPL; k++) p[k] = bfp_block_params<ASYM, true>(mfin[k], wl);
#pragma unroll
      for (int r = 0; r < RPL; r++) {
        float x[EPL], y[EPL];
        widen<DTI, EPL>(raw[r], x);
#pragma unroll
        for (int k = 0; k < EPL; k++) y[k] = bfp_q1_fast<FAST == 2, ASYM>(x[k], p[k]);
        if (col_ok && row0 + r < L) store_out<DTO, EPL, true, UNAL>(obase + r * ostride, pack_vec<DTO, EPL>(y));
      }
    } else {
      BfpBlockParams p[EPL];
#pragma unroll
      for (int k = 0; k < EPL; k++) p[k] = bfp_block_params<ASYM, false>(mfin[k], wl);
#pragma unroll
      for (int r = 0; r < RPL; r++) {
        float x[EPL], y[EPL];
        widen<DTI, EPL>(raw[r], x);
#pragma unroll
        for (int k = 0; k < EPL; k++) {
          // the oracle numbers random draws by position in the transposed [outer*inner, L] matrix
          const uint64_t ridx = (uint64_t)((o * inner + ce + k) * L + row0 + r);
          y[k] = bfp_q1<RND, ASYM>(x[k], p[k], wl, rounding, bfp_rnd_if(stoch, seed, ridx));
        }
        if (col_ok && row0 + r < L) store_out<DTO, EPL, true, UNAL>(obase + r * ostride, pack_vec<DTO, EPL>(y));
      }
    }
  }
}

template <int DTI, int DTO, int RND, bool ASYM, int RPL, int RS>
static int launch_cols_geom(const void* in, void* out, int64_t outer, int64_t L, int64_t inner, int wl, int rounding,
                            uint64_t seed, bool unal, hipStream_t s) {
  constexpr int EPL = 16 / Elem<DTI>::bytes;
  constexpr int B = RPL * RS, LPR = kWave / RS;
  const int64_t nblk = (L + B - 1) / B, cvec = (inner + EPL - 1) / EPL, ctiles = (cvec + LPR - 1) / LPR;
  const int64_t units = cvec * 2 <= LPR ? (outer * nblk * cvec + LPR - 1) / LPR : outer * nblk * ctiles;
  int64_t grid = (units + 3) / 4;
  if (grid < 1) grid = 1;
  if (grid > (1 << 20)) grid = 1 << 20;
  const int fast = (RND == DMXQ_ROUND_NEAREST && wl <= 20) ? (bfp_single_rounding_ok<DTI>(wl) ? 2 : 1) : 0;
  const int64_t total = outer * nblk * (cvec * 2 <= LPR ? cvec : ctiles * LPR);
  const ColsIdx ix{make_fastdiv31(cvec), make_fastdiv31(nblk), make_fastdiv31(ctiles), (total < ((int64_t)1 << 31) && units * LPR < ((int64_t)1 << 31)) ? 1 : 0};
#define DMXQ_COLS(F_)                                                                                             \
  do {                                                                                                            \
    if (unal)                                                                                                     \
      DMXQ_LAUNCH((bfp_cols_kernel<DTI, DTO, RND, ASYM, RPL, RS, F_, true>), dim3((unsigned)grid), dim3(kThreads), 0, \
                         s, in, out, outer, L, inner, wl, rounding, seed, ix);                                    \
    else                                                                                                          \
      DMXQ_LAUNCH((bfp_cols_kernel<DTI, DTO, RND, ASYM, RPL, RS, F_, false>), dim3((unsigned)grid), dim3(kThreads), 0, \
                         s, in, out, outer, L, inner, wl, rounding, seed, ix);                                    \
  } while (0)
  // instantiate only what can run: the literal path for the runtime-rounding build; magic-add (double / single
  // rounding) for nearest-even.  (nearest with wl > 20 is routed to the runtime-rounding build by the caller.)
  constexpr bool in16 = Elem<DTI>::bytes == 2;
  if constexpr (RND == kRuntimeRounding) {
    DMXQ_COLS(0);
  } else {
    if (in16 && fast == 2) {
      if constexpr (in16) DMXQ_COLS(2);
    } else {
      DMXQ_COLS(1);
    }
  }
#undef DMXQ_COLS
  return launch_status();
}

template <int DTI, int DTO, int RND, bool ASYM>
static int launch_cols(const void* in, void* out, int64_t outer, int64_t L, int64_t inner, int64_t B, int wl,
                       int rounding, uint64_t seed, bool unal, hipStream_t s) {
#define DMXQ_G(RPL_, RS_) return launch_cols_geom<DTI, DTO, RND, ASYM, RPL_, RS_>(in, out, outer, L, inner, wl, rounding, seed, unal, s)
  // B = 64 on rows that are whole 128-byte lines (round 3): 8 rows per lane, the 8 row groups of a block side by side in the wave, instead of
  // 32 rows per lane in two half-waves.  A lane's serial share -- its loads, then rows x 8 elements of arithmetic -- is what a small
  // tensor waits for, and the attention operands this path serves ARE small (Llama V [1, 32, 128, 128] bf16: 8.5 -> 4.6 us, Whisper
  // [12, 1500, 64] float32: 7.3 -> 4.4 us, opt-125m [24, 128, 64]: 6.3 -> 3.6 us per launch inside a graph); 134 MB of them: 53 -> 46 us
  // (64 -> 73 %); [64, 256, 56, 56] along channels 44 -> 37 us.  A row group then reads 128 contiguous bytes per row: rows that are NOT whole
  // lines (28 x 28, 14 x 14 maps) straddle two lines per piece and measured 14-24 % slower that way, so they keep the wide groups; so do the
  // other block sizes (B = 16 as 8 x 2 instead of 16 x 1: 11.7 -> 13.1 us on 32 MiB).  4 rows per lane: equal on small tensors, 15 % slower
  // on the big one.  (tools/bench_conv_shapes.py, profiles/r03_conv_shapes.txt)
  switch (B) {
    case 8: DMXQ_G(8, 1);
    case 16: DMXQ_G(16, 1);
    case 32: DMXQ_G(32, 1);
    case 64:
      if (!unal && (inner * (int64_t)Elem<DTI>::bytes) % 128 == 0) DMXQ_G(8, 8);
      DMXQ_G(32, 2);
    case 128: DMXQ_G(32, 4);
  }
#undef DMXQ_G
  return DMXQ_ERR_UNSUPPORTED;
}

}  // namespace dmxq

using namespace dmxq;

// This file is compiled THREE times (build.py: -DDMXQ_EW_PART=1 / 2 / 3), one object per input dtype: the cross product of
// 5 dtype pairs x 2 rounding builds x sym / asym x 5 block sizes x aligned / unaligned x fast paths was a 4-minute
// translation unit, the longest of the library.  Without the macro everything is compiled into one object.
#ifndef DMXQ_EW_PART
#define DMXQ_EW_PART 0
#endif
#define DMXQ_CP(P_) (DMXQ_EW_PART == 0 || DMXQ_EW_PART == (P_))

#define DMXQ_COLS_ARGS const void* in, void* out, int dtype_out, int64_t outer, int64_t L, int64_t inner, int64_t B, int wl, int rounding, \
                       bool asym, uint64_t seed, bool unal, hipStream_t s
#define DMXQ_DT(I_, O_)                                                                                           \
  if (dtype_out == O_) {                                                                                          \
    if (rounding == DMXQ_ROUND_NEAREST && wl <= 20)                                                               \
      return asym ? launch_cols<I_, O_, DMXQ_ROUND_NEAREST, true>(in, out, outer, L, inner, B, wl, rounding, seed, unal, s)  \
                  : launch_cols<I_, O_, DMXQ_ROUND_NEAREST, false>(in, out, outer, L, inner, B, wl, rounding, seed, unal, s); \
    return asym ? launch_cols<I_, O_, kRuntimeRounding, true>(in, out, outer, L, inner, B, wl, rounding, seed, unal, s)     \
                : launch_cols<I_, O_, kRuntimeRounding, false>(in, out, outer, L, inner, B, wl, rounding, seed, unal, s);   \
  }
int dmxq_cols_from_bf16(DMXQ_COLS_ARGS);
int dmxq_cols_from_f16(DMXQ_COLS_ARGS);
int dmxq_cols_from_f32(DMXQ_COLS_ARGS);
#if DMXQ_CP(1)
int dmxq_cols_from_bf16(DMXQ_COLS_ARGS) {
  DMXQ_DT(DMXQ_BF16, DMXQ_BF16)
  DMXQ_DT(DMXQ_BF16, DMXQ_F32)
  return DMXQ_ERR_UNSUPPORTED;
}
#endif
#if DMXQ_CP(2)
int dmxq_cols_from_f16(DMXQ_COLS_ARGS) {
  DMXQ_DT(DMXQ_F16, DMXQ_F16)
  DMXQ_DT(DMXQ_F16, DMXQ_F32)
  return DMXQ_ERR_UNSUPPORTED;
}
#endif
#if DMXQ_CP(3)
int dmxq_cols_from_f32(DMXQ_COLS_ARGS) {
  DMXQ_DT(DMXQ_F32, DMXQ_F32)
  return DMXQ_ERR_UNSUPPORTED;
}

// internal entry used by dmxq_bfp_qdq (bfp.hip): returns DMXQ_ERR_UNSUPPORTED when this path does not apply, in
// which case the caller falls back to the generic kernel.
extern "C" int dmxq_internal_bfp_cols(const void* in, void* out, int dtype_in, int dtype_out, int64_t outer, int64_t L,
                                      int64_t inner, int64_t B, int wl, int rounding, int symmetric, uint64_t seed,
                                      void* stream) {
  const int epl = dtype_in == DMXQ_F32 ? 4 : 8;
  if (wl > 22) return DMXQ_ERR_UNSUPPORTED;
  // rows that are whole aligned vectors: aligned form.  Anything else with at least one whole vector per row: the
  // unaligned form -- not in place (its overlapping tail vectors may belong to different waves), and element-aligned
  const bool unal = inner % epl != 0 || !aligned16(in) || !aligned16(out);
  if (unal) {
    const uintptr_t ib = dtype_in == DMXQ_F32 ? 4 : 2, ob = dtype_out == DMXQ_F32 ? 4 : 2;
    if (inner < epl || in == out || (reinterpret_cast<uintptr_t>(in) & (ib - 1)) || (reinterpret_cast<uintptr_t>(out) & (ob - 1)))
      return DMXQ_ERR_UNSUPPORTED;
  }
  if (!(B == 8 || B == 16 || B == 32 || B == 64 || B == 128)) return DMXQ_ERR_UNSUPPORTED;
  // fewer rows than half a block (RGB input of a first conv layer: L = 3): there is exactly one, ragged, block per
  // column whatever the nominal size, so run the smallest tile that still holds it instead of mostly absent rows
  while (B > 8 && L <= B / 2) B /= 2;
  hipStream_t s = (hipStream_t)stream;
  const bool asym = !symmetric;
  if (dtype_in == DMXQ_BF16) return dmxq_cols_from_bf16(in, out, dtype_out, outer, L, inner, B, wl, rounding, asym, seed, unal, s);
  if (dtype_in == DMXQ_F16) return dmxq_cols_from_f16(in, out, dtype_out, outer, L, inner, B, wl, rounding, asym, seed, unal, s);
  if (dtype_in == DMXQ_F32) return dmxq_cols_from_f32(in, out, dtype_out, outer, L, inner, B, wl, rounding, asym, seed, unal, s);
  return DMXQ_ERR_UNSUPPORTED;
}
#endif
#undef DMXQ_DT
