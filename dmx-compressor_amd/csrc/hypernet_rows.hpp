// csrc/hypernet_rows.hpp -- the per-unit arithmetic of the fused weight hypernet (mask -> SmoothQuant scale -> BFP), shared by
// hypernet.hip (one tensor per launch) and hypernet_multi.hip (the weights of a whole layer in one launch).  See hypernet.hip for
// the chain it reproduces (modeling/nn/core.py:178-198) and its dtype flow.
#pragma once
#include "bfp_math.hpp"
#include "bfp_rows.hpp"

namespace dmxq {

__device__ __forceinline__ int32_t hn_sort_key(float s) {  // same order as nm_mask.hip sort_key
  if (s != s) return 0x7FFFFFFF;
  if (s == 0.0f) return 0;
  const int32_t b = (int32_t)f2u(s);
  return b >= 0 ? b : (int32_t)(0x80000000u - (uint32_t)b);
}

template <int DT>
__device__ __forceinline__ void load8(const void* p, int64_t e, float (&v)[8]) {
  if (DT == DMXQ_F32) {
    const f32x4 a = *(const f32x4*)((const float*)p + e), b = *(const f32x4*)((const float*)p + e + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
    const u32x4 t = __builtin_nontemporal_load((const u32x4*)((const uint16_t*)p + e));
    widen<DT, 8>(t, v);
  }
}

template <int DT>
__device__ __forceinline__ float round_to(float v) {  // RNE to DT and back (exact for fp32)
  if (DT == DMXQ_BF16) return (float)(__bf16)v;
  if (DT == DMXQ_F16) return (float)(_Float16)opaque(v);
  return v;
}

struct HnArgs {
  const void* w; const void* score; const float* scale; void* out;
  int64_t n_units, L;
  int K, lpb, wl, asym;
  int small;        // n < 2^31: the column of a unit by magic-number division (common.hpp FastDiv31)
  FastDiv31 f_L;
};

// DTW weight dtype, DTS score dtype (ignored when M == 0), DTO output dtype, M = 0 (no mask) / 2 / 4 / 8
// BFP = false: mask-and-multiply only (y = x * mask, sparse.py:300) -- the typed fast path of dmxq_nm_mask for whole rows
// of 16-byte vectors (compile-time dtypes: every load of a lane is in flight before the first conversion; the generic
// kernel of nm_mask.hip switches on runtime dtypes around each access and reached 48 % of roofline with a bf16 score).
// LPBC: lanes per BFP block as a compile-time constant (0 = runtime a.lpb), see bfp_rows.hpp: the DPP block maximum is then
// free of scalar branches.
// DIVIDE: the ACTIVATION path of a SmoothQuant module (dmxq_input_hypernet): x / scale instead of w * scale, the quotient kept in
// torch's promotion of (x dtype, fp32 scale) = fp32, which is then also the dtype the BFP cast sees and hands back.
// UN units of a lane, `stride` units apart, starting at u0: all their loads are issued before the first one is ranked.
#ifndef DMXQ_HN_UNITS
#define DMXQ_HN_UNITS 4   // (tools/ab_hn_units.sh builds a second library with 8 for an A/B run)
#endif
constexpr int kHnUnits = DMXQ_HN_UNITS;
// ... and 2 for the masked BFP chain on everything but the largest sets (tools/ab_hn_units.sh, profiles/r04_ab_hypernet_units.txt: a rank's
// Llama-3-8B shard set at N = 8 32.2 -> 30.0 us, the seven single launches 228.7 -> 220.9 us; the whole 218 M-element layer in one launch
// 206.6 -> 211.4 us, so the multi-tensor launch keeps 4 above 160 M elements; 8 units per lane: 20-27 % slower everywhere)
#ifndef DMXQ_HN_UNITS_SMALL
#define DMXQ_HN_UNITS_SMALL 2
#endif
constexpr int kHnUnitsSmall = DMXQ_HN_UNITS_SMALL;
template <int DTW, int DTS, int DTO, int M, bool HAS_SCALE, bool BFP, int LPBC, bool ASYM, bool DIVIDE = false, int UNITS = kHnUnits>
__device__ __forceinline__ void hypernet_rows_units(const HnArgs& a, const int lpb, const int64_t u0, const int64_t stride) {
  // T1: dtype after the mask multiply = torch promotion of (w, score); without a mask it stays the weight dtype
  constexpr int T1 = DIVIDE ? DMXQ_F32 : ((M == 0) ? DTW : ((DTW == DMXQ_F32 || DTS == DMXQ_F32 || DTW != DTS) ? DMXQ_F32 : DTW));
  constexpr int UN = UNITS;
  {
    float xa[UN][8], sa[M != 0 ? UN : 1][8], sva[HAS_SCALE ? UN : 1][8];
#pragma unroll
    for (int r = 0; r < UN; r++) {
      const int64_t uc = u0 + r * stride < a.n_units ? u0 + r * stride : u0;  // clamped: unconditional loads
      load8<DTW>(a.w, uc * 8, xa[r]);
      if (M != 0) load8<DTS>(a.score, uc * 8, sa[M != 0 ? r : 0]);
      if (HAS_SCALE) {
        const int64_t ec = uc * 8;
        const int64_t c0 = a.small ? (int64_t)((uint32_t)ec - a.f_L.div((uint32_t)ec) * (uint32_t)a.L) : ec % a.L;
        const f32x4 s0 = *(const f32x4*)(a.scale + c0), s1 = *(const f32x4*)(a.scale + c0 + 4);
        sva[r][0] = s0.x; sva[r][1] = s0.y; sva[r][2] = s0.z; sva[r][3] = s0.w;
        sva[r][4] = s1.x; sva[r][5] = s1.y; sva[r][6] = s1.z; sva[r][7] = s1.w;
      }
    }
#pragma unroll
    for (int r = 0; r < UN; r++) {
    const int64_t u = u0 + r * stride;
    if (u >= a.n_units) break;
    const int64_t e0 = u * 8;
    float x[8];
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = xa[r][k];
    if (M != 0) {
      float s[8];
#pragma unroll
      for (int k = 0; k < 8; k++) s[k] = sa[M != 0 ? r : 0][k];
#pragma unroll
      for (int g = 0; g < 8; g += (M ? M : 8)) {
        int32_t key[M ? M : 1];
        int rank[M ? M : 1];
#pragma unroll
        for (int i = 0; i < M; i++) { key[i] = hn_sort_key(s[g + i]); rank[i] = 0; }
#pragma unroll
        for (int i = 0; i < M; i++)
#pragma unroll
          for (int j = 0; j < i; j++) {
            const bool j_first = key[j] <= key[i];
            rank[i] += j_first ? 1 : 0;
            rank[j] += j_first ? 0 : 1;
          }
#pragma unroll
        for (int i = 0; i < M; i++) x[g + i] = x[g + i] * (rank[i] >= M - a.K ? 1.0f : 0.0f);  // a real multiply: -w * 0 = -0
      }
    }
    if (HAS_SCALE) {
#pragma unroll
      for (int k = 0; k < 8; k++) x[k] = round_to<T1>(DIVIDE ? x[k] / sva[HAS_SCALE ? r : 0][k] : x[k] * sva[HAS_SCALE ? r : 0][k]);  // IEEE division (smoothquant.py:255-268)
    }
    if (!BFP) {
      store_vec<DTO, 8, true>(a.out, e0, x);
      continue;
    }
    uint32_t mb = 0u;
#pragma unroll
    for (int k = 0; k < 8; k++) mb = max(mb, f2u(x[k]) & 0x7FFFFFFFu);
    mb = group_max_u32(mb, lpb);
    float y[8];
    // the magic-add form for every lane, unconditionally (straight-line code the scheduler can interleave across the units in
    // flight); blocks it does not cover (denormal / huge maxima, bfp_math.hpp) are redone with the literal bit path behind one
    // cold wave-uniform branch -- the structure of bfp_rows.hpp.  (As a two-sided `if (all fast) ... else ...` per unit the
    // branch fenced the schedule.)
    const bool fast_ok = bfp_fast_ok(mb, a.wl);
    {
      const BfpBlockParams p = bfp_block_params<ASYM, true>(mb, a.wl);
#pragma unroll
      for (int k = 0; k < 8; k++) y[k] = bfp_q1_fast<false, ASYM>(x[k], p);
    }
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!fast_ok) != 0ull, 0)) {
      if (!fast_ok) {
        const BfpBlockParams p = bfp_block_params<ASYM, false>(mb, a.wl);
#pragma unroll
        for (int k = 0; k < 8; k++) y[k] = bfp_q1<DMXQ_ROUND_NEAREST, ASYM>(x[k], p, a.wl, DMXQ_ROUND_NEAREST, 0u);
      }
    }
#pragma unroll
    for (int k = 0; k < 8; k++) y[k] = round_to<T1>(y[k]);   // CastTo's `.to(physical_dtype)`, then the caller's dtype
    store_vec<DTO, 8, true>(a.out, e0, y);
    }
  }
}

template <int DTW, int DTS, int DTO, int M, bool HAS_SCALE, bool BFP, int LPBC, bool ASYM, bool DIVIDE = false, int UNITS = kHnUnits>
__device__ __forceinline__ void hypernet_rows_body(const HnArgs& a) {
  const int lpb = LPBC > 0 ? LPBC : __builtin_amdgcn_readfirstlane(a.lpb);
  constexpr int UN = UNITS;
  // workgroup-CONTIGUOUS tiles of kThreads x UN units (a grid-strided assignment, unit r of a lane a whole grid apart, cost the
  // hot kernel ~15 %: bfp_rows.hpp)
  // -- with a mask (two or three streams per unit): 2:4 + BFP 23.1 -> 22.5 us, the Llama-3-8B layer of bench.py 63 -> 69 %; the dense
  // scale + BFP path measured 5 % SLOWER that way (13.7 -> 14.4 us) and keeps the strided assignment.
  constexpr bool kContig = M != 0;
  const int64_t stride = kContig ? (int64_t)kThreads : (int64_t)gridDim.x * kThreads;
  for (int64_t k = 0;; k++) {
    const int64_t u0 = kContig ? ((int64_t)blockIdx.x + k * gridDim.x) * ((int64_t)kThreads * UN) + threadIdx.x
                               : (int64_t)blockIdx.x * kThreads + threadIdx.x + k * UN * stride;
    if (u0 >= a.n_units) break;  // (whole waves leave together: n_units is a multiple of the lanes of a block)
    hypernet_rows_units<DTW, DTS, DTO, M, HAS_SCALE, BFP, LPBC, ASYM, DIVIDE, UNITS>(a, lpb, u0, stride);
  }
}

}  // namespace dmxq
