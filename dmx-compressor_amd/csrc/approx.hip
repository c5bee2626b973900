// csrc/approx.hip — approximator-slot ops (GELU / Softmax / LayerNorm) for gfx950.
//
// In the reference these modules compute the exact torch.nn.functional result first
// (functional/approximate.py:300-304) and then overwrite it with a vsimd approximation from a private
// package that is not part of the repository (approximate.py:9-14, 145-147).  With vsimd absent — the public
// state of the reference — every approximator is NONE and the exact function IS the result.  These kernels
// implement that exact-function contract in fp32 (one read, one write per element); approximation
// arithmetic is PARITY-UNPINNED and is not invented here.
//
//   gelu       elementwise, erf or tanh form (GeluOp in elementwise.hip)  (torch.nn.functional.gelu)
//   softmax    over the contiguous last dim, optional input clamp  (modeling/nn/torch_modules.py:989-994)
//   layernorm  over the contiguous last dim, affine optional       (modeling/nn/torch_modules.py:1062-1082)
// Row kernels, 1 read + 1 write of HBM per element in every case:
//   * register-resident (the product path: same dtype in and out, rows a multiple of the vector width): a row per
//     64 or 32 lanes for up to 256 lane-vectors, a row per 256-thread workgroup beyond (LayerNorm); see below;
//   * fallback for everything else (odd lengths, mixed dtypes): one workgroup per row, the row staged ONCE in LDS as
//     fp32 (gfx950 has 160 KiB per CU), wave shuffles + a 4-entry LDS exchange; rows too long for LDS take a 3-pass
//     global version.
#include <atomic>
#include <math.h>

#include <type_traits>

#include "floatq.hpp"
#include "bfp_math.hpp"

namespace dmxq {

constexpr int kRowLdsFloats = 16 * 1024;  // 64 KiB per workgroup -> 2 workgroups per CU

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_maxf(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
// exp(x - m) for fp32 OUTPUTS.  fl(x - m) alone carries up to half an ulp of |x - m|, which exp turns into a RELATIVE
// error of that size: |x - m| / 2 ulps of the result (10+ ulps in the tail of an attention row; torch's CPU softmax has
// the same loss).  The rounding error of the subtraction is recovered exactly (TwoSum) and applied as
// exp(d + e) = exp(d) (1 + e): 8 more VALU operations per element on a path that moves 8 bytes per element.
__device__ __forceinline__ float exp_diff(float x, float m) {
  const float d = x - m;
  if (!(fabsf(d) < INFINITY)) return expf(d);  // -inf inputs (masks), NaN: nothing to correct
  const float bb = d - x;
  const float e = (x - (d - bb)) + (-m - bb);  // x - m == d + e exactly
  const float r = expf(d);
  return __builtin_fmaf(r, e, r);
}

// workgroup all-reduce through LDS scratch (kThreads/kWave floats)
template <bool IS_MAX>
__device__ __forceinline__ float block_allreduce(float v, float* scratch) {
  v = IS_MAX ? wave_maxf(v) : wave_sum(v);
  const int w = threadIdx.x / kWave;
  __syncthreads();  // scratch reuse
  if ((threadIdx.x & (kWave - 1)) == 0) scratch[w] = v;
  __syncthreads();
  float r = scratch[0];
#pragma unroll
  for (int i = 1; i < kThreads / kWave; i++) r = IS_MAX ? fmaxf(r, scratch[i]) : r + scratch[i];
  return r;
}

template <bool LDS_ROW>
__global__ __launch_bounds__(kThreads) void softmax_rows_kernel(const void* __restrict__ in, void* __restrict__ out,
                                                               int dti, int dto, int64_t rows, int64_t cols,
                                                               float clamp_min) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* scratch = smem;             // kThreads / kWave
  float* row = smem + kThreads / kWave;
  for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
    const int64_t base = r * cols;
    float m = -INFINITY;
    for (int64_t c = threadIdx.x; c < cols; c += kThreads) {
      float v = load_rt(in, dti, base + c);
      v = fmaxf(v, clamp_min);  // torch.clamp(x, min=input_clamp); clamp_min = -inf disables
      if (LDS_ROW) row[c] = v;
      m = fmaxf(m, v);
    }
    m = block_allreduce<true>(m, scratch);
    float s = 0.0f;
    for (int64_t c = threadIdx.x; c < cols; c += kThreads) {
      const float v = LDS_ROW ? row[c] : fmaxf(load_rt(in, dti, base + c), clamp_min);
      const float e = exp_diff(v, m);
      if (LDS_ROW) row[c] = e;
      s += e;
    }
    s = block_allreduce<false>(s, scratch);
    for (int64_t c = threadIdx.x; c < cols; c += kThreads) {
      const float e = LDS_ROW ? row[c] : exp_diff(fmaxf(load_rt(in, dti, base + c), clamp_min), m);
      store_rt(out, dto, base + c, e / s);
    }
    __syncthreads();
  }
}

template <bool LDS_ROW, bool RMS = false>
__global__ __launch_bounds__(kThreads) void layernorm_rows_kernel(const void* __restrict__ in, void* __restrict__ out,
                                                                 int dti, int dto, int64_t rows, int64_t cols,
                                                                 const void* __restrict__ w,
                                                                 const void* __restrict__ b, int dtw, float eps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* scratch = smem;
  float* row = smem + kThreads / kWave;
  for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
    const int64_t base = r * cols;
    float s = 0.0f;
    for (int64_t c = threadIdx.x; c < cols; c += kThreads) {
      const float v = load_rt(in, dti, base + c);
      if (LDS_ROW) row[c] = v;
      s += v;
    }
    const float mean = RMS ? 0.0f : block_allreduce<false>(s, scratch) / (float)cols;  // RMSNorm: no centring
    float q = 0.0f;
    for (int64_t c = threadIdx.x; c < cols; c += kThreads) {
      const float d = (LDS_ROW ? row[c] : load_rt(in, dti, base + c)) - mean;
      q += d * d;
    }
    const float var = block_allreduce<false>(q, scratch) / (float)cols;  // biased, as F.layer_norm
    const float rstd = 1.0f / sqrtf(var + eps);
    for (int64_t c = threadIdx.x; c < cols; c += kThreads) {
      float y = ((LDS_ROW ? row[c] : load_rt(in, dti, base + c)) - mean) * rstd;
      if (w) y *= load_rt(w, dtw, c);
      if (b) y += load_rt(b, dtw, c);
      store_rt(out, dto, base + c, y);
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------
// Register-resident row kernels (same dtype in and out, cols % EPL == 0).  A row is owned by LPR = 64 or 32 adjacent
// lanes (32: two rows side by side in a wave, for rows whose vector count is an odd multiple of 32 -- 768 bf16
// elements = 96 vectors = 3 per lane of a half wave, no idle lanes), as VPL lane-vectors of EPL elements (16-byte
// accesses; 8-byte ones when a 16-bit row is only a multiple of 4 elements); RPW row slots are processed together.
// Everything is compile-time typed, so the raw loads of all RPW x VPL vectors are issued back to back before the
// first conversion (a runtime dtype switch around each load made the wave wait for every load in turn).  Lanes past
// the row end and rows past the tensor end read a clamped in-bounds address instead of being predicated; such
// duplicates cannot change a row maximum and are dropped from the sums with ONE select per vector.  Reductions are
// wave shuffles within the LPR lanes: no LDS, no workgroup barrier.  These kernels were VALU-bound before they were
// memory-bound (profiles/: 16 and 26 VALU instructions per element), hence the care about instruction counts.
template <int DT, int EPL>
struct RowVec {
  static constexpr int kWords = EPL * Elem<DT>::bytes / 4;  // 2 (8 bytes) or 4 (16 bytes)
  uint32_t w[kWords];
};
template <int DT, int EPL>
__device__ __forceinline__ RowVec<DT, EPL> row_load(const void* p, int64_t e) {
  RowVec<DT, EPL> r;
  const char* a = (const char*)p + e * Elem<DT>::bytes;
  // non-temporal: every row is read once and written once (weight / bias vectors go through row_load_keep)
  if (RowVec<DT, EPL>::kWords == 4) { const u32x4 t = __builtin_nontemporal_load((const u32x4*)a); r.w[0] = t.x; r.w[1] = t.y; r.w[2] = t.z; r.w[3] = t.w; }
  else { const u32x2 t = __builtin_nontemporal_load((const u32x2*)a); r.w[0] = t.x; r.w[1] = t.y; }
  return r;
}
template <int DT, int EPL>
__device__ __forceinline__ RowVec<DT, EPL> row_load_keep(const void* p, int64_t e) {  // re-used data (weight, bias): cached
  RowVec<DT, EPL> r;
  const char* a = (const char*)p + e * Elem<DT>::bytes;
  if (RowVec<DT, EPL>::kWords == 4) { const u32x4 t = *(const u32x4*)a; r.w[0] = t.x; r.w[1] = t.y; r.w[2] = t.z; r.w[3] = t.w; }
  else { const u32x2 t = *(const u32x2*)a; r.w[0] = t.x; r.w[1] = t.y; }
  return r;
}
template <int DT, int EPL>
__device__ __forceinline__ void row_widen(const RowVec<DT, EPL>& r, float (&x)[EPL]) {
#pragma unroll
  for (int j = 0; j < RowVec<DT, EPL>::kWords; j++) {
    if (DT == DMXQ_F32) x[j] = u2f(r.w[j]);
    else if (DT == DMXQ_BF16) { x[2 * j] = u2f(r.w[j] << 16); x[2 * j + 1] = u2f(r.w[j] & 0xFFFF0000u); }
    else { x[2 * j] = half_lo(r.w[j]); x[2 * j + 1] = half_hi(r.w[j]); }
  }
}
template <int DT, int EPL>
__device__ __forceinline__ void row_store(void* p, int64_t e, const float (&y)[EPL]) {
  char* a = (char*)p + e * Elem<DT>::bytes;
  if (DT == DMXQ_F32) __builtin_nontemporal_store(f32x4{y[0], y[1], y[2], y[3]}, (f32x4*)a);
  else if (EPL == 8) __builtin_nontemporal_store(u32x4{pack2<DT>(y[0], y[1]), pack2<DT>(y[2], y[3]), pack2<DT>(y[4], y[5]), pack2<DT>(y[6], y[7])}, (u32x4*)a);
  else __builtin_nontemporal_store(u32x2{pack2<DT>(y[0], y[1]), pack2<DT>(y[2], y[3])}, (u32x2*)a);
}
// RAG variants: rows of ANY length / alignment (attention rows of 1500 or 197 elements): whole lane-vectors are read and
// written with 16-byte accesses at element alignment (93-97 % of the aligned rate on gfx950,
// profiles/r01_unaligned_access.txt); the partial last vector of a row is moved element by element by the one lane that
// owns it.
typedef u32x4 u32x4_a2 __attribute__((aligned(2)));
template <int DT, int EPL>
__device__ __forceinline__ RowVec<DT, EPL> row_load_u(const void* p, int64_t e) {
  static_assert(RowVec<DT, EPL>::kWords == 4, "16-byte lane-vectors");
  const u32x4 t = *(const u32x4_a2*)((const char*)p + e * Elem<DT>::bytes);
  RowVec<DT, EPL> r;
  r.w[0] = t.x; r.w[1] = t.y; r.w[2] = t.z; r.w[3] = t.w;
  return r;
}
template <int DT, int EPL>
__device__ __forceinline__ void row_store_u(void* p, int64_t e, const float (&y)[EPL]) {
  char* a = (char*)p + e * Elem<DT>::bytes;
  if (DT == DMXQ_F32) *(u32x4_a2*)a = u32x4{f2u(y[0]), f2u(y[1]), f2u(y[2]), f2u(y[3])};
  else *(u32x4_a2*)a = u32x4{pack2<DT>(y[0], y[1]), pack2<DT>(y[2], y[3]), pack2<DT>(y[4], y[5]), pack2<DT>(y[6], y[7])};
}
template <int DT, int EPL>
__device__ __forceinline__ RowVec<DT, EPL> row_load_tail(const void* p, int64_t e, int t) {  // first t elements, rest zero
  RowVec<DT, EPL> r;
#pragma unroll
  for (int j = 0; j < RowVec<DT, EPL>::kWords; j++) r.w[j] = 0u;
#pragma unroll
  for (int k = 0; k < EPL; k++) {
    if (k < t) {
      if (DT == DMXQ_F32) r.w[k] = ((const uint32_t*)p)[e + k];
      else r.w[k / 2] |= (uint32_t)((const uint16_t*)p)[e + k] << (16 * (k & 1));
    }
  }
  return r;
}

// The casts of a DmxModule around a row function (dmxq_softmax_cast / dmxq_layernorm_cast / dmxq_rmsnorm_cast): out =
// cast_out(f(cast_in(x))) in one pass.  16-bit rows: both casts are range-only (common.hpp range16_of) and act on the packed words
// right after the load / right before the store; float32 rows: any nearest-rounding FloatingPoint format, per element (floatq.hpp).
struct RowCast { Range16 ri, ro; CastG gi, go; int bfp_B = 0, bfp_wl = 0, bfp_lpb = 0; };  // bfp_*: the CONSUMER's BFP input cast, dmxq_softmax_cast_bfp
struct NoRowCast {};
template <bool CAST> using RowCastArg = std::conditional_t<CAST, RowCast, NoRowCast>;
template <bool CAST, int DT, int EPL, class RC>
__device__ __forceinline__ void rowcast_raw_in(RowVec<DT, EPL>& r, const RC& rc) {
  if constexpr (CAST && DT != DMXQ_F32) {
#pragma unroll
    for (int j = 0; j < RowVec<DT, EPL>::kWords; j++) r.w[j] = range16_word(r.w[j], rc.ri);
  }
}
template <bool CAST, int DT, int EPL, class RC>
__device__ __forceinline__ void rowcast_x_in(float (&x)[EPL], const RC& rc) {
  if constexpr (CAST && DT == DMXQ_F32) castg_vec<DT, EPL>(x, rc.gi);
}
// y (fp32 results of one lane-vector) -> the words that are stored: rounded once to the row dtype, then the output cast
template <bool CAST, int DT, int EPL, class RC>
__device__ __forceinline__ RowVec<DT, EPL> rowcast_pack_out(float (&y)[EPL], const RC& rc) {
  RowVec<DT, EPL> r;
  if constexpr (DT == DMXQ_F32) {
    if constexpr (CAST) castg_vec<DT, EPL>(y, rc.go);
#pragma unroll
    for (int j = 0; j < EPL; j++) r.w[j] = f2u(y[j]);
  } else {
    // (a NaN result keeps the hardware's sign: see act_cast.hip)
#pragma unroll
    for (int j = 0; j < EPL / 2; j++) {
      r.w[j] = pack2<DT>(y[2 * j], y[2 * j + 1]);
      if constexpr (CAST) r.w[j] = range16_word(r.w[j], rc.ro);
    }
  }
  return r;
}
template <int DT, int EPL, bool UNAL = false>
__device__ __forceinline__ void row_store_raw(void* p, int64_t e, const RowVec<DT, EPL>& r) {
  char* a = (char*)p + e * Elem<DT>::bytes;
  if constexpr (RowVec<DT, EPL>::kWords == 4) {
    const u32x4 v = {r.w[0], r.w[1], r.w[2], r.w[3]};
    if constexpr (UNAL) *(u32x4_a2*)a = v; else __builtin_nontemporal_store(v, (u32x4*)a);
  } else {
    __builtin_nontemporal_store(u32x2{r.w[0], r.w[1]}, (u32x2*)a);
  }
}

// The CONSUMER's BFP input cast applied to one lane-vector of a module's result (dmxq_*_cast_bfp): `o` is what the module returns
// (output cast applied, rounded to the row dtype); the block maximum runs over lpb adjacent lanes -- EVERY lane of the wave must call
// this (`valid` = the slot lies inside the row; invalid slots contribute 0) --, then the magic-add arithmetic of bfp_math.hpp with
// its cold literal redo.  q: the values to store (row_store rounds them to the row dtype: CastTo's `.to(dtype)`).
template <int DT, int EPL>
__device__ __forceinline__ void bfp_epilogue(const RowVec<DT, EPL>& o, bool valid, int lpb, int wl, float (&q)[EPL]) {
  float z[EPL];
  row_widen<DT, EPL>(o, z);
  uint32_t mb = 0u;
#pragma unroll
  for (int k = 0; k < EPL; k++) mb = max(mb, f2u(z[k]) & 0x7FFFFFFFu);
  mb = group_max_u32(valid ? mb : 0u, lpb);
  const bool fast_ok = bfp_fast_ok(mb, wl);
  {
    const BfpBlockParams bp = bfp_block_params<false, true>(mb, wl);
#pragma unroll
    for (int k = 0; k < EPL; k++) q[k] = bfp_q1_fast<false, false>(z[k], bp);
  }
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(!fast_ok) != 0ull, 0)) {
    if (!fast_ok) {
      const BfpBlockParams bp = bfp_block_params<false, false>(mb, wl);
#pragma unroll
      for (int k = 0; k < EPL; k++) q[k] = bfp_q1<DMXQ_ROUND_NEAREST, false>(z[k], bp, wl, DMXQ_ROUND_NEAREST, 0u);
    }
  }
}

// Sum / max over the LPR (64 or 32) adjacent lanes that own a row, every lane receiving the result.  Inside a row of 16 lanes: four
// DPP steps (quad_perm xor 1, xor 2, row_half_mirror, row_mirror -- VALU operand modifiers, no LDS); across the rows of 16: the four
// row totals through v_readlane.  The SAME summation tree as an xor butterfly (so the same bits), which as `__shfl_xor` compiled
// to five or six dependent ds_bpermute_b32 round trips through the LDS per reduction, two or three reductions per row.
#define DMXQ_DPP_F32(v_, ctrl_) u2f((uint32_t)__builtin_amdgcn_update_dpp(0, (int)f2u(v_), ctrl_, 0xF, 0xF, false))
template <int LPR>
__device__ __forceinline__ float seg_sum(float v) {
  static_assert(LPR == 64 || LPR == 32, "rows are owned by 64 or 32 lanes");
  v += DMXQ_DPP_F32(v, 0xB1);
  v += DMXQ_DPP_F32(v, 0x4E);
  v += DMXQ_DPP_F32(v, 0x141);
  v += DMXQ_DPP_F32(v, 0x140);
  const int b = (int)f2u(v);
  const float r0 = u2f((uint32_t)__builtin_amdgcn_readlane(b, 0)), r1 = u2f((uint32_t)__builtin_amdgcn_readlane(b, 16));
  const float r2 = u2f((uint32_t)__builtin_amdgcn_readlane(b, 32)), r3 = u2f((uint32_t)__builtin_amdgcn_readlane(b, 48));
  const float lo = r0 + r1, hi = r2 + r3;
  if (LPR == 64) return lo + hi;
  return (threadIdx.x & 32) ? hi : lo;
}
template <int LPR>
__device__ __forceinline__ float seg_max(float v) {
  static_assert(LPR == 64 || LPR == 32, "rows are owned by 64 or 32 lanes");
  v = fmaxf(v, DMXQ_DPP_F32(v, 0xB1));
  v = fmaxf(v, DMXQ_DPP_F32(v, 0x4E));
  v = fmaxf(v, DMXQ_DPP_F32(v, 0x141));
  v = fmaxf(v, DMXQ_DPP_F32(v, 0x140));
  const int b = (int)f2u(v);
  const float r0 = u2f((uint32_t)__builtin_amdgcn_readlane(b, 0)), r1 = u2f((uint32_t)__builtin_amdgcn_readlane(b, 16));
  const float r2 = u2f((uint32_t)__builtin_amdgcn_readlane(b, 32)), r3 = u2f((uint32_t)__builtin_amdgcn_readlane(b, 48));
  const float lo = fmaxf(r0, r1), hi = fmaxf(r2, r3);
  if (LPR == 64) return fmaxf(lo, hi);
  return (threadIdx.x & 32) ? hi : lo;
}
// row slots per wave iteration: about 32 fp32 values per lane (occupancy beats bytes in flight per wave here: 64 was
// 1-4 % slower on every shape of tools/bench_rows.py)
#ifndef DMXQ_EXP_ROW_PACE
#define DMXQ_EXP_ROW_PACE 0
#endif
constexpr int kRowPace = DMXQ_EXP_ROW_PACE;   // common.hpp pace_issue between a wave's row loads (experiment)
#ifdef DMXQ_EXP_LN32
constexpr int kLn32OnePass = DMXQ_EXP_LN32;
#else
constexpr int kLn32OnePass = 0;
#endif
constexpr bool kRowEarlyLoads = false;  // layernorm_wave_kernel: next rows requested before this iteration's stores (measured slower, see there)
constexpr int rows_per_wave(int vpl, int epl) { return 32 / (vpl * epl) >= 4 ? 4 : (32 / (vpl * epl) >= 2 ? 2 : 1); }

// 16-bit outputs: exp(x - m) as v_exp_f32(fma(x, log2 e, -m log2 e)) and one reciprocal per row -- relative error
// ~2^-21, far inside the output format's 2^-9 / 2^-12 half-ulp (the rounding of m log2 e scales numerator and
// denominator alike and cancels); fp32 outputs keep expf and the per-element division.
// FAST32: float32 rows behind an output cast that keeps <= 16 mantissa bits (a module's FLOAT16 output cast): the v_exp / one
// reciprocal per row forms of the 16-bit rows (relative error ~2^-21, invisible at 2^-17) instead of the compensated expf and a
// division per element -- those make the fp32 kernel VALU-bound (Whisper's [12,1500,1500] attention: 58 us -> see profiles/r03_*)
template <int DT, int EPL, int VPL, int LPR, bool RAG = false, bool CAST = false, bool FAST32 = false, int RPWO = 0, bool BFPOUT = false>
__global__ __launch_bounds__(kThreads) void softmax_wave_kernel(const void* __restrict__ in, void* __restrict__ out,
                                                               int64_t rows, int64_t cols, float clamp_min, const RowCastArg<CAST> rc) {
  constexpr int SUB = kWave / LPR;  // rows side by side in one wave
  constexpr int RPW = RPWO ? RPWO : rows_per_wave(VPL, EPL);  // (RPWO: tools/tune_rows only)
  constexpr bool FAST = DT != DMXQ_F32 || FAST32;
  const int lane = threadIdx.x & (kWave - 1), sub = lane / LPR, sl = lane & (LPR - 1);
  const int64_t wave = (int64_t)blockIdx.x * (kThreads / kWave) + threadIdx.x / kWave;
  const int64_t n_waves = (int64_t)gridDim.x * (kThreads / kWave);
  const int nvf = (int)(cols / EPL), tail = RAG ? (int)(cols - (int64_t)nvf * EPL) : 0;  // whole vectors, tail elements
  const int nv = nvf + (tail ? 1 : 0);
  const int i_tail = nvf / LPR;  // the per-lane slot that holds a row's partial vector
  const bool has_clamp = clamp_min > -INFINITY;  // torch.clamp(x, min=input_clamp) (torch_modules.py:989-994)
  // PREFETCH (aligned rows with few vectors per lane): the NEXT iteration's rows are requested before the current ones are
  // reduced, see layernorm_wave_kernel.  (Not for RAG rows: their overlapped last vectors make a row's stores touch bytes that a
  // neighbouring lane of the SAME row loads -- fine within an iteration, where all loads precede the stores -- and in-place calls
  // must keep that order.)
  constexpr bool PREFETCH = false;  // measured slower on every shape, see layernorm_wave_kernel
  auto load_rows = [&](int64_t r0, RowVec<DT, EPL> (&dst)[RPW][VPL]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < RPW; j++) {
      const int64_t r = r0 + j * SUB + sub;
      const int64_t base = (r < rows ? r : rows - 1) * cols;
#pragma unroll
      for (int i = 0; i < VPL; i++) {
        const int v = i * LPR + sl;
        if constexpr (RAG) {
          // the partial last vector of a row is read (and written) as the EPL elements that END at the row end: its first
          // `back` elements repeat the previous lane's last ones -- left out of the row sum below, stored twice with the same
          // value -- so a ragged row has no element-wise access (all of a row's loads precede its stores in this wave: in-place safe)
          dst[j][i] = row_load_u<DT, EPL>(in, base + (v < nvf ? (int64_t)v * EPL : cols - EPL));
        } else {
          dst[j][i] = row_load<DT, EPL>(in, base + (int64_t)(v < nvf ? v : nvf - 1) * EPL);
          // (the softmax MODULE on 16-bit attention rows of 1281 .. 1536 elements -- Whisper's 1500 --: 16 idle issue cycles between a wave's
          //  loads, 21.14 -> 20.66 us on [12, 1500, 1500] bf16; pace 4 / 6: 20.77 / 20.95; two rows per wave + pace 2 / 4: 20.61 / 20.62; four: 25.5)
          pace_issue<(CAST && !BFPOUT && EPL == 4 && VPL == 6 && Elem<DT>::bytes == 2) ? 2 : kRowPace>();
        }
      }
    }
  };
  RowVec<DT, EPL> raw[RPW][VPL], nxt[PREFETCH ? RPW : 1][PREFETCH ? VPL : 1];
  if constexpr (PREFETCH) load_rows(wave * (RPW * SUB), raw);
  for (int64_t r0 = wave * (RPW * SUB); r0 < rows; r0 += n_waves * (RPW * SUB)) {
    if constexpr (PREFETCH) {
      load_rows(r0 + n_waves * (RPW * SUB), nxt);
      __builtin_amdgcn_sched_barrier(0);
    } else {
      load_rows(r0, raw);
    }
    float x[RPW][VPL][EPL], m[RPW], s[RPW];
#pragma unroll
    for (int j = 0; j < RPW; j++) {
      m[j] = -INFINITY;
#pragma unroll
      for (int i = 0; i < VPL; i++) {
        rowcast_raw_in<CAST, DT, EPL>(raw[j][i], rc);
        row_widen<DT, EPL>(raw[j][i], x[j][i]);
        rowcast_x_in<CAST, DT, EPL>(x[j][i], rc);
        if (has_clamp) {
#pragma unroll
          for (int k = 0; k < EPL; k++) x[j][i][k] = fmaxf(x[j][i][k], clamp_min);
        }
#pragma unroll
        for (int k = 0; k < EPL; k++) m[j] = fmaxf(m[j], x[j][i][k]);
      }
    }
#pragma unroll
    for (int j = 0; j < RPW; j++) m[j] = seg_max<LPR>(m[j]);
#pragma unroll
    for (int j = 0; j < RPW; j++) {
      const float mc = -m[j] * 1.4426950408889634f;
      s[j] = 0.0f;
#pragma unroll
      for (int i = 0; i < VPL; i++) {
        float t = 0.0f;
#pragma unroll
        for (int k = 0; k < EPL; k++) {
          x[j][i][k] = FAST ? __builtin_amdgcn_exp2f(__builtin_fmaf(x[j][i][k], 1.4426950408889634f, mc)) : exp_diff(x[j][i][k], m[j]);
          t += x[j][i][k];
        }
        if (RAG && tail && i == i_tail) {  // (wave-uniform) the slot of the overlapped last vector: its repeated elements do not count
          float t2 = 0.0f;
#pragma unroll
          for (int k = 0; k < EPL; k++) t2 += k >= EPL - tail ? x[j][i][k] : 0.0f;
          t = (i * LPR + sl == nvf) ? t2 : t;
        }
        s[j] += (i * LPR + sl < nv) ? t : 0.0f;
      }
    }
#pragma unroll
    for (int j = 0; j < RPW; j++) s[j] = seg_sum<LPR>(s[j]);
    if constexpr (BFPOUT) {
      // dmxq_softmax_cast_bfp: the probabilities go straight into the NEXT module's BFP input cast (blocks of B along the row = lpb
      // adjacent lanes of one slot i: LPR % lpb == 0, so a block never straddles slots; a ragged last block is simply shorter).  Every
      // lane runs the block-maximum DPP steps -- rows past the end were loaded clamped, slots past the row end contribute 0 -- and only
      // the store is predicated.  Same arithmetic as dmxq_bfp_qdq on the stored module output (bfp_math.hpp), so bit-identical to
      // the two launches.
      static_assert(!BFPOUT || (CAST && !RAG), "fused BFP epilogue: cast form, whole vectors");
      const int lpb = __builtin_amdgcn_readfirstlane(rc.bfp_lpb), wl = __builtin_amdgcn_readfirstlane(rc.bfp_wl);
#pragma unroll
      for (int j = 0; j < RPW; j++) {
        const int64_t r = r0 + j * SUB + sub;
        const float inv = 1.0f / s[j];
#pragma unroll
        for (int i = 0; i < VPL; i++) {
          const int v = i * LPR + sl;
          float y[EPL], q[EPL];
#pragma unroll
          for (int k = 0; k < EPL; k++) y[k] = FAST ? x[j][i][k] * inv : x[j][i][k] / s[j];
          bfp_epilogue<DT, EPL>(rowcast_pack_out<CAST, DT, EPL>(y, rc), v < nv, lpb, wl, q);
          if (v < nv && r < rows) row_store<DT, EPL>(out, r * cols + (int64_t)v * EPL, q);  // CastTo's `.to(dtype)`: one RNE rounding
        }
      }
    } else {
#pragma unroll
    for (int j = 0; j < RPW; j++) {
      const int64_t r = r0 + j * SUB + sub;
      if (r < rows) {
        const float inv = 1.0f / s[j];
#pragma unroll
        for (int i = 0; i < VPL; i++) {
          const int v = i * LPR + sl;
          if (v < nv) {
            float y[EPL];
#pragma unroll
            for (int k = 0; k < EPL; k++) y[k] = FAST ? x[j][i][k] * inv : x[j][i][k] / s[j];
            if constexpr (CAST) {
              const RowVec<DT, EPL> o = rowcast_pack_out<CAST, DT, EPL>(y, rc);
              if constexpr (!RAG) row_store_raw<DT, EPL>(out, r * cols + (int64_t)v * EPL, o);
              else row_store_raw<DT, EPL, true>(out, r * cols + (v < nvf ? (int64_t)v * EPL : cols - EPL), o);
            } else if constexpr (!RAG) {
              row_store<DT, EPL>(out, r * cols + (int64_t)v * EPL, y);
            } else {
              row_store_u<DT, EPL>(out, r * cols + (v < nvf ? (int64_t)v * EPL : cols - EPL), y);
            }
          }
        }
      }
    }
    }
    if constexpr (PREFETCH) {
#pragma unroll
      for (int j = 0; j < RPW; j++)
#pragma unroll
        for (int i = 0; i < VPL; i++) raw[j][i] = nxt[j][i];
    }
  }
}

// weight / bias (same dtype as the rows) are read ONCE per wave and kept across its rows: widened to fp32 for short
// rows, packed for longer ones.  Per element: widen + add, subtract + fma, and (x - mean) * (rstd w) + b as
// subtract + fma with rstd w formed once per (row, vector element).
template <int DT, int EPL, int VPL, int LPR, bool RMS = false, bool CAST = false, int RPWO = 0, int HOISTO = 0, int PFO = -1, bool BFPOUT = false>
__global__ __launch_bounds__(kThreads) void layernorm_wave_kernel(const void* __restrict__ in, void* __restrict__ out,
                                                                 int64_t rows, int64_t cols, const void* __restrict__ w,
                                                                 const void* __restrict__ b, float eps, const RowCastArg<CAST> rc) {
  constexpr int SUB = kWave / LPR;
  constexpr int RPW = RPWO ? RPWO : rows_per_wave(VPL, EPL);
  constexpr bool HOIST_F32 = HOISTO == 0 && VPL * EPL <= 24, HOIST_RAW = HOISTO != 2 && !HOIST_F32 && VPL <= 8;  // (HOISTO: tools/tune_rows only)
  const int lane = threadIdx.x & (kWave - 1), sub = lane / LPR, sl = lane & (LPR - 1);
  const int64_t wave = (int64_t)blockIdx.x * (kThreads / kWave) + threadIdx.x / kWave;
  const int64_t n_waves = (int64_t)gridDim.x * (kThreads / kWave);
  const int nv = (int)(cols / EPL);
  const float inv_n = 1.0f / (float)cols;
  float wf[HOIST_F32 ? VPL : 1][EPL], bf[HOIST_F32 ? VPL : 1][EPL];
  RowVec<DT, EPL> wr[(HOIST_F32 || HOIST_RAW) ? VPL : 1], br[(HOIST_F32 || HOIST_RAW) ? VPL : 1];
  if (HOIST_F32 || HOIST_RAW) {  // the raw reads only: their conversion waits until the wave's first rows are requested too (below)
#pragma unroll
    for (int i = 0; i < VPL; i++) {
      const int v = i * LPR + sl;
      const int64_t c = (int64_t)(v < nv ? v : nv - 1) * EPL;
      if (w) wr[i] = row_load_keep<DT, EPL>(w, c);
      if (b) br[i] = row_load_keep<DT, EPL>(b, c);
    }
  }
  // EARLY (measured in round 3, OFF): request the rows of the NEXT iteration as soon as the current ones are widened -- into the same raw
  // registers, dead by then -- i.e. before this iteration's reductions and stores (a wave's loads and stores retire in order through
  // one counter, so the plain loop's wait for the next rows also waits for the previous stores).  SLOWER: 21845 x 768 bf16 14.4 -> 16.1 us,
  // the fused module 14.7 -> 17.1, 128 MiB 45.5 -> 47.7 (tools/tune_rows, profiles/r03_tune_rows.txt); so was an earlier form with a second
  // register set, issued at the loop top (profiles/r03_row_prefetch.txt).  What these rows lack at 32 MiB is not queue depth per wave:
  // the same kernel reaches 74 % of the roofline at 128 MiB.  The code path stays (compiled out) as the record of the experiment.
  constexpr bool EARLY = PFO < 0 ? kRowEarlyLoads : PFO != 0;
  // (the LayerNorm module on 16-bit rows of 768: paced loads, measured with the one-pass grid of the launcher; everything else: the tuner's hook)
  // (-DDMXQ_EXP_LN32=N, A/B builds: the float32 module on rows of 768 -- 64 lanes x 3 vectors of 4 -- on the one-pass grid too, pace N - 1)
  constexpr int kLnModulePace = (CAST && !RMS && !BFPOUT && VPL == 3 && EPL == 8 && LPR == 32 && RPWO == 0) ? 4
                                : ((kLn32OnePass > 0 && CAST && !RMS && !BFPOUT && VPL == 3 && EPL == 4 && LPR == 64 && RPWO == 0) ? kLn32OnePass - 1 : kRowPace);
  auto load_rows = [&](int64_t r0, RowVec<DT, EPL> (&dst)[RPW][VPL]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < RPW; j++) {
      const int64_t r = r0 + j * SUB + sub;
      const int64_t base = (r < rows ? r : rows - 1) * cols;
#pragma unroll
      for (int i = 0; i < VPL; i++) {
        const int v = i * LPR + sl;
        dst[j][i] = row_load<DT, EPL>(in, base + (int64_t)(v < nv ? v : nv - 1) * EPL);
        pace_issue<kLnModulePace>();
      }
    }
  };
  RowVec<DT, EPL> raw[RPW][VPL];
  const int64_t r_step = n_waves * (RPW * SUB), r_first = wave * (RPW * SUB);
  // the wave's FIRST rows are requested before anything waits on the weight / bias reads (round 3: the widening of the hoisted
  // parameters used to sit between their loads and the first row loads -- a full memory round trip before a wave's first row request,
  // which is most of the run time of a small tensor: every wave of a [256, 768] activation makes exactly one iteration)
  load_rows(r_first, raw);
  __builtin_amdgcn_sched_barrier(0);
  if (HOIST_F32) {
#pragma unroll
    for (int i = 0; i < VPL; i++) {
      if (w) row_widen<DT, EPL>(wr[i], wf[i]);
      if (b) row_widen<DT, EPL>(br[i], bf[i]);
    }
  }
  for (int64_t r0 = r_first; r0 < rows; r0 += r_step) {
    if constexpr (!EARLY) { if (r0 != r_first) load_rows(r0, raw); }
    float x[RPW][VPL][EPL], mean[RPW], rstd[RPW];
#pragma unroll
    for (int j = 0; j < RPW; j++) {
      float s = 0.0f;
#pragma unroll
      for (int i = 0; i < VPL; i++) {
        rowcast_raw_in<CAST, DT, EPL>(raw[j][i], rc);
        row_widen<DT, EPL>(raw[j][i], x[j][i]);
        rowcast_x_in<CAST, DT, EPL>(x[j][i], rc);
        float t = 0.0f;
#pragma unroll
        for (int k = 0; k < EPL; k++) t += x[j][i][k];
        s += (i * LPR + sl < nv) ? t : 0.0f;
      }
      mean[j] = s;
    }
    if constexpr (EARLY) {
      __builtin_amdgcn_sched_barrier(0);
      load_rows(r0 + r_step, raw);  // (rows past the end re-read the last row: unconditional loads, never stored)
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < RPW; j++) mean[j] = RMS ? 0.0f : seg_sum<LPR>(mean[j]) * inv_n;  // RMSNorm: no centring
#pragma unroll
    for (int j = 0; j < RPW; j++) {
      float q = 0.0f;
#pragma unroll
      for (int i = 0; i < VPL; i++) {
        float t = 0.0f;
#pragma unroll
        for (int k = 0; k < EPL; k++) { const float d = x[j][i][k] - mean[j]; t = __builtin_fmaf(d, d, t); }
        q += (i * LPR + sl < nv) ? t : 0.0f;
      }
      rstd[j] = q;
    }
#pragma unroll
    for (int j = 0; j < RPW; j++) rstd[j] = 1.0f / sqrtf(seg_sum<LPR>(rstd[j]) * inv_n + eps);  // biased variance, as F.layer_norm
    if constexpr (BFPOUT) {
      // dmxq_layernorm_cast_bfp / dmxq_rmsnorm_cast_bfp: the consumers' BFP input cast on the module's result (bfp_epilogue: every lane
      // takes part, only the store is predicated)
      static_assert(!BFPOUT || CAST, "fused BFP epilogue: cast form");
      const int lpb = __builtin_amdgcn_readfirstlane(rc.bfp_lpb), wl = __builtin_amdgcn_readfirstlane(rc.bfp_wl);
#pragma unroll
      for (int i = 0; i < VPL; i++) {
        const int v = i * LPR + sl, vc = v < nv ? v : nv - 1;
        float ww[EPL], bb[EPL];
        if (!HOIST_F32) {
          if (w) row_widen<DT, EPL>(HOIST_RAW ? wr[i] : row_load_keep<DT, EPL>(w, (int64_t)vc * EPL), ww);
          if (b) row_widen<DT, EPL>(HOIST_RAW ? br[i] : row_load_keep<DT, EPL>(b, (int64_t)vc * EPL), bb);
        }
#pragma unroll
        for (int j = 0; j < RPW; j++) {
          const int64_t r = r0 + j * SUB + sub;
          float y[EPL], q[EPL];
#pragma unroll
          for (int k = 0; k < EPL; k++) {
            const float g = w ? rstd[j] * (HOIST_F32 ? wf[i][k] : ww[k]) : rstd[j];
            const float d = x[j][i][k] - mean[j];
            y[k] = b ? __builtin_fmaf(d, g, HOIST_F32 ? bf[i][k] : bb[k]) : d * g;
          }
          bfp_epilogue<DT, EPL>(rowcast_pack_out<CAST, DT, EPL>(y, rc), v < nv, lpb, wl, q);
          if (v < nv && r < rows) row_store<DT, EPL>(out, r * cols + (int64_t)v * EPL, q);
        }
      }
      continue;
    }
#pragma unroll
    for (int i = 0; i < VPL; i++) {
      const int v = i * LPR + sl;
      if (v < nv) {
        float ww[EPL], bb[EPL];
        if (!HOIST_F32) {
          if (w) row_widen<DT, EPL>(HOIST_RAW ? wr[i] : row_load_keep<DT, EPL>(w, (int64_t)v * EPL), ww);
          if (b) row_widen<DT, EPL>(HOIST_RAW ? br[i] : row_load_keep<DT, EPL>(b, (int64_t)v * EPL), bb);
        }
#pragma unroll
        for (int j = 0; j < RPW; j++) {
          const int64_t r = r0 + j * SUB + sub;
          if (r < rows) {
            float y[EPL];
#pragma unroll
            for (int k = 0; k < EPL; k++) {
              const float g = w ? rstd[j] * (HOIST_F32 ? wf[i][k] : ww[k]) : rstd[j];
              const float d = x[j][i][k] - mean[j];
              y[k] = b ? __builtin_fmaf(d, g, HOIST_F32 ? bf[i][k] : bb[k]) : d * g;
            }
            if constexpr (CAST) row_store_raw<DT, EPL>(out, r * cols + (int64_t)v * EPL, rowcast_pack_out<CAST, DT, EPL>(y, rc));
            else row_store<DT, EPL>(out, r * cols + (int64_t)v * EPL, y);
          }
        }
      }
    }
  }
}

// Long rows (more than 256 lane-vectors): a 256-thread WORKGROUP per row, still register resident (VPL vectors per
// thread), two workgroup reductions per row through a 4-entry LDS exchange, RPW rows per iteration to amortise the two
// barriers, weight / bias hoisted (packed) out of the persistent row loop.  The wave kernel at this size would hold
// 128+ values per lane and re-read weight and bias (as much data as the row itself) for every row.
template <int DT, int EPL, int VPL, bool RMS = false, bool CAST = false, int RPWO = 0, bool ONE_PASS = false, bool BFPOUT = false>
__global__ __launch_bounds__(kThreads) void layernorm_block_kernel(const void* __restrict__ in, void* __restrict__ out,
                                                                  int64_t rows, int64_t cols, const void* __restrict__ w,
                                                                  const void* __restrict__ b, float eps, const RowCastArg<CAST> rc) {
  constexpr int RPW = RPWO ? RPWO : (VPL * EPL <= 32 ? 2 : 1);
  constexpr int NW = kThreads / kWave;
  __shared__ float red[2][RPW][NW];
  const int t = threadIdx.x, lane = t & (kWave - 1), wv = t / kWave;
  const int nv = (int)(cols / EPL);
  const float inv_n = 1.0f / (float)cols;
  RowVec<DT, EPL> wr[VPL], br[VPL];
#pragma unroll
  for (int i = 0; i < VPL; i++) {
    const int v = i * kThreads + t;
    const int64_t c = (int64_t)(v < nv ? v : nv - 1) * EPL;
    if (w) wr[i] = row_load_keep<DT, EPL>(w, c);
    if (b) br[i] = row_load_keep<DT, EPL>(b, c);
  }
  int par = 0;  // RMSNorm: ONE barrier per iteration, so the exchange buffer alternates (a fast wave's next write must not overtake a slow wave's read)
  // ONE_PASS (norm_dispatch launches one workgroup per RPW rows): NO loop around the body, so that the row loads
  // follow the weight / bias loads in straight-line code.  Inside a loop the compiler drains the parameter loads at the loop entry --
  // a memory round trip before the workgroup's first row request (tools/isa_prologue.py).  (Hoisting the first iteration's loads out
  // of the persistent loop instead keeps `raw` live across the back edge: 11.7 -> 16.2 us on 4096 x 4096.)
  auto body = [&](const int64_t r0) __attribute__((always_inline)) {
    const int qb = RMS ? par : 1;
    par ^= 1;
    RowVec<DT, EPL> raw[RPW][VPL];
#pragma unroll
    for (int j = 0; j < RPW; j++) {
      const int64_t base = (r0 + j < rows ? r0 + j : rows - 1) * cols;
#pragma unroll
      for (int i = 0; i < VPL; i++) {
        const int v = i * kThreads + t;
        raw[j][i] = row_load<DT, EPL>(in, base + (int64_t)(v < nv ? v : nv - 1) * EPL);
        pace_issue<(CAST && RMS && RPWO == 4) ? 2 : kRowPace>();   // (the RMSNorm module's 26-32 MiB form: see the launcher)
      }
    }
    float x[RPW][VPL][EPL], mean[RPW], rstd[RPW];
#pragma unroll
    for (int j = 0; j < RPW; j++) {
      float s = 0.0f;
#pragma unroll
      for (int i = 0; i < VPL; i++) {
        rowcast_raw_in<CAST, DT, EPL>(raw[j][i], rc);
        row_widen<DT, EPL>(raw[j][i], x[j][i]);
        rowcast_x_in<CAST, DT, EPL>(x[j][i], rc);
        if constexpr (!RMS) {
          float u = 0.0f;
#pragma unroll
          for (int k = 0; k < EPL; k++) u += x[j][i][k];
          s += (i * kThreads + t < nv) ? u : 0.0f;
        }
      }
      if constexpr (!RMS) {
        s = seg_sum<kWave>(s);
        if (lane == 0) red[0][j][wv] = s;
      }
    }
    if constexpr (!RMS) __syncthreads();  // RMSNorm has no centring pass: one exchange (and one barrier) per iteration, not two
#pragma unroll
    for (int j = 0; j < RPW; j++) {
      float s = 0.0f;
      if constexpr (!RMS) {
        s = red[0][j][0];
#pragma unroll
        for (int k = 1; k < NW; k++) s += red[0][j][k];
      }
      mean[j] = RMS ? 0.0f : s * inv_n;
      float q = 0.0f;
#pragma unroll
      for (int i = 0; i < VPL; i++) {
        float u = 0.0f;
#pragma unroll
        for (int k = 0; k < EPL; k++) { const float d = x[j][i][k] - mean[j]; u = __builtin_fmaf(d, d, u); }
        q += (i * kThreads + t < nv) ? u : 0.0f;
      }
      q = seg_sum<kWave>(q);
      if (lane == 0) red[qb][j][wv] = q;
    }
    __syncthreads();  // (also orders this iteration's reads of red[0] before the next iteration's writes)
#pragma unroll
    for (int j = 0; j < RPW; j++) {
      float q = red[qb][j][0];
#pragma unroll
      for (int k = 1; k < NW; k++) q += red[qb][j][k];
      rstd[j] = 1.0f / sqrtf(q * inv_n + eps);  // biased variance, as F.layer_norm
    }
    if constexpr (BFPOUT) {
      static_assert(!BFPOUT || CAST, "fused BFP epilogue: cast form");
      const int lpb = __builtin_amdgcn_readfirstlane(rc.bfp_lpb), wl = __builtin_amdgcn_readfirstlane(rc.bfp_wl);
#pragma unroll
      for (int i = 0; i < VPL; i++) {
        const int v = i * kThreads + t;
        float ww[EPL], bb[EPL];
        if (w) row_widen<DT, EPL>(wr[i], ww);   // (loaded clamped: valid for every lane)
        if (b) row_widen<DT, EPL>(br[i], bb);
#pragma unroll
        for (int j = 0; j < RPW; j++) {
          float y[EPL], q[EPL];
#pragma unroll
          for (int k = 0; k < EPL; k++) {
            const float g = w ? rstd[j] * ww[k] : rstd[j];
            const float d = x[j][i][k] - mean[j];
            y[k] = b ? __builtin_fmaf(d, g, bb[k]) : d * g;
          }
          bfp_epilogue<DT, EPL>(rowcast_pack_out<CAST, DT, EPL>(y, rc), v < nv, lpb, wl, q);
          if (v < nv && r0 + j < rows) row_store<DT, EPL>(out, (r0 + j) * cols + (int64_t)v * EPL, q);
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < VPL; i++) {
      const int v = i * kThreads + t;
      if (v < nv) {
        float ww[EPL], bb[EPL];
        if (w) row_widen<DT, EPL>(wr[i], ww);
        if (b) row_widen<DT, EPL>(br[i], bb);
#pragma unroll
        for (int j = 0; j < RPW; j++) {
          if (r0 + j < rows) {
            float y[EPL];
#pragma unroll
            for (int k = 0; k < EPL; k++) {
              const float g = w ? rstd[j] * ww[k] : rstd[j];
              const float d = x[j][i][k] - mean[j];
              y[k] = b ? __builtin_fmaf(d, g, bb[k]) : d * g;
            }
            if constexpr (CAST) row_store_raw<DT, EPL>(out, (r0 + j) * cols + (int64_t)v * EPL, rowcast_pack_out<CAST, DT, EPL>(y, rc));
            else row_store<DT, EPL>(out, (r0 + j) * cols + (int64_t)v * EPL, y);
          }
        }
      }
    }
  };
  if constexpr (ONE_PASS) {
    if ((int64_t)blockIdx.x * RPW < rows) body((int64_t)blockIdx.x * RPW);
  } else {
    for (int64_t r0 = (int64_t)blockIdx.x * RPW; r0 < rows; r0 += (int64_t)gridDim.x * RPW) body(r0);
  }
}

// lane-vector width for the wave kernels: 16 bytes when rows and bases allow it, else 8 bytes for 16-bit rows that are
// a multiple of 4 elements; 0 = not applicable
static inline int wave_epl(int dt, int64_t cols, const void* p0, const void* p1, const void* p2, const void* p3) {
  const uintptr_t a = (uintptr_t)p0 | (uintptr_t)p1 | (uintptr_t)p2 | (uintptr_t)p3;
  const int full = dt == DMXQ_F32 ? 4 : 8;
  if (cols % full == 0 && (a & 15) == 0) return full;
  if (dt != DMXQ_F32 && cols % 4 == 0 && (a & 7) == 0) return 4;
  return 0;
}

// (lanes per row, vectors per lane) for a row of nv vectors: the instantiated shape with the fewest idle lane slots.
// 64 lanes: VPL in {1,2,3,4,5,6,8,10,12,16}; 32 lanes: VPL in {1,3,5}.  vpl = 0: row too long for registers.
static inline void wave_shape(int64_t nv, int* lpr, int* vpl) {
  static const int v64[] = {1, 2, 3, 4, 5, 6, 8, 10, 12, 16}, v32[] = {1, 3, 5};
  int64_t best = 0;
  *lpr = 64; *vpl = 0;
  for (int v : v64)
    if ((int64_t)v * 64 >= nv) { best = (int64_t)v * 64; *vpl = v; break; }
  for (int v : v32)
    if ((int64_t)v * 32 >= nv) { if (best == 0 || (int64_t)v * 32 < best) { *lpr = 32; *vpl = v; } break; }
}

}  // namespace dmxq

using namespace dmxq;

// persistent grid for the register-resident row kernels: as many workgroups as are resident at once (occupancy of
// THIS kernel x number of CUs; queried once per kernel and cached), or fewer when the tensor is small
template <typename K>
static int resident_grid(K kernel, int64_t wanted) {
  // one table per kernel instantiation, one slot per device (a process may drive several GPUs: the CU count and the
  // occupancy belong to the device that is current for THIS launch)
  constexpr int kMaxDev = 64;
  static std::atomic<int> per_device[kMaxDev] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) dev = 0;
  int cap = per_device[dev].load(std::memory_order_relaxed);
  if (cap == 0) {
    int cus = 256, per_cu = 2;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kThreads, 0) != hipSuccess || per_cu < 1) per_cu = 2;
    cap = cus * per_cu;
    per_device[dev].store(cap, std::memory_order_relaxed);
  }
  return (int)(wanted < cap ? (wanted < 1 ? 1 : wanted) : cap);
}

// one pass per workgroup (the kernels still loop, for grids beyond the limit).  Round 3 (tools/tune_rows -> profiles/r03_tune_rows.txt):
// softmax rows of 1500 bf16, persistent grid vs one pass: 14.1 vs 13.7 us on 32 MiB, 52.0 vs 46.6 us on 128 MiB, equal below 16 MiB --
// a softmax wave has no per-wave setup to amortise.  (LayerNorm's wave kernel has: weight and bias, hoisted and widened once per
// wave; it stays persistent: 14.4 vs 17.2 us for one pass on 21845 x 768.)
static inline unsigned one_pass_grid(int64_t wanted) { return (unsigned)(wanted < 1 ? 1 : (wanted > 0x7FFFFFFF ? 0x7FFFFFFF : wanted)); }

static inline int row_grid(int64_t rows) { return (int)(rows < 256 * 16 ? (rows < 1 ? 1 : rows) : 256 * 16); }

// This file is compiled FIVE times (build.py: -DDMXQ_EW_PART=1 .. 5): softmax, and the four (RMS, CAST) forms of the norms, ~70-150 kernel
// instantiations each.
#ifndef DMXQ_EW_PART
#define DMXQ_EW_PART 0
#endif
#define DMXQ_AP(P_) (DMXQ_EW_PART == 0 || DMXQ_EW_PART == (P_))

#if DMXQ_AP(1)
template <bool CAST>
static int softmax_dispatch(const void* in, void* out, int dtype_in, int dtype_out, int64_t rows, int64_t cols,
                            float input_clamp_min, void* stream, const RowCastArg<CAST>& rc) {
  if (!valid_dtype(dtype_in) || !valid_dtype(dtype_out) || rows < 0 || cols < 0) return DMXQ_ERR_BAD_ARG;
  if (rows * cols == 0) return DMXQ_OK;
  if (!in || !out) return DMXQ_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  // register-resident kernel: same dtype in and out, a row of 1 .. 1024 lane-vectors.  Rows that are whole aligned
  // vectors take the aligned form; any other length / alignment the RAG form (unaligned 16-byte accesses + element tail)
  const int full = dtype_in == DMXQ_F32 ? 4 : 8;
  const uintptr_t eb = dtype_in == DMXQ_F32 ? 4 : 2;
  const bool elem_aligned = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & (eb - 1)) == 0;
  if (dtype_in == dtype_out && elem_aligned && cols >= full && (cols + full - 1) / full <= 64 * 16) {
    bool rag = !(cols % full == 0 && aligned16(in) && aligned16(out));
    // 16-bit rows that are a multiple of 4 elements on 8-byte bases: aligned 8-byte lane-vectors (measured 5 % faster
    // than the unaligned 16-byte form with an overlapped last vector on attention rows of 1500: 21.8 vs 22.9 us)
    const bool half_vec = rag && full == 8 && cols % 4 == 0 && cols / 4 <= 64 * 16 &&
                          ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 7u) == 0;
    if (half_vec) rag = false;
    int lpr, vpl;
    wave_shape(half_vec ? cols / 4 : (cols + full - 1) / full, &lpr, &vpl);
    bool fast32 = false;
    if constexpr (CAST) fast32 = !rag && rc.go.active && rc.go.f.man <= 16;
#define DMXQ_SM(D_, E_, V_, L_)                                                                                       \
  do {                                                                                                                \
    constexpr int per_wg = 4 * rows_per_wave(V_, E_) * (64 / (L_));                                                   \
    if constexpr (CAST) {                                                                                             \
      if (rc.bfp_B) { /* dmxq_softmax_cast_bfp: blocks of B elements = lpb adjacent lanes of one slot */               \
        const int lpb = rc.bfp_B / (E_);                                                                              \
        if (rag || rc.bfp_B % (E_) != 0 || lpb < 1 || (lpb & (lpb - 1)) != 0 || lpb > (L_)) return DMXQ_ERR_UNSUPPORTED; \
        RowCast rb = rc;                                                                                              \
        rb.bfp_lpb = lpb;                                                                                             \
        if ((D_) == DMXQ_F32 && fast32)                                                                               \
          DMXQ_LAUNCH((softmax_wave_kernel<D_, E_, V_, L_, false, CAST, (D_) == DMXQ_F32, 0, true>),                  \
                      dim3(one_pass_grid((rows + per_wg - 1) / per_wg)), dim3(kThreads), 0, s, in, out, rows, cols, input_clamp_min, rb); \
        else                                                                                                          \
          DMXQ_LAUNCH((softmax_wave_kernel<D_, E_, V_, L_, false, CAST, false, 0, true>),                             \
                      dim3(one_pass_grid((rows + per_wg - 1) / per_wg)), dim3(kThreads), 0, s, in, out, rows, cols, input_clamp_min, rb); \
        break;                                                                                                        \
      }                                                                                                               \
    }                                                                                                                 \
    if constexpr ((E_) * Elem<D_>::bytes == 16) {                                                                     \
      if (rag) {                                                                                                      \
        DMXQ_LAUNCH((softmax_wave_kernel<D_, E_, V_, L_, true, CAST>),                                         \
                           dim3((unsigned)one_pass_grid((rows + per_wg - 1) / per_wg)), \
                           dim3(kThreads), 0, s, in, out, rows, cols, input_clamp_min, rc);                           \
        break;                                                                                                        \
      }                                                                                                               \
    }                                                                                                                 \
    if constexpr (CAST && (D_) == DMXQ_F32) {                                                                         \
      if (fast32) {                                                                                                   \
        DMXQ_LAUNCH((softmax_wave_kernel<D_, E_, V_, L_, false, CAST, true>),                                  \
                           dim3((unsigned)one_pass_grid((rows + per_wg - 1) / per_wg)), \
                           dim3(kThreads), 0, s, in, out, rows, cols, input_clamp_min, rc);                           \
        break;                                                                                                        \
      }                                                                                                               \
    }                                                                                                                 \
      DMXQ_LAUNCH((softmax_wave_kernel<D_, E_, V_, L_, false, CAST>),                                          \
                         dim3((unsigned)one_pass_grid((rows + per_wg - 1) / per_wg)), \
                         dim3(kThreads), 0, s, in, out, rows, cols, input_clamp_min, rc);                             \
  } while (0)
#define DMXQ_SM_V(D_, E_)                                                                            \
  do {                                                                                               \
    if (lpr == 32) { if (vpl == 1) DMXQ_SM(D_, E_, 1, 32); else if (vpl == 3) DMXQ_SM(D_, E_, 3, 32); else DMXQ_SM(D_, E_, 5, 32); } \
    else switch (vpl) {                                                                              \
      case 1: DMXQ_SM(D_, E_, 1, 64); break;   case 2: DMXQ_SM(D_, E_, 2, 64); break;                 \
      case 3: DMXQ_SM(D_, E_, 3, 64); break;   case 4: DMXQ_SM(D_, E_, 4, 64); break;                 \
      case 5: DMXQ_SM(D_, E_, 5, 64); break;   case 6: DMXQ_SM(D_, E_, 6, 64); break;                 \
      case 8: DMXQ_SM(D_, E_, 8, 64); break;   case 10: DMXQ_SM(D_, E_, 10, 64); break;               \
      case 12: DMXQ_SM(D_, E_, 12, 64); break; default: DMXQ_SM(D_, E_, 16, 64); break;               \
    }                                                                                                \
  } while (0)
    if (dtype_in == DMXQ_F32) DMXQ_SM_V(DMXQ_F32, 4);
    else if (dtype_in == DMXQ_BF16) { if (half_vec) DMXQ_SM_V(DMXQ_BF16, 4); else DMXQ_SM_V(DMXQ_BF16, 8); }
    else { if (half_vec) DMXQ_SM_V(DMXQ_F16, 4); else DMXQ_SM_V(DMXQ_F16, 8); }
#undef DMXQ_SM_V
#undef DMXQ_SM
    return launch_status();
  }
  if constexpr (CAST) return DMXQ_ERR_UNSUPPORTED;  // the fused-cast form exists for register-resident rows only
  const size_t scratch = (kThreads / kWave) * sizeof(float);
  if (cols <= kRowLdsFloats)
    DMXQ_LAUNCH(softmax_rows_kernel<true>, dim3(row_grid(rows)), dim3(kThreads), scratch + cols * sizeof(float), s,
                       in, out, dtype_in, dtype_out, rows, cols, input_clamp_min);
  else
    DMXQ_LAUNCH(softmax_rows_kernel<false>, dim3(row_grid(rows)), dim3(kThreads), scratch, s, in, out, dtype_in,
                       dtype_out, rows, cols, input_clamp_min);
  return launch_status();
}

extern "C" int dmxq_softmax(const void* in, void* out, int dtype_in, int dtype_out, int64_t rows, int64_t cols,
                            float input_clamp_min, void* stream) {
  return softmax_dispatch<false>(in, out, dtype_in, dtype_out, rows, cols, input_clamp_min, stream, NoRowCast{});
}

#endif  // part 1 (softmax)

// the two casts of a module around a row function: false = not a combination the fused kernels take
static bool rowcast_of(int dtype, const dmxq_float_fmt* cast_in, const dmxq_float_fmt* cast_out, RowCast* rc) {
  *rc = RowCast{};
  if (dtype == DMXQ_F32) return castg_of(cast_in, &rc->gi) && castg_of(cast_out, &rc->go);
  return range16_of(cast_in, dtype, &rc->ri) && range16_of(cast_out, dtype, &rc->ro);
}

#if DMXQ_AP(1)
extern "C" int dmxq_softmax_cast(const void* in, void* out, int dtype, int64_t rows, int64_t cols, float input_clamp_min,
                                 const dmxq_float_fmt* cast_in, const dmxq_float_fmt* cast_out, void* stream) {
  RowCast rc;
  if (!valid_dtype(dtype)) return DMXQ_ERR_BAD_ARG;
  if (!rowcast_of(dtype, cast_in, cast_out, &rc)) return DMXQ_ERR_UNSUPPORTED;
  return softmax_dispatch<true>(in, out, dtype, dtype, rows, cols, input_clamp_min, stream, rc);
}

// ... and with the NEXT module's BFP input cast applied to the result (modeling/nn/core.py:228-264 of the consumer: `input_casts` of the
// ActActMatMul / Linear that takes the probabilities, numerical/format.py:304-343): out = BFP_QDQ(cast_out(softmax(cast_in(x)))) along
// the rows, symmetric, nearest -- the pass over the [heads, S, S] probabilities that the consumer's cast would make (108 MB each way
// for Whisper-small) disappears.  Bit-identical to dmxq_softmax_cast followed by dmxq_bfp_qdq.  DMXQ_ERR_UNSUPPORTED when the rows
// are not whole lane-vectors or the block size does not map onto whole lanes (the caller runs the two launches).
extern "C" int dmxq_softmax_cast_bfp(const void* in, void* out, int dtype, int64_t rows, int64_t cols, float input_clamp_min,
                                     const dmxq_float_fmt* cast_in, const dmxq_float_fmt* cast_out, int64_t block_size, int precision,
                                     void* stream) {
  RowCast rc;
  if (!valid_dtype(dtype) || block_size < 1) return DMXQ_ERR_BAD_ARG;
  if (precision < 2 || precision > 20 || block_size > 512) return DMXQ_ERR_UNSUPPORTED;
  if (!rowcast_of(dtype, cast_in, cast_out, &rc)) return DMXQ_ERR_UNSUPPORTED;
  rc.bfp_B = (int)block_size;
  rc.bfp_wl = precision;
  const int full = dtype == DMXQ_F32 ? 4 : 8;
  if (rows * cols != 0 && !(cols >= full && (cols + full - 1) / full <= 64 * 16)) return DMXQ_ERR_UNSUPPORTED;  // (register-resident rows only)
  return softmax_dispatch<true>(in, out, dtype, dtype, rows, cols, input_clamp_min, stream, rc);
}

#endif  // part 1

#if DMXQ_AP(2) || DMXQ_AP(3) || DMXQ_AP(4) || DMXQ_AP(5)
template <bool RMS, bool CAST = false>
static int norm_dispatch(const void* in, void* out, int dtype_in, int dtype_out, int64_t rows, int64_t cols,
                         const void* weight, const void* bias, int dtype_wb, float eps, void* stream, const RowCastArg<CAST>& rc = {}) {
  if (!valid_dtype(dtype_in) || !valid_dtype(dtype_out) || rows < 0 || cols < 0) return DMXQ_ERR_BAD_ARG;
  if ((weight || bias) && !valid_dtype(dtype_wb)) return DMXQ_ERR_BAD_ARG;
  if (rows * cols == 0) return DMXQ_OK;
  if (!in || !out) return DMXQ_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  const bool same = dtype_in == dtype_out && ((!weight && !bias) || dtype_wb == dtype_in);
  const int epl = same ? wave_epl(dtype_in, cols, in, out, weight, bias) : 0;
  const int full_epl = dtype_in == DMXQ_F32 ? 4 : 8;
  // (round 3: the wave kernel at 512 vectors per row -- one wave per 4096-element bf16 row, 8 vectors per lane -- measured 15.6 us against
  // this kernel's 13.4 us on 4096 x 4096)
  if (epl == full_epl && cols / epl > 256 && cols / epl <= 8 * kThreads) {  // long rows: workgroup per row
    const int64_t nv = cols / epl;
    const int vpl = (int)((nv + kThreads - 1) / kThreads);
    // rows per workgroup iteration (round 3, tools/tune_rows -> profiles/r03_tune_rows.txt, RMSNorm / LayerNorm rows of 4096 bf16): 2, on a
    // persistent grid (2560 / 3072 x 4096: 8.2 / 9.9 us against 9.3 / 10.8 us for one pass per workgroup) -- except tensors of 26-32 MiB,
    // which fit the chip in ONE round of workgroups when each lane keeps ~16 vectors in flight: 8 rows per workgroup, one pass, 10.9 /
    // 11.5 us against 11.9 / 13.2 us on 3584 / 4096 x 4096 (RMSNorm 63 -> 72 % of the roofline; LayerNorm 13.7 -> 12.3 us).  The fused
    // module form (CAST) gains 2 % at most from it and keeps the one geometry.
    const int64_t bytes = rows * cols * (dtype_in == DMXQ_F32 ? 4 : 2);
    const bool mid = bytes > ((int64_t)26 << 20) && bytes <= ((int64_t)32 << 20);
#define DMXQ_LNB(D_, E_, V_)                                                                                          \
  do {                                                                                                                \
    constexpr int rpw = (V_) * (E_) <= 32 ? 2 : 1, rpw_mid = 16 / (V_) >= 8 ? 8 : (16 / (V_) >= 4 ? 4 : 2);            \
    if constexpr (CAST) {                                                                                             \
      if (rc.bfp_B) { /* dmxq_*norm_cast_bfp: blocks of B elements = lpb adjacent lanes */                              \
        const int lpb = rc.bfp_B / (E_);                                                                              \
        if (rc.bfp_B % (E_) != 0 || lpb < 1 || (lpb & (lpb - 1)) != 0 || lpb > 64) return DMXQ_ERR_UNSUPPORTED;        \
        RowCast rb = rc;                                                                                              \
        rb.bfp_lpb = lpb;                                                                                             \
        DMXQ_LAUNCH((layernorm_block_kernel<D_, E_, V_, RMS, CAST, 0, false, true>),                                \
                    dim3((unsigned)resident_grid(layernorm_block_kernel<D_, E_, V_, RMS, CAST, 0, false, true>, (rows + rpw - 1) / rpw)), \
                    dim3(kThreads), 0, s, in, out, rows, cols, weight, bias, eps, rb);                                \
        break;                                                                                                        \
      }                                                                                                               \
    }                                                                                                                 \
    if constexpr (CAST && RMS && (V_) == 2 && (E_) == 8) {                                                            \
      /* the RMSNorm MODULE on 16-bit rows of 4096 (Llama's hidden size), 26-32 MiB: four rows per workgroup iteration on the persistent  */ \
      /* grid, 16 idle issue cycles between a lane's loads: 13.5-13.65 -> 12.66 us on 4096 x 4096 bf16 (profiles/r05_tune_rows_pace.txt; */ \
      /* four rows alone: 12.9-13.2)                                                                                                    */ \
      if (mid && !rc.bfp_B) {                                                                                         \
        DMXQ_LAUNCH((layernorm_block_kernel<D_, E_, V_, RMS, CAST, 4>),                                               \
                    dim3((unsigned)resident_grid(layernorm_block_kernel<D_, E_, V_, RMS, CAST, 4>, (rows + 3) / 4)), dim3(kThreads), 0, s, \
                    in, out, rows, cols, weight, bias, eps, rc);                                                      \
        break;                                                                                                        \
      }                                                                                                               \
    }                                                                                                                 \
    if constexpr (!CAST && rpw_mid != rpw) {                                                                          \
      if (mid) {                                                                                                      \
        DMXQ_LAUNCH((layernorm_block_kernel<D_, E_, V_, RMS, CAST, rpw_mid, true>), dim3(one_pass_grid((rows + rpw_mid - 1) / rpw_mid)),   \
                    dim3(kThreads), 0, s, in, out, rows, cols, weight, bias, eps, rc);                                \
        break;                                                                                                        \
      }                                                                                                               \
    }                                                                                                                 \
    DMXQ_LAUNCH((layernorm_block_kernel<D_, E_, V_, RMS, CAST>),                                                    \
                       dim3((unsigned)resident_grid(layernorm_block_kernel<D_, E_, V_, RMS, CAST>, (rows + rpw - 1) / rpw)), \
                       dim3(kThreads), 0, s, in, out, rows, cols, weight, bias, eps, rc);                             \
  } while (0)
#define DMXQ_LNB_V(D_, E_)                                                                                            \
  do {                                                                                                                \
    switch (vpl) {                                                                                                    \
      case 2: DMXQ_LNB(D_, E_, 2); break; case 3: DMXQ_LNB(D_, E_, 3); break; case 4: DMXQ_LNB(D_, E_, 4); break;      \
      case 5: DMXQ_LNB(D_, E_, 5); break; case 6: DMXQ_LNB(D_, E_, 6); break; default: DMXQ_LNB(D_, E_, 8); break;     \
    }                                                                                                                 \
  } while (0)
    if (dtype_in == DMXQ_F32) DMXQ_LNB_V(DMXQ_F32, 4);
    else if (dtype_in == DMXQ_BF16) DMXQ_LNB_V(DMXQ_BF16, 8);
    else DMXQ_LNB_V(DMXQ_F16, 8);
#undef DMXQ_LNB_V
#undef DMXQ_LNB
    return launch_status();
  }
  if (epl && cols <= (int64_t)64 * epl * 16) {
    int lpr, vpl;
    wave_shape(cols / epl, &lpr, &vpl);
#define DMXQ_LN(D_, E_, V_, L_)                                                                                       \
  do {                                                                                                                \
    constexpr int per_wg = 4 * rows_per_wave(V_, E_) * (64 / (L_));                                                   \
    if constexpr (CAST) {                                                                                             \
      if (rc.bfp_B) {                                                                                                 \
        const int lpb = rc.bfp_B / (E_);                                                                              \
        if (rc.bfp_B % (E_) != 0 || lpb < 1 || (lpb & (lpb - 1)) != 0 || lpb > (L_)) return DMXQ_ERR_UNSUPPORTED;      \
        RowCast rb = rc;                                                                                              \
        rb.bfp_lpb = lpb;                                                                                             \
        DMXQ_LAUNCH((layernorm_wave_kernel<D_, E_, V_, L_, RMS, CAST, 0, 0, -1, true>),                             \
                    dim3((unsigned)resident_grid(layernorm_wave_kernel<D_, E_, V_, L_, RMS, CAST, 0, 0, -1, true>, (rows + per_wg - 1) / per_wg)), \
                    dim3(kThreads), 0, s, in, out, rows, cols, weight, bias, eps, rb);                                \
        break;                                                                                                        \
      }                                                                                                               \
    }                                                                                                                 \
    if constexpr (CAST && !RMS && (V_) == 3 && (((E_) == 8 && (L_) == 32) || (kLn32OnePass > 0 && (E_) == 4 && (L_) == 64))) {   \
      /* the LayerNorm MODULE on 16-bit rows of 768 (opt-125m / Whisper-small hidden size): ONE pass per workgroup instead of the persistent */ \
      /* grid, with 32 idle issue cycles between a wave's loads (layernorm_wave_kernel kLnModulePace).  Same-lease library A/B, 24000 x 768  */ \
      /* bf16: 15.26 -> 14.32 us (60.4 -> 64.4 %; pace 2: 14.55; four rows per wave on the persistent grid, pace 0 / 2: 15.50 / 15.53);       */ \
      /* 32768 / 48000 / 96000 rows: 19.9 -> 19.6, 28.3 -> 26.9, 53.4 -> 51.6 us (profiles/r05_tune_rows_pace.txt)                            */ \
      DMXQ_LAUNCH((layernorm_wave_kernel<D_, E_, V_, L_, RMS, CAST>), dim3(one_pass_grid((rows + per_wg - 1) / per_wg)), dim3(kThreads), 0, s, \
                  in, out, rows, cols, weight, bias, eps, rc);                                                        \
      break;                                                                                                          \
    }                                                                                                                 \
    DMXQ_LAUNCH((layernorm_wave_kernel<D_, E_, V_, L_, RMS, CAST>),                                                 \
                       dim3((unsigned)resident_grid(layernorm_wave_kernel<D_, E_, V_, L_, RMS, CAST>, (rows + per_wg - 1) / per_wg)), \
                       dim3(kThreads), 0, s, in, out, rows, cols, weight, bias, eps, rc);                             \
  } while (0)
#define DMXQ_LN_V(D_, E_)                                                                            \
  do {                                                                                               \
    if (lpr == 32) { if (vpl == 1) DMXQ_LN(D_, E_, 1, 32); else if (vpl == 3) DMXQ_LN(D_, E_, 3, 32); else DMXQ_LN(D_, E_, 5, 32); } \
    else switch (vpl) {                                                                              \
      case 1: DMXQ_LN(D_, E_, 1, 64); break;   case 2: DMXQ_LN(D_, E_, 2, 64); break;                 \
      case 3: DMXQ_LN(D_, E_, 3, 64); break;   case 4: DMXQ_LN(D_, E_, 4, 64); break;                 \
      case 5: DMXQ_LN(D_, E_, 5, 64); break;   case 6: DMXQ_LN(D_, E_, 6, 64); break;                 \
      case 8: DMXQ_LN(D_, E_, 8, 64); break;   case 10: DMXQ_LN(D_, E_, 10, 64); break;               \
      case 12: DMXQ_LN(D_, E_, 12, 64); break; default: DMXQ_LN(D_, E_, 16, 64); break;               \
    }                                                                                                \
  } while (0)
    if (dtype_in == DMXQ_F32) DMXQ_LN_V(DMXQ_F32, 4);
    else if (dtype_in == DMXQ_BF16) { if (epl == 8) DMXQ_LN_V(DMXQ_BF16, 8); else DMXQ_LN_V(DMXQ_BF16, 4); }
    else { if (epl == 8) DMXQ_LN_V(DMXQ_F16, 8); else DMXQ_LN_V(DMXQ_F16, 4); }
#undef DMXQ_LN_V
#undef DMXQ_LN
    return launch_status();
  }
  if constexpr (CAST) return DMXQ_ERR_UNSUPPORTED;  // the fused-cast form exists for register-resident rows only
  const size_t scratch = (kThreads / kWave) * sizeof(float);
  if (cols <= kRowLdsFloats)
    DMXQ_LAUNCH((layernorm_rows_kernel<true, RMS>), dim3(row_grid(rows)), dim3(kThreads), scratch + cols * sizeof(float),
                       s, in, out, dtype_in, dtype_out, rows, cols, weight, bias, dtype_wb, eps);
  else
    DMXQ_LAUNCH((layernorm_rows_kernel<false, RMS>), dim3(row_grid(rows)), dim3(kThreads), scratch, s, in, out,
                       dtype_in, dtype_out, rows, cols, weight, bias, dtype_wb, eps);
  return launch_status();
}
#endif

// (one object per (RMS, CAST) instantiation of norm_dispatch -- ~70-140 kernels each: build.py compiles parts 2-5 in parallel)
#if DMXQ_AP(2)
extern "C" int dmxq_layernorm(const void* in, void* out, int dtype_in, int dtype_out, int64_t rows, int64_t cols,
                              const void* weight, const void* bias, int dtype_wb, float eps, void* stream) {
  return norm_dispatch<false>(in, out, dtype_in, dtype_out, rows, cols, weight, bias, dtype_wb, eps, stream);
}

#endif
#if DMXQ_AP(4)
// RMSNorm (torch_modules.py:1144-1170 -> F.rms_norm): the same register-resident row kernels without the centring pass:
// y = x * rsqrt(mean(x^2) + eps) * weight, fp32, one rounding.  (1 / sqrtf(.) here vs torch's rsqrt: both correctly
// rounded to within an ulp of fp32, far inside the tolerance the tests state for 16-bit outputs.)
extern "C" int dmxq_rmsnorm(const void* in, void* out, int dtype_in, int dtype_out, int64_t rows, int64_t cols,
                            const void* weight, int dtype_w, float eps, void* stream) {
  return norm_dispatch<true>(in, out, dtype_in, dtype_out, rows, cols, weight, nullptr, dtype_w, eps, stream);
}

#endif
// A LayerNorm / RMSNorm DmxModule in one pass: out = cast_out(norm(cast_in(x); weight, bias)), weight / bias in the row dtype.
#if DMXQ_AP(3)
extern "C" int dmxq_layernorm_cast(const void* in, void* out, int dtype, int64_t rows, int64_t cols, const void* weight, const void* bias,
                                   float eps, const dmxq_float_fmt* cast_in, const dmxq_float_fmt* cast_out, void* stream) {
  RowCast rc;
  if (!valid_dtype(dtype)) return DMXQ_ERR_BAD_ARG;
  if (!rowcast_of(dtype, cast_in, cast_out, &rc)) return DMXQ_ERR_UNSUPPORTED;
  return norm_dispatch<false, true>(in, out, dtype, dtype, rows, cols, weight, bias, dtype, eps, stream, rc);
}
#endif
#if DMXQ_AP(5)
extern "C" int dmxq_rmsnorm_cast(const void* in, void* out, int dtype, int64_t rows, int64_t cols, const void* weight, float eps,
                                 const dmxq_float_fmt* cast_in, const dmxq_float_fmt* cast_out, void* stream) {
  RowCast rc;
  if (!valid_dtype(dtype)) return DMXQ_ERR_BAD_ARG;
  if (!rowcast_of(dtype, cast_in, cast_out, &rc)) return DMXQ_ERR_UNSUPPORTED;
  return norm_dispatch<true, true>(in, out, dtype, dtype, rows, cols, weight, nullptr, dtype, eps, stream, rc);
}
#endif
// ... and with the BFP input cast of the modules that consume the result (the q / k / v or gate / up Linears after a pre-attention /
// pre-MLP norm: identical `input_casts` formats, modeling/nn/core.py:228-264) applied in the same launch; bit-identical to
// dmxq_layernorm_cast / dmxq_rmsnorm_cast followed by dmxq_bfp_qdq(.., block_size, precision, nearest, symmetric).
#if DMXQ_AP(3)
extern "C" int dmxq_layernorm_cast_bfp(const void* in, void* out, int dtype, int64_t rows, int64_t cols, const void* weight, const void* bias,
                                       float eps, const dmxq_float_fmt* cast_in, const dmxq_float_fmt* cast_out, int64_t block_size,
                                       int precision, void* stream) {
  RowCast rc;
  if (!valid_dtype(dtype) || block_size < 1) return DMXQ_ERR_BAD_ARG;
  if (precision < 2 || precision > 20 || block_size > 512) return DMXQ_ERR_UNSUPPORTED;
  if (!rowcast_of(dtype, cast_in, cast_out, &rc)) return DMXQ_ERR_UNSUPPORTED;
  rc.bfp_B = (int)block_size;
  rc.bfp_wl = precision;
  return norm_dispatch<false, true>(in, out, dtype, dtype, rows, cols, weight, bias, dtype, eps, stream, rc);
}
#endif
#if DMXQ_AP(5)
extern "C" int dmxq_rmsnorm_cast_bfp(const void* in, void* out, int dtype, int64_t rows, int64_t cols, const void* weight, float eps,
                                     const dmxq_float_fmt* cast_in, const dmxq_float_fmt* cast_out, int64_t block_size, int precision,
                                     void* stream) {
  RowCast rc;
  if (!valid_dtype(dtype) || block_size < 1) return DMXQ_ERR_BAD_ARG;
  if (precision < 2 || precision > 20 || block_size > 512) return DMXQ_ERR_UNSUPPORTED;
  if (!rowcast_of(dtype, cast_in, cast_out, &rc)) return DMXQ_ERR_UNSUPPORTED;
  rc.bfp_B = (int)block_size;
  rc.bfp_wl = precision;
  return norm_dispatch<true, true>(in, out, dtype, dtype, rows, cols, weight, nullptr, dtype, eps, stream, rc);
}
#endif  // parts 2-5 (norms)
