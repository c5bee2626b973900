// csrc/bfp_rows.hpp — the hot kernel: BFP Q->DQ over a flat stream of row blocks (block_dim = -1, L % B == 0).
// Shared by bfp.hip (product) and tools/tune_bfp.hip (variant A/B harness).
//
// Layout: the tensor is n_vec 16-byte input vectors; one lane owns one vector per step (8 x bf16/fp16 or
// 4 x fp32), a block of B elements spans lpb = B/EPL adjacent lanes of one wave, and the block max is an
// integer max over abs bit patterns reduced with DPP moves — no LDS, no second pass, each element is read
// once and written once (2 + 2 B/element for 16-bit I/O: HBM-bound).
//
// Work decomposition: workgroup-contiguous TILES of THREADS*UNROLL vectors (64 KiB of bf16 at 256 x 16).  A full
// tile issues all UNROLL loads unconditionally and back to back (256 B in flight per lane), then converts and
// stores vector by vector behind counted vmcnt waits; only the last, partial tile takes the predicated path.
// MODE bits: 1 = non-temporal loads, 2 = non-temporal stores (tools/tune_bfp: both help on a streaming pass).
#pragma once
#include "bfp_math.hpp"

namespace dmxq {

constexpr int kRowsNtLoad = 1, kRowsNtStore = 2;
constexpr int kRowsSc1Store = 4, kRowsSc0Store = 8;  // experiment flavours (tools/tune_bfp): write-through / sc0 stores via inline asm

// 16-byte stores with an explicit cache policy (KIND 2: sc1, 3: sc0); everything else through store_out
template <int DTO, int EPL, int KIND>
__device__ __forceinline__ void store_out_kind(void* p, const OutVec<DTO, EPL>& o) {
  constexpr int W = OutVec<DTO, EPL>::kWords;
  if constexpr (W % 4 == 0 && KIND >= 2) {
#pragma unroll
    for (int k = 0; k < W; k += 4) {
      const u32x4 v = {o.w[k], o.w[k + 1], o.w[k + 2], o.w[k + 3]};
      u32x4* dst = (u32x4*)p + k / 4;
      if (KIND == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst), "v"(v) : "memory");
      else asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(dst), "v"(v) : "memory");
    }
  } else {
    store_out<DTO, EPL, KIND == 1>(p, o);
  }
}

// one 16-byte input vector -> its packed outputs, given the block's max bits.  PATH_FAST selects the magic-add
// arithmetic (nearest-even only); `vi` = index of the input vector (numbers the random draws).
// EPLV = elements per lane-vector: 16 / sizeof(in), or half of that when the lane loads 8 bytes (see IVB below)
template <int DTI, int DTO, int RND, bool ASYM, int FAST, bool PATH_FAST, int EPLV = 16 / Elem<DTI>::bytes>
__device__ __forceinline__ OutVec<DTO, EPLV> bfp_rows_vector(const u32x4& raw, uint32_t mb, int64_t vi, int wl,
                                                             int rounding, bool stoch, uint64_t seed) {
  constexpr int EPL = EPLV;
  const BfpBlockParams p = bfp_block_params<ASYM, PATH_FAST && FAST != 4>(mb, wl);
  float xw[16 / Elem<DTI>::bytes], x[EPL], y[EPL];
  widen<DTI, 16 / Elem<DTI>::bytes>(raw, xw);
#pragma unroll
  for (int k = 0; k < EPL; k++) x[k] = xw[k];
  if (PATH_FAST && FAST == 4) {  // any rounding mode: literal rounding, clamp instead of the exponent-field clip
    const int64_t e0 = vi * EPL;
    if constexpr (RND == DMXQ_ROUND_STOCHASTIC && (EPL == 4 || EPL == 8)) {
      uint32_t rnd[EPL];
      bfp_rnd_vec<EPL>(seed, e0, rnd);   // one hash per lane-vector (common.hpp)
#pragma unroll
      for (int k = 0; k < EPL; k++) y[k] = bfp_q1_bitfast<RND, ASYM>(x[k], p, wl, rounding, rnd[k]);
    } else {
#pragma unroll
      for (int k = 0; k < EPL; k++)
        y[k] = bfp_q1_bitfast<RND, ASYM>(x[k], p, wl, rounding, bfp_rnd_for<RND>(stoch, seed, (uint64_t)(e0 + k)));
    }
  } else if (PATH_FAST) {
#pragma unroll
    for (int k = 0; k < EPL; k++) y[k] = bfp_q1_fast<FAST == 2, ASYM>(x[k], p);
  } else {
    const int64_t e0 = vi * EPL;
#pragma unroll
    for (int k = 0; k < EPL; k++)
      y[k] = bfp_q1<RND, ASYM>(x[k], p, wl, rounding, bfp_rnd_for<RND>(stoch, seed, (uint64_t)(e0 + k)));
  }
  // single-rounding fast path on a 16-bit input with the same 16-bit output: results are exactly representable
  // (not for asymmetric formats: their extra code -2^(e+1) overflows fp16 at e = 15 and must round to -inf)
  return pack_vec<DTO, EPL, PATH_FAST && FAST == 2 && DTO == DTI && !ASYM>(y);
}

// FAST: 0 = literal bit path only; 1 = magic-add path, double rounding; 2 = magic-add path, single rounding
// (valid only for RND == nearest; the dispatcher picks 2 when bfp_single_rounding_ok<DTI>(wl)); 4 = literal rounding
// in any mode + clamp (bfp_math.hpp (5); the runtime-rounding build).
// UNROLL vectors are in flight per lane; they are converted and stored in groups of GROUP: wait for the group's
// loads, quantise all of them into registers, then issue the group's stores back to back (read bursts and write
// bursts instead of a read/write interleave; tools/tune_bfp picks UNROLL and GROUP).
// IVB = input bytes per lane-vector: 16, or 8 when the output dtype is wider than the input (16-bit -> fp32): the
// lane then owns 4 elements, loads 8 B and still STORES 16 contiguous bytes, so every store instruction of a wave
// covers one contiguous KiB (32-B-per-lane outputs written as two strided 16-B halves cost ~40 % of the bandwidth).
template <int IVB>
__device__ __forceinline__ u32x4 load_rawv(const void* p, uint32_t off) {
  if (IVB == 16) return load_raw16<true>(p, off);
  const u32x2 t = __builtin_nontemporal_load((const u32x2*)((const char*)p + off));
  return u32x4{t.x, t.y, 0u, 0u};
}

// The LAST tile of a tensor that is not a whole number of tiles: bfp_rows_tile's schedule with predicated loads (zeros elsewhere: an
// all-zero block takes the fast path, and is never stored) and predicated stores.  A block never straddles the predicate (n_vec % lpb
// == 0, lpb | THREADS).  Until round 4 this tile ran vector by vector -- load, wait, quantise, store, UNROLL times in series: a tile
// that was 80-95 % full cost ~15 serial round trips, 19 us instead of 12 at 4300 x 4096 (tools/probe_deep.py), and made every one-round
// plan erratic between the sizes it was tuned on (profiles/r04_tune_bfp_fit.txt).  A function of its own, NOT inlined: sharing one
// body with the full tile (a generic lambda over a PARTIAL tag) changed the full tile's code -- outputs in place of the raw vectors,
// 108 instead of 229 VGPRs -- and cost the headline launch 9 % (tools/tune_bfp -DTUNE_MIN: 10.84 -> 11.81 us).
// `rem` = n_vec - (first vector of this lane): vector u of the lane exists iff u * THREADS < rem.
template <int DTI, int DTO, int RND, bool ASYM, int UNROLL, int MODE, int THREADS, int FAST, int GROUP, int IVB, int LPBC>
__device__ __attribute__((noinline)) void bfp_rows_tile_partial(const char* __restrict__ src, char* __restrict__ dst, int64_t rem, int64_t v0,
                                                                int lpb_rt, int wl, int rounding, bool stoch, uint64_t seed) {
  const int lpb = LPBC > 0 ? LPBC : lpb_rt;
  constexpr bool NTS = (MODE & kRowsNtStore) != 0;
  constexpr int SK = (MODE & kRowsSc1Store) ? 2 : ((MODE & kRowsSc0Store) ? 3 : (NTS ? 1 : 0));
  constexpr int EPL = IVB / Elem<DTI>::bytes;
  constexpr int OVB = EPL * Elem<DTO>::bytes;
  const uint32_t lane_in = threadIdx.x * (uint32_t)IVB, lane_out = threadIdx.x * (uint32_t)OVB;
  u32x4 raw[UNROLL];
#pragma unroll
  for (int u = 0; u < UNROLL; u++) {
    if ((int64_t)u * THREADS < rem) raw[u] = load_rawv<IVB>(src + u * (THREADS * IVB), lane_in);
    else raw[u] = u32x4{0u, 0u, 0u, 0u};
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int g = 0; g < UNROLL; g += GROUP) {
    uint32_t mb[GROUP];
    constexpr bool kFast = FAST == 4 || (FAST != 0 && RND == DMXQ_ROUND_NEAREST);
    bool all_fast = kFast;
#pragma unroll
    for (int u = 0; u < GROUP; u++) {
      mb[u] = group_max_u32(absmax_bits<DTI>(raw[g + u]), lpb);
      if (kFast) all_fast = all_fast && (FAST == 4 ? bfp_bitfast_ok(mb[u], RND == kRuntimeRounding ? rounding : RND) : bfp_fast_ok(mb[u], wl));
    }
    OutVec<DTO, EPL> o[GROUP];
#pragma unroll
    for (int u = 0; u < GROUP; u++) {
      o[u] = bfp_rows_vector<DTI, DTO, RND, ASYM, FAST, kFast, EPL>(raw[g + u], mb[u], v0 + (int64_t)(g + u) * THREADS, wl, rounding, stoch, seed);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (kFast && __builtin_amdgcn_ballot_w64(!all_fast) != 0ull) {  // rare: see bfp_rows_tile
#pragma unroll
      for (int u = 0; u < GROUP; u++) {
        u32x4 r = u32x4{0u, 0u, 0u, 0u};
        if ((int64_t)(g + u) * THREADS < rem) r = load_rawv<IVB>(src + (g + u) * (THREADS * IVB), lane_in);
        const uint32_t m = group_max_u32(absmax_bits<DTI>(r), lpb);
        if (__builtin_amdgcn_ballot_w64(!(FAST == 4 ? bfp_bitfast_ok(m, RND == kRuntimeRounding ? rounding : RND) : bfp_fast_ok(m, wl))) != 0ull)
          o[u] = bfp_rows_vector<DTI, DTO, RND, ASYM, FAST, false, EPL>(r, m, v0 + (int64_t)(g + u) * THREADS, wl, rounding, stoch, seed);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < GROUP; u++)
      if ((int64_t)(g + u) * THREADS < rem) store_out_kind<DTO, EPL, SK>(dst + (g + u) * (THREADS * OVB) + lane_out, o[u]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ONE tile (THREADS*UNROLL lane-vectors) of a flat tensor of n_vec vectors: `tile` is the tile index inside that tensor.
// Shared by the single-tensor kernel below and the multi-tensor kernel of bfp.hip.
// LPBC: lanes per block as a compile-time constant (0 = the runtime value `lpb_rt`): the DPP reduction of the block maximum
// then has no scalar branches (six `s_cbranch` + `s_nop` per vector otherwise, which also fence the VALU scheduling).
template <int DTI, int DTO, int RND, bool ASYM, int UNROLL, int MODE, int THREADS, int FAST, int GROUP, int IVB, int LPBC = 0, int PACE = 0>
__device__ __forceinline__ void bfp_rows_tile(const void* __restrict__ in, void* __restrict__ out, int64_t n_vec,
                                              int64_t tile, int lpb_rt, int wl, int rounding, bool stoch, uint64_t seed) {
  const int lpb = LPBC > 0 ? LPBC : lpb_rt;
  static_assert(UNROLL % GROUP == 0, "GROUP must divide UNROLL");
  constexpr bool NTS = (MODE & kRowsNtStore) != 0;
  constexpr int SK = (MODE & kRowsSc1Store) ? 2 : ((MODE & kRowsSc0Store) ? 3 : (NTS ? 1 : 0));
  constexpr int64_t TILE = (int64_t)THREADS * UNROLL;
  constexpr int EPL = IVB / Elem<DTI>::bytes;
  constexpr int OVB = EPL * Elem<DTO>::bytes;  // output bytes per input vector
  const uint32_t lane_in = threadIdx.x * (uint32_t)IVB, lane_out = threadIdx.x * (uint32_t)OVB;
  // workgroup-uniform tile bases + 32-bit lane offsets
  const char* src = (const char*)in + tile * (TILE * IVB);
  char* dst = (char*)out + tile * (TILE * OVB);
  const int64_t v0 = tile * TILE + threadIdx.x;
  if ((tile + 1) * TILE <= n_vec) {  // full tile (workgroup-uniform): no predicates
    u32x4 raw[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      raw[u] = load_rawv<IVB>(src + u * (THREADS * IVB), lane_in);
      if (u + 1 < UNROLL) pace_issue<PACE>();
    }
    __builtin_amdgcn_sched_barrier(0);  // every load is issued before any arithmetic: 16 B x UNROLL in flight per lane
#pragma unroll
    for (int g = 0; g < UNROLL; g += GROUP) {
      // block maxima of the whole group first; the fast/literal choice is made once per group, wave-uniformly
      // (every active lane's block must admit the magic-add path, see bfp_math.hpp)
      uint32_t mb[GROUP];
      constexpr bool kFast = FAST == 4 || (FAST != 0 && RND == DMXQ_ROUND_NEAREST);
      bool all_fast = kFast;
#pragma unroll
      for (int u = 0; u < GROUP; u++) {
        mb[u] = group_max_u32(absmax_bits<DTI>(raw[g + u]), lpb);
        if (kFast) all_fast = all_fast && (FAST == 4 ? bfp_bitfast_ok(mb[u], RND == kRuntimeRounding ? rounding : RND) : bfp_fast_ok(mb[u], wl));
      }
      OutVec<DTO, EPL> o[GROUP];
#pragma unroll
      for (int u = 0; u < GROUP; u++) {
        o[u] = bfp_rows_vector<DTI, DTO, RND, ASYM, FAST, kFast, EPL>(raw[g + u], mb[u], v0 + (int64_t)(g + u) * THREADS, wl,
                                                                 rounding, stoch, seed);
        __builtin_amdgcn_sched_barrier(0);  // vector by vector: short live ranges (4 workgroups per CU need <= 128 VGPRs)
      }
      if (kFast && __builtin_amdgcn_ballot_w64(!all_fast) != 0ull) {
        // rare: some block of this wave cannot take the magic-add path (bfp_fast_ok).  Redo the affected vectors
        // with the literal bit path from a fresh read of the inputs (still unmodified: this tile's stores come
        // later), instead of keeping every raw vector alive across a two-sided branch.  (Unrolled: a
        // runtime-indexed o[u] would live in scratch.)
#pragma unroll
        for (int u = 0; u < GROUP; u++) {
          const u32x4 r = load_rawv<IVB>(src + (g + u) * (THREADS * IVB), lane_in);
          const uint32_t m = group_max_u32(absmax_bits<DTI>(r), lpb);
          if (__builtin_amdgcn_ballot_w64(!(FAST == 4 ? bfp_bitfast_ok(m, RND == kRuntimeRounding ? rounding : RND) : bfp_fast_ok(m, wl))) != 0ull)
            o[u] = bfp_rows_vector<DTI, DTO, RND, ASYM, FAST, false, EPL>(r, m, v0 + (int64_t)(g + u) * THREADS, wl, rounding,
                                                                     stoch, seed);
        }
      }
      __builtin_amdgcn_sched_barrier(0);  // ... and the group's stores go out as one burst
#pragma unroll
      for (int u = 0; u < GROUP; u++) store_out_kind<DTO, EPL, SK>(dst + (g + u) * (THREADS * OVB) + lane_out, o[u]);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if constexpr (UNROLL >= 4) {
    // last, partial tile (workgroup-uniform): the same schedule with predicated loads and stores, in a function of its own
    bfp_rows_tile_partial<DTI, DTO, RND, ASYM, UNROLL, MODE, THREADS, FAST, GROUP, IVB, LPBC>(src, dst, n_vec - v0, v0, lpb, wl, rounding, stoch, seed);
  } else {
    // 1 or 2 vectors per lane (the multi-round plans): nothing to overlap, and a call in these kernels costs their FULL tiles 5-9 %
    // (bf16 -> float32 with run-time rounding, 64 MiB: 19.0 -> 20.6 us) -- vector by vector, inline
    for (int u = 0; u < UNROLL; u++) {
      const int64_t vi = v0 + (int64_t)u * THREADS;
      if (vi < n_vec) {
        const u32x4 raw = load_rawv<IVB>(src + u * (THREADS * IVB), lane_in);
        const uint32_t mb = group_max_u32(absmax_bits<DTI>(raw), lpb);
        const OutVec<DTO, EPL> o = bfp_rows_vector<DTI, DTO, RND, ASYM, FAST, false, EPL>(raw, mb, vi, wl, rounding, stoch, seed);
        store_out<DTO, EPL, NTS>(dst + u * (THREADS * OVB) + lane_out, o);
      }
    }
  }
}

template <int DTI, int DTO, int RND, bool ASYM, int UNROLL, int MODE, int THREADS, int FAST = 0, int GROUP = UNROLL,
          int IVB = 16, int LPBC = 0, int PACE = 0>
__global__ __launch_bounds__(THREADS) void bfp_rows_kernel(const void* __restrict__ in, void* __restrict__ out,
                                                          int64_t n_vec, int lpb_arg /*lanes per block*/, int wl,
                                                          int rounding, uint64_t seed) {
  constexpr int64_t TILE = (int64_t)THREADS * UNROLL;
  const bool stoch = RND == DMXQ_ROUND_STOCHASTIC || ((RND == kRuntimeRounding) && rounding == DMXQ_ROUND_STOCHASTIC);
  const int lpb = __builtin_amdgcn_readfirstlane(lpb_arg);
  const int64_t n_tiles = (n_vec + TILE - 1) / TILE;
#define DMXQ_TILE_LOOP(L_)                                                                                          \
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x)                                                \
    bfp_rows_tile<DTI, DTO, RND, ASYM, UNROLL, MODE, THREADS, FAST, GROUP, IVB, L_, PACE>(in, out, n_vec, tile, lpb, wl, rounding, stoch, seed)
  // One-round geometries (UNROLL >= 4) on the magic-add paths: the lanes-per-block of the usual block sizes as a compile-time
  // constant, chosen ONCE per launch -- with a runtime value every vector's DPP reduction is a chain of six scalar
  // branches, which cost 6.7 % of the 64 MiB headline launch (tools/tune_bfp: 11.71 -> 10.93 us).  Multi-round 512x2
  // tiles measured no difference and keep the single runtime form.
  if constexpr (LPBC == 0 && UNROLL >= 4 && (FAST == 1 || FAST == 2 || (FAST == 4 && RND != kRuntimeRounding))) {
    switch (lpb) {
      case 2: DMXQ_TILE_LOOP(2); break;
      case 4: DMXQ_TILE_LOOP(4); break;
      case 8: DMXQ_TILE_LOOP(8); break;
      case 16: DMXQ_TILE_LOOP(16); break;
      default: DMXQ_TILE_LOOP(0); break;
    }
  } else {
    DMXQ_TILE_LOOP(LPBC);
  }
#undef DMXQ_TILE_LOOP
}

// COMPACT form of the flat-stream kernel for the deepest one-round classes (19+ vectors per lane; symmetric 16-bit -> same 16-bit,
// nearest-even single rounding): every result overwrites the raw vector it came from, so a lane holds 4 x UNROLL registers of tile
// data instead of the ~14 per vector of bfp_rows_tile (whose whole-tile store group keeps raw AND result vectors alive: 229 VGPRs at 16
// vectors, scratch from 19).  A kernel of its own: giving bfp_rows_tile this form costs the 32 MiB launch 9 % (see bfp_rows_tile_partial).
template <int DT, int UNROLL, int THREADS, int GROUP, int LPBC>
__device__ __forceinline__ void bfp_rows_tile_compact(const char* __restrict__ src, char* __restrict__ dst, int64_t v0, int lpb_rt, int wl) {
  static_assert(UNROLL % GROUP == 0 && DT != DMXQ_F32, "16-bit tensors; GROUP divides UNROLL");
  constexpr int EPL = 8;
  const int lpb = LPBC > 0 ? LPBC : lpb_rt;
  const uint32_t lane = threadIdx.x * 16u;
  u32x4 raw[UNROLL];
#pragma unroll
  for (int u = 0; u < UNROLL; u++) raw[u] = load_rawv<16>(src + u * (THREADS * 16), lane);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int g = 0; g < UNROLL; g += GROUP) {
    uint32_t mb[GROUP];
    bool all_fast = true;
#pragma unroll
    for (int u = 0; u < GROUP; u++) {
      mb[u] = group_max_u32(absmax_bits<DT>(raw[g + u]), lpb);
      all_fast = all_fast && bfp_fast_ok(mb[u], wl);
    }
#pragma unroll
    for (int u = 0; u < GROUP; u++) {
      const OutVec<DT, EPL> o = bfp_rows_vector<DT, DT, DMXQ_ROUND_NEAREST, false, 2, true, EPL>(raw[g + u], mb[u], v0 + (int64_t)(g + u) * THREADS, wl,
                                                                                             DMXQ_ROUND_NEAREST, false, 0ull);
      raw[g + u] = u32x4{o.w[0], o.w[1], o.w[2], o.w[3]};
      __builtin_amdgcn_sched_barrier(0);
    }
    if (__builtin_amdgcn_ballot_w64(!all_fast) != 0ull) {  // rare: literal bit path from a fresh read (see bfp_rows_tile)
#pragma unroll
      for (int u = 0; u < GROUP; u++) {
        const u32x4 r = load_rawv<16>(src + (g + u) * (THREADS * 16), lane);
        const uint32_t m = group_max_u32(absmax_bits<DT>(r), lpb);
        if (__builtin_amdgcn_ballot_w64(!bfp_fast_ok(m, wl)) != 0ull) {
          const OutVec<DT, EPL> o = bfp_rows_vector<DT, DT, DMXQ_ROUND_NEAREST, false, 2, false, EPL>(r, m, v0 + (int64_t)(g + u) * THREADS, wl,
                                                                                                  DMXQ_ROUND_NEAREST, false, 0ull);
          raw[g + u] = u32x4{o.w[0], o.w[1], o.w[2], o.w[3]};
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < GROUP; u++) __builtin_nontemporal_store(raw[g + u], (u32x4*)(dst + (g + u) * (THREADS * 16) + lane));
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int DT, int UNROLL, int THREADS, int GROUP>
__global__ __launch_bounds__(THREADS) void bfp_rows_compact_kernel(const void* __restrict__ in, void* __restrict__ out, int64_t n_vec,
                                                                  int lpb_arg, int wl) {
  constexpr int64_t TILE = (int64_t)THREADS * UNROLL;
  constexpr int MODE = kRowsNtLoad | kRowsNtStore;
  const int lpb = __builtin_amdgcn_readfirstlane(lpb_arg);
  const int64_t tile = blockIdx.x;   // one tile per workgroup: the host launches exactly ceil(n_vec / TILE) of them
  const char* src = (const char*)in + tile * (TILE * 16);
  char* dst = (char*)out + tile * (TILE * 16);
  const int64_t v0 = tile * TILE + threadIdx.x;
  const bool full = (tile + 1) * TILE <= n_vec;
#define DMXQ_COMPACT(L_)                                                                                                              \
  do {                                                                                                                                \
    if (full) bfp_rows_tile_compact<DT, UNROLL, THREADS, GROUP, L_>(src, dst, v0, lpb, wl);                                             \
    else bfp_rows_tile_partial<DT, DT, DMXQ_ROUND_NEAREST, false, UNROLL, MODE, THREADS, 2, GROUP, 16, L_>(src, dst, n_vec - v0, v0, lpb, wl, \
                                                                                                         DMXQ_ROUND_NEAREST, false, 0ull); \
  } while (0)
  switch (lpb) {
    case 2: DMXQ_COMPACT(2); break;
    case 4: DMXQ_COMPACT(4); break;
    case 8: DMXQ_COMPACT(8); break;
    case 16: DMXQ_COMPACT(16); break;
    default: DMXQ_COMPACT(0); break;
  }
#undef DMXQ_COMPACT
}

// Multi-tensor form: up to kMultiMax flat tensors in ONE launch (small weights are launch-bound one by one: an empty
// launch costs ~1.6 us, a 768x768 bf16 tensor streams in 0.4 us).  The tile space of all tensors is concatenated;
// a workgroup finds its tensor with a scalar search over the descriptors (kernel arguments: s_load, no memory traffic).
constexpr int kMultiMax = 48;
constexpr int kMultiThreads = 256, kMultiUnroll = 4;  // 16 KiB (16-bit) tiles: tensors of any size balance over the CUs (opt-125m's 73 weights: 256 x 2 98.3 us, x 4 94.2, x 8 96.8)
struct MultiDesc { const void* in; void* out; int64_t n_vec; int64_t tile0; /* first tile of this tensor */ };
struct MultiArgs { MultiDesc d[kMultiMax]; int n; int lpb, wl; };

// Round 5 (as stream.hpp stream_multi_kernel): the first tiles of tensors 1 .. 10 arrive as ten SCALAR arguments -- preloaded into SGPRs
// with the wave -- so that a set of up to 11 tensors is resolved without touching the argument block (scanned from it, a workgroup made two
// dependent scalar-memory round trips before its first load); larger sets finish the scan in the block.  MODE: non-temporal stores for
// large sets only -- a layer's quantised weights (<= 32 MiB) are read back by its GEMMs at once and fit the Infinity Cache.
constexpr int kMultiPre = 10;
template <int DTI, int DTO, bool ASYM, int UNROLL, int THREADS, int FAST, int IVB = 16, int MODE = kRowsNtLoad | kRowsNtStore>
__global__ __launch_bounds__(THREADS) void bfp_rows_multi_kernel(uint32_t e0, uint32_t e1, uint32_t e2, uint32_t e3, uint32_t e4, uint32_t e5,
                                                                uint32_t e6, uint32_t e7, uint32_t e8, uint32_t e9, const MultiArgs a) {
  const uint32_t tile = blockIdx.x;
  int k = (e0 <= tile) + (e1 <= tile) + (e2 <= tile) + (e3 <= tile) + (e4 <= tile) + (e5 <= tile) + (e6 <= tile) + (e7 <= tile) + (e8 <= tile) +
          (e9 <= tile);
  if (k == kMultiPre) {
    for (int i = kMultiPre + 1; i < a.n; i++) k = ((uint32_t)a.d[i].tile0 <= tile) ? i : k;  // tile0 ascending
  }
  const int lpb = __builtin_amdgcn_readfirstlane(a.lpb);
  bfp_rows_tile<DTI, DTO, DMXQ_ROUND_NEAREST, ASYM, UNROLL, MODE, THREADS, FAST, UNROLL, IVB>(
      a.d[k].in, a.d[k].out, a.d[k].n_vec, (int64_t)tile - a.d[k].tile0, lpb, a.wl, DMXQ_ROUND_NEAREST, false, 0ull);
}

}  // namespace dmxq
