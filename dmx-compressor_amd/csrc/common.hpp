// csrc/common.hpp — shared device helpers for the gfx950 (CDNA4, wave64) kernels of libdmxq.
//
// Everything here is bit-level fp32 arithmetic that must match the reference's CPU path exactly
// (quant/quant_cpu/quant_cpu.cpp, bit_helper.cpp, sim_helper.cpp of d-matrix-ai/dmx-compressor), so the
// library is built with -fno-fast-math -ffp-contract=off and keeps fp32 denormals on (gfx950 default).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/dmxq.h"

namespace dmxq {

constexpr int kWave = 64;        // CDNA4 wavefront width
constexpr int kThreads = 256;    // workgroup size: 4 waves, one per SIMD
constexpr int kMaxBlocks = 256 * 8;  // memory-bound grid cap: 256 CUs x 8 workgroups, grid-stride beyond

__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }

// hash32(seed, linear element index): the same counter-based stream as oracle/oracle.c rnd_bits, so the
// stochastic modes are reproducible and kernel-vs-oracle bit-exact (the reference's RNG is an unseeded
// global mt19937, quant_cpu.cpp:32-34: only statistical parity is possible against it).  32-bit avalanche hash of
// the index's low half, keyed by the seed and the index's high half: two v_mul_lo_u32 per draw instead of the
// twelve a 64-bit mixer needs.
__device__ __forceinline__ uint32_t rnd_bits(uint64_t seed, uint64_t idx) {
  const uint32_t key = (uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x9E3779B9u) ^ ((uint32_t)(idx >> 32) * 0x85EBCA6Bu);
  uint32_t x = (uint32_t)idx ^ key;
  x ^= x >> 16; x *= 0x7FEB352Du;
  x ^= x >> 15; x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}

// random bits only when the (wave-uniform) mode is stochastic: the empty asm keeps the optimiser from turning the
// branch into an unconditional evaluation + select (the hash is ~10 VALU ops, two of them quarter-rate multiplies)
__device__ __forceinline__ uint32_t rnd_if(bool stoch, uint64_t seed, uint64_t idx) {
  uint32_t r = 0u;
  if (stoch) {
    r = rnd_bits(seed, idx);
    asm volatile("" : "+v"(r));
  }
  return r;
}

// The stochastic draws of the BFP cast (round 5; oracle/oracle.c bfp_rnd is the same function): ONE avalanche hash per aligned group of
// 8 elements, expanded to the group's four pair words by a Weyl step, a xor-shift, a 24-BIT multiply (v_mul_u32_u24: full rate, unlike
// the 32-bit one) and another xor-shift; an element takes its pair's word (even index) or that word with its halves swapped (odd
// index) -- round_bitwise uses the low 23 - wl bits, <= 16 of them for every format with wl >= 7, so the two elements of a pair draw
// from disjoint halves.  ~5 operations per element when a lane holds the whole group (bfp_rnd_vec) instead of a hash with two
// quarter-rate multiplies per element (~15 cycles' worth): the stochastic mode of the row kernel was bound by it (50 % of the
// roofline).  Quality is MEASURED, not assumed (tests/test_golden.py): P(round up) equals the dropped fraction for every element
// position, and the rounding decisions of any two positions of a group are uncorrelated (|r| < 0.01 over 2 x 10^5 groups) -- a first
// expansion without the multiply (Weyl step + xor-shift only) was 15 % faster and left neighbouring pairs correlated at r = 0.46:
// rejected.  The stream is this library's own -- the reference's is an unseeded global mt19937 -- so kernel and oracle changed
// together; an element's draw still depends on (seed, linear index) only.
__device__ __forceinline__ uint32_t bfp_rnd_word(uint32_t h, uint32_t pair) {
  if (pair == 0u) return h;   // (the group's hash itself serves its first pair)
  uint32_t w = h + pair * 0x9E3779B9u;
  w ^= w >> 15;
  w = __umul24(w, 0xB5297Bu);   // (low 24 bits of w) x K, low 32 bits of the product
  return w ^ (w >> 12);
}
__device__ __forceinline__ uint32_t bfp_rnd(uint64_t seed, uint64_t idx) {
  const uint32_t w = bfp_rnd_word(rnd_bits(seed, idx >> 3), ((uint32_t)idx & 7u) >> 1);
  return ((uint32_t)idx & 1u) ? __builtin_amdgcn_alignbit(w, w, 16) : w;
}
__device__ __forceinline__ uint32_t bfp_rnd_if(bool stoch, uint64_t seed, uint64_t idx) {
  uint32_t r = 0u;
  if (stoch) {
    r = bfp_rnd(seed, idx);
    asm volatile("" : "+v"(r));
  }
  return r;
}
// the draw of element `idx` for a kernel built for rounding mode RND (compile time), or for any mode (kRuntimeRounding: `stoch` says
// whether the launch is stochastic)
template <int RND>
__device__ __forceinline__ uint32_t bfp_rnd_for(bool stoch, uint64_t seed, uint64_t idx) {
  if (RND == -1 /* kRuntimeRounding */) return bfp_rnd_if(stoch, seed, idx);
  return RND == DMXQ_ROUND_STOCHASTIC ? bfp_rnd(seed, idx) : 0u;
}
// EPL (4 or 8) consecutive draws starting at e0, e0 % EPL == 0: one hash, then the expansion
template <int EPL>
__device__ __forceinline__ void bfp_rnd_vec(uint64_t seed, int64_t e0, uint32_t (&r)[EPL]) {
  static_assert(EPL == 4 || EPL == 8, "a lane-vector of 4 or 8 elements");
  const uint32_t h = rnd_bits(seed, (uint64_t)e0 >> 3);
  const uint32_t p0 = EPL == 8 ? 0u : ((uint32_t)e0 & 4u) >> 1;   // first pair of this vector inside its group of 8
#pragma unroll
  for (int k = 0; k < EPL; k += 2) {
    const uint32_t w = bfp_rnd_word(h, p0 + (uint32_t)(k >> 1));
    r[k] = w;
    r[k + 1] = __builtin_amdgcn_alignbit(w, w, 16);
  }
}

// quant_cpu.cpp:211-237 round_bitwise: keep `man_bits` (0..22) mantissa bits of an fp32 bit pattern.
// nearest == round-half-to-even on the bit pattern, written as the branch-free (half-1)+lsb form:
// dropped > half carries, dropped < half does not, dropped == half carries iff the kept LSB is odd.
// RND is a compile-time DMXQ_ROUND_NEAREST on the hot instantiations, or kRuntimeRounding, in which case the
// (wave-uniform) `rounding` argument selects the mode.
constexpr int kRuntimeRounding = -1;
template <int RND>
__device__ __forceinline__ uint32_t round_bitwise(uint32_t t, int man_bits, int rounding, uint32_t rnd) {
  const int sh = 23 - man_bits;
  const uint32_t mask = (1u << sh) - 1u;
  const int r = (RND == kRuntimeRounding) ? rounding : RND;
  uint32_t add;
  if (r == DMXQ_ROUND_NEAREST) add = (mask >> 1) + ((t >> sh) & 1u);
  else if (r == DMXQ_ROUND_DOWN) add = 0u;
  else if (r == DMXQ_ROUND_UP) add = 1u << sh;
  else add = rnd & mask;
  return (t + add) & ~mask;
}

// ---------------------------------------------------------------------------------------------------------
// 16-byte vector I/O.  One lane always moves 16 B of input per step (global_load_dwordx4): 8 x bf16/fp16 or
// 4 x fp32.  Elements are widened to fp32 (exact) for the arithmetic and narrowed with a true RNE convert
// (what `.to(physical_dtype)` does in numerical/cast.py:306).
template <int DT> struct Elem;
template <> struct Elem<DMXQ_F32> { static constexpr int bytes = 4; };
template <> struct Elem<DMXQ_F16> { static constexpr int bytes = 2; };
template <> struct Elem<DMXQ_BF16> { static constexpr int bytes = 2; };

__device__ __forceinline__ float opaque(float v);
template <int DT>
__device__ __forceinline__ float load1(const void* p, int64_t i) {
  if (DT == DMXQ_F32) return ((const float*)p)[i];
  if (DT == DMXQ_F16) return opaque((float)((const _Float16*)p)[i]);
  return u2f((uint32_t)((const uint16_t*)p)[i] << 16);
}

template <int DT>
__device__ __forceinline__ void store1(void* p, int64_t i, float v) {
  if (DT == DMXQ_F32) ((float*)p)[i] = v;
  else if (DT == DMXQ_F16) ((_Float16*)p)[i] = (_Float16)opaque(v);
  else ((__bf16*)p)[i] = (__bf16)v;  // v_cvt_pk_bf16_f32: RNE, NaN-preserving
}

// runtime-dtype scalar access (cold paths: wave-uniform switch)
__device__ __forceinline__ float load_rt(const void* p, int dt, int64_t i) {
  if (dt == DMXQ_F32) return load1<DMXQ_F32>(p, i);
  if (dt == DMXQ_F16) return load1<DMXQ_F16>(p, i);
  return load1<DMXQ_BF16>(p, i);
}
__device__ __forceinline__ void store_rt(void* p, int dt, int64_t i, float v) {
  if (dt == DMXQ_F32) store1<DMXQ_F32>(p, i, v);
  else if (dt == DMXQ_F16) store1<DMXQ_F16>(p, i, v);
  else store1<DMXQ_BF16>(p, i, v);
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// fp16 halves of a dword -> fp32 (explicit shifts: bit_cast of an ext-vector element to a 2 x f16 vector
// was observed to read element 0 for every index under hipcc 7.2)
// opaque(): keeps hipcc from folding an fp16<->fp32 conversion into a neighbouring multiply as v_fma_mix*(a, b, +0):
// that form adds +0.0 and turns a -0.0 product into +0.0 (observed: `(q - zp) * sc` stored as fp16).
__device__ __forceinline__ float opaque(float v) { asm("" : "+v"(v)); return v; }  // not volatile: free to schedule
__device__ __forceinline__ float half_lo(uint32_t w) { return opaque((float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xFFFFu))); }
__device__ __forceinline__ float half_hi(uint32_t w) { return opaque((float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16))); }

// N fp32 values <- N consecutive elements starting at element index i (N*bytes must be 16 or 32 aligned as used)
template <int DT, int N>
__device__ __forceinline__ void load_vec(const void* p, int64_t i, float (&x)[N]) {
  if (DT == DMXQ_F32) {
#pragma unroll
    for (int k = 0; k < N; k += 4) {
      f32x4 v = *(const f32x4*)((const float*)p + i + k);
      x[k] = v.x; x[k + 1] = v.y; x[k + 2] = v.z; x[k + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int k = 0; k < N; k += 8) {
      u32x4 v = *(const u32x4*)((const uint16_t*)p + i + k);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (DT == DMXQ_BF16) {
          x[k + 2 * j] = u2f(v[j] << 16);
          x[k + 2 * j + 1] = u2f(v[j] & 0xFFFF0000u);
        } else {
          x[k + 2 * j] = half_lo(v[j]);
          x[k + 2 * j + 1] = half_hi(v[j]);
        }
      }
    }
  }
}

template <int DT>
__device__ __forceinline__ uint32_t pack2(float a, float b) {
  if (DT == DMXQ_BF16) {
    bf16x2 h; h.x = (__bf16)a; h.y = (__bf16)b;
    return __builtin_bit_cast(uint32_t, h);
  } else {
    f16x2 h; h.x = (_Float16)opaque(a); h.y = (_Float16)opaque(b);
    return __builtin_bit_cast(uint32_t, h);
  }
}

template <int DT, int N, bool NT = false>
__device__ __forceinline__ void store_vec(void* p, int64_t i, const float (&y)[N]) {
  if (DT == DMXQ_F32) {
#pragma unroll
    for (int k = 0; k < N; k += 4) {
      f32x4 v = {y[k], y[k + 1], y[k + 2], y[k + 3]};
      f32x4* dst = (f32x4*)((float*)p + i + k);
      if (NT) __builtin_nontemporal_store(v, dst); else *dst = v;
    }
  } else if (N % 8 == 0) {
#pragma unroll
    for (int k = 0; k < N; k += 8) {
      u32x4 v = {pack2<DT>(y[k], y[k + 1]), pack2<DT>(y[k + 2], y[k + 3]), pack2<DT>(y[k + 4], y[k + 5]),
                 pack2<DT>(y[k + 6], y[k + 7])};
      u32x4* dst = (u32x4*)((uint16_t*)p + i + k);
      if (NT) __builtin_nontemporal_store(v, dst); else *dst = v;
    }
  } else {  // N == 4 sixteen-bit outputs: one 8-byte store
    u32x2 v = {pack2<DT>(y[0], y[1]), pack2<DT>(y[2], y[3])};
    u32x2* dst = (u32x2*)((uint16_t*)p + i);
    if (NT) __builtin_nontemporal_store(v, dst); else *dst = v;
  }
}

// packed output vector: N elements of dtype DT held in registers until the store burst
template <int DT, int N>
struct OutVec {
  static constexpr int kWords = N * Elem<DT>::bytes / 4;
  uint32_t w[kWords];
};

// EXACT16: the caller guarantees every value is exactly representable in the 16-bit output dtype (BFP results of
// 16-bit inputs): fp16 then packs with v_cvt_pkrtz_f16_f32 (one instruction per pair; truncation == RNE when exact)
template <int DT, int N, bool EXACT16 = false>
__device__ __forceinline__ OutVec<DT, N> pack_vec(const float (&y)[N]) {
  OutVec<DT, N> o;
  if (DT == DMXQ_F32) {
#pragma unroll
    for (int k = 0; k < N; k++) o.w[k] = f2u(y[k]);
  } else if (DT == DMXQ_F16 && EXACT16) {
#pragma unroll
    for (int k = 0; k < N / 2; k++) o.w[k] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(y[2 * k], y[2 * k + 1]));
  } else {
#pragma unroll
    for (int k = 0; k < N / 2; k++) o.w[k] = pack2<DT>(y[2 * k], y[2 * k + 1]);
  }
  return o;
}

// UNAL: the address is only element-aligned.  gfx950 serves such 16-byte accesses at 93-97 % of the aligned rate
// (profiles/r01_unaligned_access.txt); the 2-byte-aligned pointer types only stop the compiler from assuming more.
typedef u32x4 u32x4_unal __attribute__((aligned(2)));
typedef u32x2 u32x2_unal __attribute__((aligned(2)));
template <int DT, int N, bool NT, bool UNAL = false>
__device__ __forceinline__ void store_out(void* p, const OutVec<DT, N>& o) {
  constexpr int W = OutVec<DT, N>::kWords;
  if (W % 4 == 0) {
#pragma unroll
    for (int k = 0; k < W; k += 4) {
      u32x4 v = {o.w[k], o.w[k + 1], o.w[k + 2], o.w[k + 3]};
      if (UNAL) { u32x4_unal* dst = (u32x4_unal*)p + k / 4; if (NT) __builtin_nontemporal_store(v, dst); else *dst = v; }
      else { u32x4* dst = (u32x4*)p + k / 4; if (NT) __builtin_nontemporal_store(v, dst); else *dst = v; }
    }
  } else {  // 2 words: 4 sixteen-bit outputs
    u32x2 v = {o.w[0], o.w[1]};
    if (UNAL) { if (NT) __builtin_nontemporal_store(v, (u32x2_unal*)p); else *(u32x2_unal*)p = v; }
    else { if (NT) __builtin_nontemporal_store(v, (u32x2*)p); else *(u32x2*)p = v; }
  }
}

// ---------------------------------------------------------------------------------------------------------
// max over aligned groups of LANES adjacent lanes, using DPP only (no LDS traffic):
// quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror give xor-1/2/4/8 partners once the
// smaller groups are already uniform; 32 and 64 lanes fall back to ds_swizzle / readlane-style shuffles.
// `lanes` is wave-uniform (a kernel argument), so these are scalar branches.
__device__ __forceinline__ float group_max(float m, int lanes) {
  const int LANES = lanes;
  if (LANES >= 2) m = fmaxf(m, u2f(__builtin_amdgcn_update_dpp(0, f2u(m), 0xB1, 0xF, 0xF, false)));   // quad_perm 1,0,3,2
  if (LANES >= 4) m = fmaxf(m, u2f(__builtin_amdgcn_update_dpp(0, f2u(m), 0x4E, 0xF, 0xF, false)));   // quad_perm 2,3,0,1
  if (LANES >= 8) m = fmaxf(m, u2f(__builtin_amdgcn_update_dpp(0, f2u(m), 0x141, 0xF, 0xF, false)));  // row_half_mirror
  if (LANES >= 16) m = fmaxf(m, u2f(__builtin_amdgcn_update_dpp(0, f2u(m), 0x140, 0xF, 0xF, false))); // row_mirror
  if (LANES >= 32) m = fmaxf(m, __shfl_xor(m, 16));
  if (LANES >= 64) m = fmaxf(m, __shfl_xor(m, 32));
  return m;
}

// same reduction on unsigned integers (abs-value bit patterns order like the magnitudes they encode, and a
// NaN pattern is the largest of all, which reproduces torch.max's NaN propagation in get_max_entry)
__device__ __forceinline__ uint32_t group_max_u32(uint32_t m, int lanes) {
  // `lanes` is wave-uniform but only known at run time on most paths.  The four DPP stages are executed UNCONDITIONALLY and
  // each result is kept or dropped with a select on a scalar condition: 8 VALU operations and no control flow.  (As a chain
  // of `if (lanes >= k)` every vector paid up to six scalar branches, which also fence the VALU scheduling around them:
  // 6.7 % of the 64 MiB headline launch, tools/tune_bfp.)  A compile-time `lanes` folds the selects away.
  const uint32_t a = max(m, (uint32_t)__builtin_amdgcn_update_dpp(0, m, 0xB1, 0xF, 0xF, false));   // quad_perm 1,0,3,2
  m = lanes >= 2 ? a : m;
  const uint32_t b = max(m, (uint32_t)__builtin_amdgcn_update_dpp(0, m, 0x4E, 0xF, 0xF, false));   // quad_perm 2,3,0,1
  m = lanes >= 4 ? b : m;
  const uint32_t c = max(m, (uint32_t)__builtin_amdgcn_update_dpp(0, m, 0x141, 0xF, 0xF, false));  // row_half_mirror
  m = lanes >= 8 ? c : m;
  const uint32_t d = max(m, (uint32_t)__builtin_amdgcn_update_dpp(0, m, 0x140, 0xF, 0xF, false));  // row_mirror
  m = lanes >= 16 ? d : m;
  if (lanes >= 32) m = max(m, (uint32_t)__shfl_xor((int)m, 16));
  if (lanes >= 64) m = max(m, (uint32_t)__shfl_xor((int)m, 32));
  return m;
}

// ---------------------------------------------------------------------------------------------------------
// raw 16-byte input vector: load once, then (a) widen to fp32 and (b) take max|x| on the raw bit patterns.
typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));

// PACED LOAD ISSUE (round 5).  N x 8 idle issue cycles of this wave between two of its tile loads.  A wave that issues its 16 loads
// back to back fills the CU's address queue with ITS tile before the next wave gets a slot; paced, the waves of a CU interleave load
// by load.  Measured on 4096 x 4096 bf16 (tools/tune_pace, tools/tune_issue, tools/tune_bfp_pace -> profiles/r05_tune_pace.txt): a trivial op on deep flat tiles
// 128 x 16 12.7 -> 10.8 us, 256 x 16 11.7 -> 10.9, 512 x 16 11.4 -> 10.9 (the plateau the hot BFP kernel sits on: pacing does
// nothing for it); rows 8 KiB apart (lastdim_kernel, x * s) 12.4 -> 11.2 -- but ops with ~50+ VALU per vector LOSE (x / s 11.9 -> 14.1,
// INT8 per channel 12.3 -> 12.9: their arithmetic waits for data that now arrives later), and the optimum is not smooth in N.  Hence an op
// trait (`static constexpr int kLoadPace`), default 0, set only where the gain was >= 5 % over a range of N.
template <class OP, class = void> struct OpLoadPace { static constexpr int value = 0; };
template <class OP> struct OpLoadPace<OP, decltype((void)OP::kLoadPace)> { static constexpr int value = OP::kLoadPace; };
// (the op's pace for a launch with 8-byte loads -- 16-bit -> float32 --, where it differs: `static constexpr int kLoadPaceWide`)
template <class OP, class = void> struct OpLoadPaceWide { static constexpr int value = OpLoadPace<OP>::value; };
template <class OP> struct OpLoadPaceWide<OP, decltype((void)OP::kLoadPaceWide)> { static constexpr int value = OP::kLoadPaceWide; };
template <int N>
__device__ __forceinline__ void pace_issue() {
  if constexpr (N > 0) {
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (N >= 8) __builtin_amdgcn_s_sleep(N / 8);   // (64 cycles each)
#pragma unroll
    for (int q = 0; q < N % 8; q++) asm volatile("s_nop 7");
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <bool NT, typename OFF = int64_t, bool UNAL = false>
__device__ __forceinline__ u32x4 load_raw16(const void* p, OFF byte_off) {
  if (UNAL) {
    const u32x4_unal* src = (const u32x4_unal*)((const char*)p + byte_off);
    return NT ? __builtin_nontemporal_load(src) : *src;
  }
  const u32x4* src = (const u32x4*)((const char*)p + byte_off);
  return NT ? __builtin_nontemporal_load(src) : *src;
}

// GUARD: wrap fp16 conversions in opaque().  Keep it on: besides the fma_mix issue (see opaque()), hipcc 7.2 was
// observed to mis-select the half of the dword when it is free to fold these conversions (wrong elements).
template <int DT, int EPL, bool GUARD = true>
__device__ __forceinline__ void widen(const u32x4& v, float (&x)[EPL]) {
  if (DT == DMXQ_F32) {
#pragma unroll
    for (int j = 0; j < 4; j++) x[j] = u2f(v[j]);
  } else {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (DT == DMXQ_BF16) {
        x[2 * j] = u2f(v[j] << 16);
        x[2 * j + 1] = u2f(v[j] & 0xFFFF0000u);
      } else if (GUARD) {
        x[2 * j] = half_lo(v[j]);
        x[2 * j + 1] = half_hi(v[j]);
      } else {
        x[2 * j] = (float)__builtin_bit_cast(_Float16, (uint16_t)(v[j] & 0xFFFFu));
        x[2 * j + 1] = (float)__builtin_bit_cast(_Float16, (uint16_t)(v[j] >> 16));
      }
    }
  }
}

// fp32 bit pattern of max|x| over the 16-byte vector (exact: widening is monotone)
template <int DT>
__device__ __forceinline__ uint32_t absmax_bits(const u32x4& v) {
  if (DT == DMXQ_F32) {
    const uint32_t a = max(v[0] & 0x7FFFFFFFu, v[1] & 0x7FFFFFFFu);
    const uint32_t b = max(v[2] & 0x7FFFFFFFu, v[3] & 0x7FFFFFFFu);
    return max(a, b);
  }
  // packed 16-bit: clear both sign bits, v_pk_max_u16 tree, then fold the two halves
  const u16x2 a0 = __builtin_bit_cast(u16x2, v[0] & 0x7FFF7FFFu), a1 = __builtin_bit_cast(u16x2, v[1] & 0x7FFF7FFFu);
  const u16x2 a2 = __builtin_bit_cast(u16x2, v[2] & 0x7FFF7FFFu), a3 = __builtin_bit_cast(u16x2, v[3] & 0x7FFF7FFFu);
  const u16x2 m2 = __builtin_elementwise_max(__builtin_elementwise_max(a0, a1), __builtin_elementwise_max(a2, a3));
  const uint32_t w = __builtin_bit_cast(uint32_t, m2);
  const uint32_t h = max(w & 0xFFFFu, w >> 16);
  if (DT == DMXQ_BF16) return h << 16;
  return f2u((float)__builtin_bit_cast(_Float16, (uint16_t)h));
}

// ---------------------------------------------------------------------------------------------------------
// Division by a value shared by many elements (a per-tensor / per-group / per-channel scale) inside the affine
// fixed-point cast  v = clamp(round(x / scale + zp))  (numerical/cast.py:293).  hipcc expands an fp32 `/` into ~11 VALU
// operations (v_div_scale x2, v_rcp, four refinement FMAs, v_div_fmas, v_div_fixup) and the affine INT8 kernels were
// bound by them (profiles/r02_pmc_second_tier.txt: 27 VALU per element, VALU issue time ~ kernel time).
// With rs = RN(1 / d) formed ONCE per vector / channel by a true IEEE division, each quotient is
//     q0 = RN(n * rs);   r = n - d * q0  (ONE fma, exact: q0 is a faithful rounding of n / d);   q = RN(q0 + r * rs)
// which IS the IEEE quotient RN(n / d) (Markstein: correctly rounded reciprocal + exact residual => correctly rounded
// quotient) whenever d is in [2^-20, 2^20] (checked once per vector by the caller: recip_ok) and 2^-100 <= |n| <= 2^100
// (then q0 is normal and r, a multiple of 2^(e_n - 47), is exactly representable); tests/test_gpu_round2.py checks that
// claim bit for bit on 10^7 operand pairs.  Outside that range of n the CLAMPED INTEGER result of the cast cannot depend
// on the last bit of the quotient:
//   |n| < 2^-100: |q| < 2^-80 either way, so q + zp rounds to zp (or to a zero with the sign of n when zp = 0);
//   |n| > 2^100 : |q| > 2^80 either way and the result is the clamp limit;  n = +-0: r is formed as -(d q0 - n), which
//   keeps q = q0 = +-0 with the sign of n;  n = +-Inf / NaN: q0 is already the IEEE result (the correction would turn
//   Inf into NaN) and is selected by one v_cmp_class.
// 5 VALU operations per element, no branch.  NOT a general division: callers that return the quotient itself
// (SmoothQuant's x / s) keep the IEEE `/`.
struct Recip { float d, rs; };
__device__ __forceinline__ Recip make_recip(float d) { return Recip{d, 1.0f / d}; }
__host__ __device__ __forceinline__ bool recip_ok(float d) { return d >= 9.5367431640625e-07f && d <= 1048576.0f; }  // [2^-20, 2^20]
__device__ __forceinline__ float div_for_clamped_int(float n, const Recip& c) {
  const float q0 = n * c.rs;
  const float r = -__builtin_fmaf(c.d, q0, -n);
  const float q = __builtin_fmaf(r, c.rs, q0);
  return __builtin_amdgcn_classf(q0, 0x001 | 0x002 | 0x004 | 0x200) ? q0 : q;  // sNaN, qNaN, -inf, +inf
}

// The QUOTIENT itself through the reciprocal (round 5: SmoothQuant's x / s, whose result IS the quotient -- until now an IEEE division per
// element, ~13 VALU cycles' worth with its quarter-rate v_rcp).  Same three operations as above; by Markstein's theorem the result is
// RN(n / d) whenever rs = RN(1 / d), q0 is normal and the residual is exact, i.e. for d in [2^-20, 2^20] (recip_ok) and n = +-0 or
// 2^-100 <= |n| <= 2^100 (q0 >= 2^-120; the residual, a multiple of 2^(e_n - 47), is representable; no overflow).  Anything else --
// tiny, huge, Inf, NaN -- is the caller's cold IEEE redo: div_by_recip_ok is three compares (abs modifiers are free), so a quotient
// costs 6 operations instead of ~13.  Checked bit for bit against the IEEE division (tests/test_gpu_round5.py: random, every exponent,
// all-ones mantissas, zeros of both signs).
__device__ __forceinline__ float div_by_recip(float n, float d, float rs) {
  const float q0 = n * rs;
  const float r = -__builtin_fmaf(d, q0, -n);   // (this form keeps the sign of a zero numerator: common.hpp div_for_clamped_int)
  return __builtin_fmaf(r, rs, q0);
}
__device__ __forceinline__ bool div_by_recip_ok(float n) {
  const float a = __builtin_fabsf(n);
  return (a >= 0x1p-100f && a <= 0x1p100f) || n == 0.0f;
}

// The affine integer cast  (clamp(rne'(x / d + z)) - z) * d  of N elements that share ONE scale (a vector inside a quantisation
// group), two elements per instruction through the packed fp32 pipe: v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32 carry the
// reciprocal quotient of div_for_clamped_int, the + z, the (a + 0.5f) - 0.5f rounding helper (sim_helper.cpp:14-21 in its fp32
// form) and the dequantisation; v_rndne and the clamp (one v_med3) stay per element: ~7.5 VALU per element instead of ~13
// (the kernel's time followed its VALU count, profiles/r02_pmc_second_tier.txt).  Same values bit for bit as the per-element form
// for every FINITE quotient; a lane whose q0 is Inf / NaN (where the correction step and v_med3 are wrong: NaN must stay NaN)
// reports `special` and the caller redoes its vector per element behind one cold wave-uniform branch.
typedef float f32x2 __attribute__((ext_vector_type(2)));
// ZP0: the caller knows z == +0.0f (wave-uniform, decided per tile): the + z and - z steps are skipped -- exact, see FixedOp::tile_variant.
// Special quotients (round 5): a q0 that is Inf / NaN makes its corrected quotient q a NaN (d (+-Inf) - n is +-Inf or NaN, and
// (-+Inf) rs + (+-Inf) = NaN), so ONE class test on the SUM of the vector's quotients replaces a test per element (N / 2 packed adds + 2
// operations instead of N compares); a finite sum that overflows only sends the lane through the exact redo needlessly.
template <int N, bool ZP0 = false>
__device__ __forceinline__ bool affine_int_pairs(const float (&x)[N], float (&y)[N], float d, float rs, float z, float t_min, float t_max) {
  static_assert(N % 2 == 0, "pairs");
  f32x2 acc = {0.0f, 0.0f};
#pragma unroll
  for (int k = 0; k < N; k += 2) {
    const f32x2 n2 = {x[k], x[k + 1]};
    const f32x2 q0 = n2 * rs;
    const f32x2 t = __builtin_elementwise_fma((f32x2){d, d}, q0, -n2);    // -(r): r = n - d q0, exact
    const f32x2 q = __builtin_elementwise_fma(-t, (f32x2){rs, rs}, q0);
    acc = k == 0 ? q : acc + q;
    f32x2 u = q;
    if (!ZP0) u = u + z;
    u = (u + 0.5f) - 0.5f;
    f32x2 v;
    v.x = __builtin_amdgcn_fmed3f(__builtin_rintf(u.x), t_min, t_max);
    v.y = __builtin_amdgcn_fmed3f(__builtin_rintf(u.y), t_min, t_max);
    if (!ZP0) v = v - z;
    const f32x2 o = v * d;
    y[k] = o.x;
    y[k + 1] = o.y;
  }
  return __builtin_amdgcn_classf(acc.x + acc.y, 0x001 | 0x002 | 0x004 | 0x200);  // sNaN, qNaN, -inf, +inf
}

// a value the program knows to be the same in every lane, moved to scalar registers (what follows it -- index arithmetic,
// table loads -- then runs on the scalar unit / as s_load)
__device__ __forceinline__ int64_t uniform_i64(int64_t v) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

// A read of memory that no thread of the running kernel writes (parameter tables produced by EARLIER launches), through the
// constant address space: with a wave-uniform address the compiler emits an s_load whatever stores it has seen -- through a plain
// global pointer it must prove that nothing in the kernel may have clobbered the location, gives up on most of our kernels and
// issues a vector load, whose first consumer then waits an L2 round trip.
template <class T>
__device__ __forceinline__ T load_uniform_const(const T* p) {
  return *(const __attribute__((address_space(4))) T*)(uintptr_t)p;
}

// Tile geometry of the flat-stream kernel for a tensor of n_vec lane-vectors: workgroup-contiguous tiles of
// threads x unroll vectors.  Measured on 4096-column bf16 tensors of 256 .. 16384 rows with tools/tune_bfp
// (profiles/r02_tune_bfp_sweep.txt), 256 CUs:
//   * up to 32 MiB of input the best shape keeps the WHOLE tensor in flight in one round of <= 2 workgroups per CU
//     (every CU reads its share, computes, writes it: the phases stay in lockstep and HBM sees pure read bursts
//     followed by pure write bursts): 512x1 (<= 4 MiB), 128x2 (<= 14 MiB: many small workgroups ramp fastest),
//     512x4 (<= 16 MiB), 128x8 (<= 20 MiB), 512x16 (<= 32 MiB: the 4096x4096 bf16 headline tensor, 256 tiles);
//   * beyond that several rounds per CU are needed anyway, and small 512x2 tiles (4 resident workgroups per CU that
//     desynchronise, so reads of one overlap writes of another) win: 77-79 % of 8 TB/s vs 68-71 % for 512x16.
//   * round 4 (max_depth: 18 for the symmetric 16-bit -> same-16-bit single-rounding build, whose 17 / 18-vector tiles fit 256 VGPRs;
//     16 for the other symmetric same-dtype builds): between 20 and 36 MiB ONE round of <= 256 workgroups with exactly the depth
//     that takes -- see below.  Needed the partial last
//     tile on the same schedule as a full one first (bfp_rows_tile_partial): run vector by vector it made every such plan erratic.
// Compute units of the device the plans are made for (round 6).  Every size class below was measured on the 256 CUs of an
// MI355X in SPX mode and is, at bottom, a statement about WORK PER CU ("the whole tensor in one round of <= 2 workgroups per CU"): the
// classes are therefore kept in units of 1/256 of the chip and scaled by the device's own count -- a partition (CPX: 32 CUs) or another
// part of the family gets plans with the same per-CU shape instead of plans that silently assume 256 -- and a count other than 256 says
// so once on stderr, because nothing was MEASURED there.  No device (the GPU-less build container: dmxq_bfp_qdq_describe, argument
// checks): 256.  DMXQ_PLAN_CUS overrides (tests).
inline int plan_cus() {
  static const int cus = [] {
    if (const char* e = getenv("DMXQ_PLAN_CUS")) {
      const int v = atoi(e);
      if (v > 0) return v;
    }
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
      (void)hipGetLastError();
      return 256;
    }
    if (n != 256)
      fprintf(stderr, "[dmxq] this device reports %d compute units; the launch plans were tuned on 256 (MI355X, SPX) and are scaled to %d/256 of "
                      "their size classes -- unmeasured geometry, results are unaffected\n", n, n);
    return n;
  }();
  return cus;
}
// a size of THIS device's launch (lane-vectors, bytes) in the units the size classes were measured in: the same work per CU on 256 CUs
inline int64_t plan_norm(int64_t size_here) {
  const int cus = plan_cus();
  return cus == 256 ? size_here : (size_here * 256 + cus - 1) / cus;
}

struct RowsPlan { int id, threads, unroll; int64_t tiles; };
inline RowsPlan rows_plan(int64_t n_vec_device, bool allow_big, int max_depth = 0 /* exact-depth one-round plans up to this many vectors per lane */) {
  auto mk = [&](int id, int t, int u) { return RowsPlan{id, t, u, (n_vec_device + (int64_t)t * u - 1) / ((int64_t)t * u)}; };
  // the classes below read in the units they were measured in: lane-vectors on 256 CUs
  const int64_t n_vec = plan_norm(n_vec_device);
  if (n_vec <= ((int64_t)1 << 18)) return mk(0, 512, 1);
  // (round 4: 128 x 2 up to 14 MiB instead of 12 -- 1600 / 1792 rows of 4096 bf16: 58.7 / 61.2 % against 50.3 / 56.1 % for 512 x 4; at 1920
  //  rows 512 x 4 leads 61.0 to 55.9; exactly fitting one-round depths of 5 .. 8 vectors lose to both: tools/tune_bfp -DTUNE_SMALLFIT)
  if (n_vec <= ((int64_t)7 << 17)) return mk(1, 128, 2);
  if (allow_big) {  // (the any-rounding build would spill at many vectors per lane: it goes straight to 512x2)
    if (n_vec <= ((int64_t)1 << 20)) return mk(2, 512, 4);
    // (round 3, tools/tune_bfp TUNE_SET=wg -> profiles/r03_tune_bfp_mid.txt: 512x6 measured 10.4 us on 3072 x 4096 bf16 against 9.35 for
    //  512x16 and ~9.5 on 2560 rows against 7.98 for 128x8: 60 % -> 67-69 % of the roofline at 20-24 MiB)
    if (n_vec <= ((int64_t)5 << 18)) return mk(3, 128, 8);
    if (max_depth >= 11) {
      // the depth that fills ONE round of one workgroup per CU (256 on the MI355X): U = ceil(n_vec / (256 x 512)) for 11 .. 20 (19 / 20: the compact kernel,
      // bfp_rows_compact_kernel); id = 100 + U.  Against 512 x 16
      // below 32 MiB: +4 .. +9 points at 2688 - 3584 rows of 4096 bf16 (192 - 224 of 256 CUs busy there), +1 at 3840; against 512 x 2
      // above: +5 .. +10 at 4100 - 4608 rows (profiles/r04_tune_bfp_oneround.txt, r04_mid_shapes.txt).  19+ vectors spill in
      // bfp_rows_tile (whole-tile store groups; 20 vectors in groups of 10: 62-64 % at 4700 - 5120 rows against 66-68 % for 512 x 2);
      // the COMPACT kernel runs 19 vectors at 74-75 % (4700 / 4864 rows, +8) and 20 in groups of 10 at 71-72 % (5120 rows, +4);
      // 22 / 24 vectors gain 1-2.6 points: not built (profiles/r04_tune_bfp_compact.txt).
      const int64_t units = (n_vec + ((int64_t)1 << 17) - 1) >> 17;
      if (units >= 11 && units <= max_depth) return mk(100 + (int)units, 512, (int)units);
    }
    if (n_vec <= ((int64_t)1 << 21)) return mk(4, 512, 16);
  }
  return mk(5, 512, 2);
}

inline int grid_for(int64_t work_items_of_one_thread) {
  int64_t b = (work_items_of_one_thread + kThreads - 1) / kThreads;
  if (b < 1) b = 1;
  if (b > kMaxBlocks) b = kMaxBlocks;
  return (int)b;
}

// n / d for n < 2^31 and a launch-invariant d < 2^31 as one multiply-high and a shift (magic number from the host):
// M = ceil(2^(31+l) / d), l = ceil(log2 d); q = umulhi(n, M) >> (l - 1); d = 1 is the identity.  (A 32-bit division by
// a runtime value is ~28 VALU instructions, and the walker below needs three per lane-vector.)
struct FastDiv31 {
  uint32_t M, sh, d;
  __device__ __forceinline__ uint32_t div(uint32_t n) const { return d == 1u ? n : __umulhi(n, M) >> sh; }
};
inline FastDiv31 make_fastdiv31(int64_t d64) {
  const uint32_t d = d64 < 1 ? 1u : (d64 > 0x7FFFFFFF ? 0x7FFFFFFFu : (uint32_t)d64);
  if (d == 1u) return FastDiv31{0u, 0u, 1u};
  int l = 0;
  while (((uint64_t)1 << l) < d) l++;
  return FastDiv31{(uint32_t)((((uint64_t)1 << (31 + l)) + d - 1) / d), (uint32_t)(l - 1), d};
}

// n / d for ANY n < 2^32 and a launch-invariant d >= 1 without a branch (Granlund-Montgomery: t = umulhi(M, n); q = (t + ((n - t) >> s1)) >> s2):
// FastDiv31's `d == 1 ? n : ...` is a branch around a kernel-argument load when it runs first thing in a kernel -- two dependent scalar
// loads ahead of the workgroup's first data load (the reductions' workgroup-id decode: +0.3 us on a 8 us kernel).
struct FastDivU32 {
  uint32_t M, s1, s2, d;
  __device__ __forceinline__ uint32_t div(uint32_t n) const { const uint32_t t = __umulhi(M, n); return (t + ((n - t) >> s1)) >> s2; }
};
inline FastDivU32 make_fastdiv_u32(int64_t d64) {
  const uint32_t d = d64 < 1 ? 1u : (d64 > 0xFFFFFFFFll ? 0xFFFFFFFFu : (uint32_t)d64);
  int l = 0;
  while (((uint64_t)1 << l) < d) l++;
  const uint64_t m = ((((uint64_t)1 << l) - d) << 32) / d + 1;   // < 2^32
  return FastDivU32{(uint32_t)m, (uint32_t)(l < 1 ? l : 1), (uint32_t)(l > 1 ? l - 1 : 0), d};
}

// Launch-error scoping.  HIP keeps ONE "last error" per host thread, shared with every other HIP user of the thread
// (torch, RCCL): hipGetLastError() alone would report -- and clear -- somebody else's earlier failure as ours.  So the
// first launch of a call records whether an error was ALREADY pending (DMXQ_LAUNCH -> launch_pre), and launch_status()
// only claims (and clears) an error that appeared since then; a foreign pending error is left untouched for its owner, and
// the call then returns DMXQ_ERR_PENDING (its launches cannot be verified) -- never DMXQ_OK.
struct LaunchTls { bool armed = false; hipError_t pre = hipSuccess; };
inline LaunchTls& launch_tls() { static thread_local LaunchTls t; return t; }
inline void launch_pre() {
  LaunchTls& t = launch_tls();
  if (!t.armed) { t.pre = hipPeekAtLastError(); t.armed = true; }
}
inline int launch_status() {
  LaunchTls& t = launch_tls();
  const bool foreign = t.armed && t.pre != hipSuccess;
  t.armed = false;
  const hipError_t post = hipPeekAtLastError();
  if (foreign) {
    // An error of another HIP user was pending before our first launch.  If the code changed, the new one is ours: claim it.
    // If not, our launches cannot be verified (ours may have failed with the same code, or theirs masks it), and the caller's
    // output may be uninitialised: never report OK.  The foreign error itself is left in place for its owner.
    if (post != t.pre) { (void)hipGetLastError(); return DMXQ_ERR_LAUNCH; }
    return DMXQ_ERR_PENDING;
  }
  if (post == hipSuccess) return DMXQ_OK;
  (void)hipGetLastError();
  return DMXQ_ERR_LAUNCH;
}
#define DMXQ_LAUNCH(...) do { ::dmxq::launch_pre(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

inline bool valid_dtype(int d) { return d == DMXQ_F32 || d == DMXQ_F16 || d == DMXQ_BF16; }
inline bool valid_rounding(int r) { return r >= 0 && r <= 3; }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// dtype / rounding dispatch: calls f(std::integral_constant...) style via macros kept local to each .hip file
// ---------------------------------------------------------------------------------------------------------------------------
// Range-only casts of bf16 words (elementwise.hip float_range_bf16_kernel, dmxq_binary_cast, dmxq_rope_cast): a FloatingPoint
// format that keeps at least bf16's 7 mantissa bits only clamps the magnitude (Inf / NaN included: the reference reserves no codes
// for them) and flushes what lies below its smallest normal value -- done on both halves of a dword at once.
typedef int16_t i16x2 __attribute__((ext_vector_type(2)));
struct Range16 { uint32_t limit2, minb2; };  // both halves of a dword: clamp limit, and (smallest-normal threshold - 1) as int16 (bit patterns of the 16-bit dtype)
__device__ __forceinline__ uint32_t range16_word(uint32_t w, const Range16& r) {
  const uint32_t aw = w & 0x7FFF7FFFu;
  const u16x2 cl = __builtin_elementwise_min(__builtin_bit_cast(u16x2, aw), __builtin_bit_cast(u16x2, r.limit2));
  const uint32_t res = (w & 0x80008000u) | __builtin_bit_cast(uint32_t, cl);
  // per half: keep iff |x| bits >= minb  <=>  (minb - 1) - |x| < 0 as int16 (both below 2^15: no overflow) -> arithmetic shift = mask.
  // v_and, v_pk_min_u16, v_and, v_pk_sub_i16, v_pk_ashrrev_i16, v_bitop3: 6 operations per dword (the saturating-subtract form
  // compiled to two 16-bit compares, two selects and a permute: 10)
  const i16x2 t = __builtin_bit_cast(i16x2, r.minb2) - __builtin_bit_cast(i16x2, aw);
  const i16x2 keep = t >> (i16x2){15, 15};
  return res & __builtin_bit_cast(uint32_t, keep);
}
// false: this format is not a range-only cast of `dtype` values.  The reference's largest exponent is 2^(exp_bits-1) whatever the
// bias (quant_cpu.cpp:359-402 through numerical/format.py:166-167), so its "FP16" saturates at 2^16 (2 - 2^-10), not at 65504.
inline bool range16_of(const dmxq_float_fmt* f, int dtype, Range16* r) {
  if (dtype != DMXQ_BF16 && dtype != DMXQ_F16) return false;
  if (!f || f->exp_bits == 0) { *r = Range16{0xFFFFFFFFu, 0xFFFFFFFFu}; return true; }  // SAME: identity (no limit, threshold 0)
  const int dman = dtype == DMXQ_BF16 ? 7 : 10;
  if (f->exp_bits < 1 || f->exp_bits > 8 || f->man_bits < dman || f->man_bits > 22 || !f->flush_subnormal) return false;
  const int min_exp = -(f->exp_bias - 1), max_u = 1 << (f->exp_bits - 1);  // unbiased exponents of the smallest / largest binade
  uint32_t limit, minb;
  if (dtype == DMXQ_BF16) {
    if (min_exp < -126 || min_exp > 127) return false;
    const int max_e = max_u + 127;
    // |x| bits above the largest binade are replaced by bf16(max_val): (max_e + 1) << 7 when man > 7 (max_val = 2^max_u (2 - 2^-man)
    // rounds up), (max_e << 7) | 0x7F when man == 7; max_e = 255: no limit at all
    limit = max_e >= 255 ? 0xFFFFu : (f->man_bits > 7 ? (uint32_t)(max_e + 1) << 7 : ((uint32_t)max_e << 7) | 0x7Fu);
    minb = (uint32_t)(127 + min_exp) << 7;   // bits of 2^min_exp: below it the value is flushed to +0
  } else {
    if (min_exp > 15) return false;
    // fp16 words: max_val >= 65520 rounds to Inf in the tensor dtype (finite fp16 values are never clamped, NaN becomes +-Inf)
    if (max_u >= 128) limit = 0xFFFFu;  // exp_bits = 8: nothing exceeds the largest binade, Inf and NaN pass through
    else if (max_u >= 16 || (max_u == 15 && f->man_bits > 10)) limit = 0x7C00u;
    else limit = f->man_bits > 10 ? (uint32_t)(max_u + 1 + 15) << 10 : ((uint32_t)(max_u + 15) << 10) | 0x3FFu;
    if (max_u + 15 < 1) return false;
    minb = min_exp >= -14 ? (uint32_t)(min_exp + 15) << 10 : (min_exp >= -24 ? 1u << (min_exp + 24) : 1u);  // normal / subnormal bits of 2^min_exp
  }
  const uint32_t m1 = (minb - 1u) & 0xFFFFu;
  *r = Range16{limit | (limit << 16), m1 | (m1 << 16)};
  return true;
}

}  // namespace dmxq
