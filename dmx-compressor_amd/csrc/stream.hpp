// csrc/stream.hpp — the streaming skeleton shared by the per-element kernels (float / fixed / scale / gelu).
//
// Same memory schedule as the BFP rows kernel (bfp_rows.hpp, tuned in tools/tune_bfp): a workgroup owns a
// CONTIGUOUS tile of THREADS*UNROLL 16-byte input vectors; all UNROLL non-temporal loads of a full tile are
// issued back to back, then every vector is converted into registers, then all stores go out as one burst.
// OP::apply_vec(x[EPL], y[EPL], e0) maps EPL consecutive elements (flat index e0..) to their outputs in fp32.
#pragma once
#include <type_traits>

#include "common.hpp"

namespace dmxq {

// Optional per-vector prefetch: an OP with a nested `Prep` type and `prepare(e0)` gets it called for ALL vectors of a
// tile right after the data loads, before any arithmetic, and receives the result in apply_vec(x, y, e0, prep).  The
// per-group scale / zero-point reads of the affine ops are a dependent L2 access per vector; fetched inside the
// vector-by-vector compute loop (whose sched_barriers keep live ranges short) each of them exposed its full latency.
template <class OP, class = void>
struct OpPrep {
  using type = int;
  static __device__ __forceinline__ type get(const OP&, int64_t) { return 0; }
  template <int N>
  static __device__ __forceinline__ void apply(const OP& op, const float (&x)[N], float (&y)[N], int64_t e0, const type&) {
    op.apply_vec(x, y, e0);
  }
};
template <class OP>
struct OpPrep<OP, std::void_t<typename OP::Prep>> {
  using type = typename OP::Prep;
  static __device__ __forceinline__ type get(const OP& op, int64_t e0) { return op.prepare(e0); }
  template <int N>
  static __device__ __forceinline__ void apply(const OP& op, const float (&x)[N], float (&y)[N], int64_t e0, const type& p) {
    op.apply_vec(x, y, e0, p);
  }
};

// Optional per-TILE side data: an OP with `bool tile_prepare(int64_t e0, int64_t len, Prep&) const` is asked, with wave-uniform
// arguments, whether the elements [e0, e0 + len) all use the same Prep (then filled in from scalar loads).
template <class OP, class = void> struct OpTilePrep { static constexpr bool value = false; };
template <class OP> struct OpTilePrep<OP, std::void_t<decltype(OP::kTilePrep)>> { static constexpr bool value = OP::kTilePrep; };

// Optional straight-line tile forms: an OP with `static constexpr int kTileVariants = 3` offers, for tiles whose side data came from
// tile_prepare, `int tile_variant(Prep&) const` (wave-uniform; 0 = the general apply_vec; may complete the Prep, e.g. a reciprocal),
// `bool apply_vec_tile<V>(x, y, prep)` for V = 1, 2 -- true: this lane's vector needs `apply_vec_exact(x, y, e0, prep)` instead, which
// the kernel runs in one cold loop just before the tile's store burst (in-place safe).  One scalar branch per TILE picks the body, nothing per vector.
template <class OP, class = void> struct OpTileVariants { static constexpr int value = 1; };
template <class OP> struct OpTileVariants<OP, std::void_t<decltype(OP::kTileVariants)>> { static constexpr int value = OP::kTileVariants; };

template <class OP, class = void> struct OpWaitAll { static constexpr bool value = false; };
template <class OP> struct OpWaitAll<OP, std::void_t<decltype(OP::kWaitAll)>> { static constexpr bool value = OP::kWaitAll; };

template <int V, class OP, class PREP, int N>
__device__ __forceinline__ void tile_apply(const OP& op, const float (&x)[N], float (&y)[N], int64_t e0, const PREP& p) {
  OpPrep<OP>::apply(op, x, y, e0, p);
}
template <int V, class OP, class PREP, int N>
__device__ __forceinline__ bool tile_apply_flag(const OP& op, const float (&x)[N], float (&y)[N], const PREP& p) {
  return op.template apply_vec_tile<V>(x, y, p);
}
template <class OP, class PREP, int N>
__device__ __forceinline__ void tile_apply_exact(const OP& op, const float (&x)[N], float (&y)[N], int64_t e0, const PREP& p) {
  op.apply_vec_exact(x, y, e0, p);
}

// Optional raw-word hooks (act_cast.hip): an OP with `static constexpr bool kRawHooks = true` sees every 16-byte input vector
// before it is widened (raw_in) and every packed output vector before it is stored (raw_out) -- the range-only casts of 16-bit
// tensors act on the packed words (common.hpp range16_word), two elements per operation.
template <class OP, class = void> struct OpRawHooks { static constexpr bool value = false; };
template <class OP> struct OpRawHooks<OP, std::void_t<decltype(OP::kRawHooks)>> { static constexpr bool value = OP::kRawHooks; };

// IVB = input bytes per lane-vector: 16, or 8 for aligned 16-bit -> float32 launches -- the lane then owns 4 elements and its results
// are ONE 16-byte store, every store instruction of a wave covering whole lines (with 16-byte loads they are two 16-byte stores 32 bytes
// apart: half-written lines per instruction, ~40 % of the bandwidth; bfp_rows.hpp, lastdim.hpp).
template <int IVB, bool UNAL, bool NTL = true>
__device__ __forceinline__ u32x4 stream_load(const char* p, uint32_t off) {
  if constexpr (IVB == 16) return load_raw16<NTL, uint32_t, UNAL>(p, off);
  const u32x2 t = NTL ? __builtin_nontemporal_load((const u32x2*)(p + off)) : *(const u32x2*)(p + off);
  return u32x4{t.x, t.y, 0u, 0u};
}
template <int DTI, int EPL>
__device__ __forceinline__ void stream_widen(const u32x4& raw, float (&x)[EPL]) {
  constexpr int FULL = 16 / Elem<DTI>::bytes;
  if constexpr (EPL == FULL) {
    widen<DTI, FULL>(raw, x);
  } else {
    float xw[FULL];
    widen<DTI, FULL>(raw, xw);
#pragma unroll
    for (int k = 0; k < EPL; k++) x[k] = xw[k];
  }
}
// ONE tile (THREADS x UNROLL lane-vectors) of a flat tensor of n_vec vectors; `tile` is the tile index inside that tensor.  Shared by the
// single-tensor kernel below and the multi-tensor kernel (stream_multi_kernel: many small tensors of one op in one launch).
template <int DTI, int DTO, int UNROLL, int THREADS, class OP, bool UNAL, int IVB, int PACE, bool NTS = true, bool NTL = true>
__device__ __forceinline__ void stream_tile(const void* __restrict__ in, void* __restrict__ out, int64_t n_vec, int64_t tile, const OP& op) {
  constexpr int EPL = IVB / Elem<DTI>::bytes;
  constexpr int OVB = EPL * Elem<DTO>::bytes;
  constexpr int64_t TILE = (int64_t)THREADS * UNROLL;
  const uint32_t lane_in = threadIdx.x * (uint32_t)IVB, lane_out = threadIdx.x * (uint32_t)OVB;
  const char* src = (const char*)in + tile * (TILE * IVB);
  char* dst = (char*)out + tile * (TILE * OVB);
  const int64_t v0 = tile * TILE + threadIdx.x;
  if ((tile + 1) * TILE <= n_vec) {
    // load burst + compute + store burst of a full tile, with the per-vector side data coming from prep_of(u).  The side data
    // (scale / zero-point reads) is requested BEFORE the tile's own loads: vector memory returns in order, so behind them it
    // would only arrive after the whole tile, and the first vector's arithmetic could not start while the rest streams in.
    u32x4 raw[UNROLL];
    auto load_tile = [&]() __attribute__((always_inline)) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < UNROLL; u++) {
        raw[u] = stream_load<IVB, UNAL, NTL>(src + u * (THREADS * IVB), lane_in);
        if (u + 1 < UNROLL) pace_issue<PACE>();
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    // `variant`: std::integral_constant<int, V>; V = 0 is the op's general per-vector form, V > 0 a straight-line form the op
    // offers for tiles whose (wave-uniform) side data allows it (OpTileVariants below)
    auto finish = [&](auto prep_of, auto variant) __attribute__((always_inline)) {
      constexpr int V = decltype(variant)::value;
      OutVec<DTO, EPL> o[UNROLL];
      uint32_t redo = 0u;  // (V > 0) bit u: this lane's vector u needs the op's exact form
#ifdef DMXQ_EXP_WAITALL_ALL
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
      if constexpr (V == 0 && OpWaitAll<OP>::value) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // The whole tile's data before the first vector's arithmetic (round 5).  Left to the compiler the waits are per vector
      // (vmcnt(15), (14), ...): a wave then starts converting while the rest of its tile -- and the other workgroups' tiles -- still
      // stream in, finishes early and starts STORING while others still read.  Measured on the INT8 group cast, 4096 x 4096 bf16,
      // 128 x 16 tiles: 13.06 us per-vector waits, 11.48 us with this one wait (tools/tune_stream, -DDMXQ_EXP_*): read bursts
      // followed by write bursts, the finding of the BFP kernel's tile schedule, applies inside a tile's wait
      // pattern too.  Per op, measured (profiles/r05_tune_stream_waitall.txt): light ops gain 3-8 % (INT8 without a scale 11.06 -> 10.68 us,
      // the FLOAT16 cast of float32 tensors 12.04 -> 11.00), VALU-heavy ones on deep tiles LOSE 5-15 % (SiLU 512 x 16 11.35 -> 12.91:
      // their arithmetic no longer overlaps the tail of the loads) -- hence a trait (`static constexpr bool kWaitAll = true`), not a rule.
      if constexpr (V > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#pragma unroll
      for (int u = 0; u < UNROLL; u++) {
        float x[EPL], y[EPL];
        if constexpr (OpRawHooks<OP>::value) op.raw_in(raw[u]);
        stream_widen<DTI, EPL>(raw[u], x);
        if constexpr (V == 0) tile_apply<V>(op, x, y, (v0 + (int64_t)u * THREADS) * EPL, prep_of(u));
        else redo |= tile_apply_flag<V>(op, x, y, prep_of(u)) ? 1u << u : 0u;
        o[u] = pack_vec<DTO, EPL>(y);
        if constexpr (OpRawHooks<OP>::value) op.raw_out(o[u]);
#ifdef DMXQ_EXP_NOFENCE
        if constexpr (V == 0) __builtin_amdgcn_sched_barrier(0);
#else
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
      if constexpr (V > 0) {
        // The straight-line forms do not cover every input (Inf / NaN quotients): flagged vectors are redone BEFORE the store burst,
        // from a fresh load -- nothing of this tile has been stored yet, so the load sees the ORIGINAL elements when `out` aliases
        // `in` too (include/dmxq.h: exact aliasing is allowed; round 5 redid them after the burst and read its own results back) --
        // in ONE cold loop per tile whose result replaces o[u] through a select chain over compile-time indices (o[] stays in
        // registers).  Redone in place behind a branch per vector, the compiler laid 16 cold blocks of ~250 instructions between the
        // hot ones: the tile body no longer fitted the instruction cache (INT8 per group, zero point 0: 12.2 us where the form WITH
        // a zero point, whose cold blocks happened to be placed out of line, ran 11.6; tools/tune_stream int8g0 / int8g).
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(redo != 0u) != 0ull, 0)) {
#pragma unroll 1
          for (int u = 0; u < UNROLL; u++) {
            if ((redo >> u) & 1u) {
              const u32x4 r1 = stream_load<IVB, UNAL, NTL>(src + u * (THREADS * IVB), lane_in);
              float x[EPL], y[EPL];
              stream_widen<DTI, EPL>(r1, x);
              tile_apply_exact(op, x, y, (v0 + (int64_t)u * THREADS) * EPL, prep_of(u));
              const OutVec<DTO, EPL> o1 = pack_vec<DTO, EPL>(y);
#pragma unroll
              for (int k = 0; k < UNROLL; k++)
                if (k == u) o[k] = o1;
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < UNROLL; u++) store_out<DTO, EPL, NTS, UNAL>(dst + u * (THREADS * OVB) + lane_out, o[u]);
    };
    if constexpr (OpTilePrep<OP>::value) {
      // the whole tile shares one set of side data (one quantisation group): fetched once, from a wave-uniform address,
      // instead of an index computation and two dependent loads per 16-byte vector
      typename OP::Prep tp;
      if (op.tile_prepare(tile * (TILE * EPL), TILE * EPL, tp)) {  // wave-uniform
        load_tile();
        auto pf = [&](int) -> const typename OP::Prep& { return tp; };
        if constexpr (OpTileVariants<OP>::value == 3) {
          // (the scalar table reads are waited for HERE, behind the tile's loads)
          const int v = op.tile_variant(tp);  // (also completes tp)
          if (v == 2) finish(pf, std::integral_constant<int, 2>{});
          else if (v == 1) finish(pf, std::integral_constant<int, 1>{});
          else finish(pf, std::integral_constant<int, 0>{});
        } else {
          finish(pf, std::integral_constant<int, 0>{});
        }
        return;
      }
    }
    typename OpPrep<OP>::type prep[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) prep[u] = OpPrep<OP>::get(op, (v0 + (int64_t)u * THREADS) * EPL);
    load_tile();
    finish([&](int u) -> const typename OpPrep<OP>::type& { return prep[u]; }, std::integral_constant<int, 0>{});
  } else {
    // the last, partial tile, vector by vector.  (The hot BFP kernel runs its partial tile on the full tile's schedule, in a function
    // of its own -- bfp_rows_tile_partial.  The same here measured a DISASTER: with a non-inlined call in the kernel the full tiles
    // of float_qdq went 12.0 -> 21.9 us, fixed_qdq 11.0 -> 16.2, silu 11.2 -> 12.2 on 4096 x 4096 bf16; and these kernels keep >= 2
    // workgroups per CU, so the serial tail of ONE workgroup costs them at most ~5 % at a few sizes, not the 40 % it cost a
    // one-workgroup-per-CU plan: tools/probe_partial.py.)
    for (int u = 0; u < UNROLL; u++) {
      const int64_t vi = v0 + (int64_t)u * THREADS;
      if (vi < n_vec) {
        u32x4 raw = stream_load<IVB, UNAL, NTL>(src + u * (THREADS * IVB), lane_in);
        float x[EPL], y[EPL];
        if constexpr (OpRawHooks<OP>::value) op.raw_in(raw);
        stream_widen<DTI, EPL>(raw, x);
        op.apply_vec(x, y, vi * EPL);
        OutVec<DTO, EPL> o1 = pack_vec<DTO, EPL>(y);
        if constexpr (OpRawHooks<OP>::value) op.raw_out(o1);
        store_out<DTO, EPL, NTS, UNAL>(dst + u * (THREADS * OVB) + lane_out, o1);
      }
    }
  }
}

template <int DTI, int DTO, int UNROLL, int THREADS, class OP, bool UNAL = false, int IVB = 16, int PACE = 0>
__global__ __launch_bounds__(THREADS) void stream_kernel(const void* __restrict__ in, void* __restrict__ out,
                                                        int64_t n, OP op) {
  static_assert(IVB == 16 || (IVB == 8 && !UNAL && !OpRawHooks<OP>::value), "8-byte input vectors: aligned tensors, no raw-word hooks");
  constexpr int EPL = IVB / Elem<DTI>::bytes;
  constexpr int64_t TILE = (int64_t)THREADS * UNROLL;
  const int64_t n_vec = n / EPL;
  const int64_t n_tiles = (n_vec + TILE - 1) / TILE;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) stream_tile<DTI, DTO, UNROLL, THREADS, OP, UNAL, IVB, PACE>(in, out, n_vec, tile, op);
  // scalar tail: the n % EPL elements after the last whole vector
  if (blockIdx.x == 0 && threadIdx.x < (unsigned)(n - n_vec * EPL)) {
    const int64_t e = n_vec * EPL + threadIdx.x;
    float x1[1] = {load1<DTI>(in, e)}, y1[1];
    op.apply_one(x1[0], y1[0], e);
    store1<DTO>(out, e, y1[0]);
  }
}

// Geometry: workgroup-contiguous tiles of THREADS x UNROLL 16-byte vectors, one tile per workgroup.  Round 3 measured every op of the
// library over 15 geometries and 13 tensor sizes with the kernels themselves (tools/tune_stream -> profiles/r03_tune_stream.txt).
// What holds for ALL of them, light or VALU-heavy, is the rule by tensor size -- keep the whole tensor in flight in ONE round of
// resident workgroups while that is possible, with as few loads per lane as that takes:
//     <=  2 MiB of input   256 x 1
//     <= 12 MiB            256 x 2      (silu on 8 MiB: 4.4 us; the 256 x 8 tiles of round 2 took 7.0)
//     <= 20 MiB            256 x 4
//     <= 32 MiB            the op's own geometry, kTileThreads x kTileUnroll (default 256 x 4 for kHeavy ops, 256 x 16 otherwise): here
//                          the ops differ by how much arithmetic sits between the load burst and the store burst -- e.g. INT8 without a
//                          scale 64 x 16, with a per-group scale 128 x 16, minifloat cast 64 x 8, silu 256 x 8, the fused GELU module
//                          512 x 4, the fused SiLU module 512 x 16 (4096 x 4096 bf16: 11.1 / 12.4 / 11.9 / 11.4 / 13.7 / 11.7 us)
//     beyond               256 x 2: several rounds per CU are needed anyway, and small tiles whose workgroups drift apart (reads of one
//                          overlapping writes of another) measured best for every op (77 % of 8 TB/s at 112 MiB)
template <class OP, class = void> struct OpHeavy { static constexpr bool value = false; };
template <class OP> struct OpHeavy<OP, decltype((void)OP::kHeavy)> { static constexpr bool value = OP::kHeavy; };
template <class OP, class = void> struct OpTileUnroll { static constexpr int value = OpHeavy<OP>::value ? 4 : 16; };
template <class OP> struct OpTileUnroll<OP, decltype((void)OP::kTileUnroll)> { static constexpr int value = OP::kTileUnroll; };
template <class OP, class = void> struct OpTileThreads { static constexpr int value = 256; };
template <class OP> struct OpTileThreads<OP, decltype((void)OP::kTileThreads)> { static constexpr int value = OP::kTileThreads; };

// An OP whose arithmetic is tied to the 16-byte lane-vector (blocks = adjacent lanes: blockfmt.hip) says `kFixedVector = true`
template <class OP, class = void> struct OpFixedVector { static constexpr bool value = false; };
template <class OP> struct OpFixedVector<OP, std::void_t<decltype(OP::kFixedVector)>> { static constexpr bool value = OP::kFixedVector; };

template <int DTI, int DTO, class OP>
static int launch_stream(const void* in, void* out, int64_t n, const OP& op, hipStream_t s) {
  constexpr bool WIDE = Elem<DTO>::bytes > Elem<DTI>::bytes;
  if constexpr (WIDE && !OpRawHooks<OP>::value && !OpFixedVector<OP>::value) {
    // 16-bit -> float32 on aligned tensors: lane-vectors of 4 elements (8 bytes in, 16 out), the geometry rule applied to the OUTPUT bytes
    if (aligned16(in) && aligned16(out)) {
      const int64_t nv = n / 4;
      const int64_t nvc = plan_norm(nv);   // (size classes: measured on 256 CUs, common.hpp plan_cus)
#define DMXQ_STREAM8(T_, U_)                                                                                           \
  do {                                                                                                                 \
    int64_t tiles = (nv + (int64_t)(T_) * (U_) - 1) / ((int64_t)(T_) * (U_));                                          \
    if (tiles < 1) tiles = 1;                                                                                          \
    if (tiles > (1 << 20)) tiles = 1 << 20;                                                                            \
    DMXQ_LAUNCH((stream_kernel<DTI, DTO, U_, T_, OP, false, 8>), dim3((unsigned)tiles), dim3(T_), 0, s, in, out, n, op); \
  } while (0)
      if (nvc <= ((int64_t)1 << 17)) DMXQ_STREAM8(256, 1);
      else if (nvc <= ((int64_t)3 << 18)) DMXQ_STREAM8(256, 2);
      else if (nvc <= ((int64_t)1 << 21)) DMXQ_STREAM8(256, 4);
      else DMXQ_STREAM8(256, 2);
#undef DMXQ_STREAM8
      return launch_status();
    }
  }
  constexpr int EPL = 16 / Elem<DTI>::bytes;
  constexpr int UB = WIDE ? 8 : 16;  // (a widening output doubles the registers a vector holds)
  constexpr int TT = OpTileThreads<OP>::value, TU = OpTileUnroll<OP>::value < UB ? OpTileUnroll<OP>::value : UB;
  static_assert(TU == 16 || TU == 8 || TU == 4 || TU == 2, "kTileUnroll");
  static_assert(TT == 64 || TT == 128 || TT == 256 || TT == 512, "kTileThreads");
  const int64_t n_vec = n / EPL;
  if (!aligned16(in) || !aligned16(out)) {  // views that start mid-allocation: same schedule on unaligned 16-byte accesses
    int64_t tiles = (n_vec + (int64_t)256 * 4 - 1) / ((int64_t)256 * 4);
    if (tiles < 1) tiles = 1;
    if (tiles > (1 << 20)) tiles = 1 << 20;
    DMXQ_LAUNCH((stream_kernel<DTI, DTO, 4, 256, OP, true>), dim3((unsigned)tiles), dim3(256), 0, s, in, out, n, op);
    return launch_status();
  }
#define DMXQ_STREAM(T_, U_)                                                                                       \
  do {                                                                                                            \
    int64_t tiles = (n_vec + (int64_t)(T_) * (U_) - 1) / ((int64_t)(T_) * (U_));                                  \
    if (tiles < 1) tiles = 1;                                                                                     \
    if (tiles > (1 << 20)) tiles = 1 << 20;                                                                       \
    DMXQ_LAUNCH((stream_kernel<DTI, DTO, U_, T_, OP>), dim3((unsigned)tiles), dim3(T_), 0, s, in, out, n, op); \
  } while (0)
  const int64_t nvc = plan_norm(n_vec);   // (size classes: measured on 256 CUs, common.hpp plan_cus)
  if (nvc <= ((int64_t)1 << 17)) DMXQ_STREAM(256, 1);
  else if (nvc <= ((int64_t)3 << 18)) DMXQ_STREAM(256, 2);
  else if (nvc <= ((int64_t)5 << 18)) DMXQ_STREAM(256, 4);
  else if (nvc <= ((int64_t)1 << 21)) {
    // the op's own geometry, with the op's own pace between a wave's loads (common.hpp OpLoadPace; 0 for most ops)
#ifdef DMXQ_EXP_STREAM_PACE
    constexpr int kPace = DMXQ_EXP_STREAM_PACE;
#else
    constexpr int kPace = OpLoadPace<OP>::value;
#endif
    int64_t tiles = (n_vec + (int64_t)TT * TU - 1) / ((int64_t)TT * TU);
    if (tiles < 1) tiles = 1;
    DMXQ_LAUNCH((stream_kernel<DTI, DTO, TU, TT, OP, false, 16, kPace>), dim3((unsigned)tiles), dim3(TT), 0, s, in, out, n, op);
  }
  else DMXQ_STREAM(256, 2);
#undef DMXQ_STREAM
  return launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------------
// MANY tensors of one op in ONE launch (round 5): the tile body above, the tensor found from the workgroup index.  For sets of sibling
// parameters that are launch-bound one by one -- the INT8 casts of a decoder layer's weights, the bias casts of its six Linear
// modules (768 .. 3072 elements each: ~2.3 us of launch for nothing to stream).  Every tensor has its OWN op instance (its scale /
// zero-point tables, its channel map); whole 16-byte vectors, 16-byte aligned, < 2^31 vectors: the callers give every other tensor a
// launch of its own, so the result is always what one call per tensor gives.  Geometry: launch_stream's size rule applied to the
// TOTAL of the set.  (The hand-written predecessor of this kernel, fixed_multi_kernel, ran 256 x 4 tiles with a table read and a
// division per vector: 13.8 us for the 56 MB of an opt-125m layer's six float32 weights, 51 % of the roofline.)
// (A/B switch: -DDMXQ_EXP_MULTI_NTS=1 builds the small-set launches with non-temporal loads and stores too)
#ifdef DMXQ_EXP_MULTI_NTS
constexpr bool kMultiSmallNT = DMXQ_EXP_MULTI_NTS != 0;
#else
constexpr bool kMultiSmallNT = false;
#endif
template <class OP> struct StreamMultiDesc { const void* in; void* out; int64_t n_vec, tile0; OP op; };
template <class OP> struct StreamMultiArgs {
  static constexpr int kMax = (int)(3600 / sizeof(StreamMultiDesc<OP>)) < 32 ? (int)(3600 / sizeof(StreamMultiDesc<OP>)) : 32;  // (a 4 KiB argument block)
  int n;
  StreamMultiDesc<OP> d[kMax];
};
// WHICH tensor a workgroup belongs to comes from ten scalar arguments -- e[i] = the first tile of tensor i + 1 (0xFFFFFFFF beyond the
// last) -- that arrive in SGPRs with the wave (kernel-argument preloading covers the first 44 bytes of SCALAR arguments): no memory
// access for sets of up to 11 tensors.  Scanned from the argument block instead, a workgroup made two DEPENDENT scalar-memory round
// trips (the tile table, then its descriptor) before its first data load: ~1 us on a 10 us launch.  Larger sets finish the scan in the
// argument block.
constexpr int kStreamMultiPre = 10;
// NTS: non-temporal loads AND stores.  Off for sets of up to 32 MiB: their results are consumed at once (a layer's quantised weights by its
// GEMMs), their inputs are read again by the next forward, and both fit the Infinity Cache -- opt-125m layer, one hipGraph of the forward:
// 188.5 -> 185.1 us with plain stores, -> 182.0 us with plain loads too.
template <int DTI, int DTO, int UNROLL, int THREADS, class OP, bool NTS>
__global__ __launch_bounds__(THREADS) void stream_multi_kernel(uint32_t e0, uint32_t e1, uint32_t e2, uint32_t e3, uint32_t e4, uint32_t e5,
                                                              uint32_t e6, uint32_t e7, uint32_t e8, uint32_t e9, int n,
                                                              const StreamMultiArgs<OP> a) {
  const uint32_t gt = blockIdx.x;
  int k = (e0 <= gt) + (e1 <= gt) + (e2 <= gt) + (e3 <= gt) + (e4 <= gt) + (e5 <= gt) + (e6 <= gt) + (e7 <= gt) + (e8 <= gt) + (e9 <= gt);
  if (k == kStreamMultiPre) {
    for (int i = kStreamMultiPre + 1; i < n; i++) k = ((uint32_t)a.d[i].tile0 <= gt) ? i : k;
  }
  const StreamMultiDesc<OP>& d = a.d[k];
  stream_tile<DTI, DTO, UNROLL, THREADS, OP, false, 16, 0, NTS, NTS>(d.in, d.out, d.n_vec, (int64_t)gt - d.tile0, d.op);
}

// TWO ops in one launch (round 5): the tensors of OPA first, then those of OPB -- a layer's INT8 weight casts and its float bias casts, which
// were two launches.  Small sets only (at most 21 tensors, fewer than 65,536 tiles): the 20 first-tile boundaries travel as 16-bit halves of
// ten preloaded scalar arguments, the eleventh is the number of OPA tensors -- no memory access before the descriptor fetch.
template <class OPA, class OPB> struct StreamMulti2Args {
  static constexpr int kMaxA = 12, kMaxB = 12, kMaxTensors = 21;
  int nA, nB;
  StreamMultiDesc<OPA> a[kMaxA];
  StreamMultiDesc<OPB> b[kMaxB];
};
__device__ __forceinline__ int multi2_index(uint32_t gt, uint32_t e) { return ((e & 0xFFFFu) <= gt) + ((e >> 16) <= gt); }
// (OPB's tiles are at most as deep as ITS own geometry allows: a FloatOp body at 16 vectors per lane spills, and a kernel with scratch
//  pays for it on every wave -- the first build of this kernel ran the opt-125m layer 80 us slower)
template <int DTI, int DTO, int UNROLL, int UNROLLB, int THREADS, class OPA, class OPB, bool NTS>
__global__ __launch_bounds__(THREADS) void stream_multi2_kernel(uint32_t e0, uint32_t e1, uint32_t e2, uint32_t e3, uint32_t e4, uint32_t e5,
                                                               uint32_t e6, uint32_t e7, uint32_t e8, uint32_t e9, int nA,
                                                               const StreamMulti2Args<OPA, OPB> a) {
  const uint32_t gt = blockIdx.x;
  const int k = multi2_index(gt, e0) + multi2_index(gt, e1) + multi2_index(gt, e2) + multi2_index(gt, e3) + multi2_index(gt, e4) +
                multi2_index(gt, e5) + multi2_index(gt, e6) + multi2_index(gt, e7) + multi2_index(gt, e8) + multi2_index(gt, e9);
  if (k < nA) {
    const StreamMultiDesc<OPA>& d = a.a[k];
    stream_tile<DTI, DTO, UNROLL, THREADS, OPA, false, 16, 0, NTS, NTS>(d.in, d.out, d.n_vec, (int64_t)gt - d.tile0, d.op);
  } else {
    const StreamMultiDesc<OPB>& d = a.b[k - nA];
    stream_tile<DTI, DTO, UNROLLB, THREADS, OPB, false, 16, 0, NTS, NTS>(d.in, d.out, d.n_vec, (int64_t)gt - d.tile0, d.op);
  }
}
// DMXQ_ERR_UNSUPPORTED: the set does not fit this form (the caller launches the two ops separately)
template <int DTI, int DTO, class OPA, class OPB>
static int launch_stream_multi2(StreamMulti2Args<OPA, OPB>& a, hipStream_t s) {
  using A = StreamMulti2Args<OPA, OPB>;
  if (a.nA < 1 || a.nB < 1 || a.nA > A::kMaxA || a.nB > A::kMaxB || a.nA + a.nB > A::kMaxTensors) return DMXQ_ERR_UNSUPPORTED;
  constexpr int TT = OpTileThreads<OPA>::value, TU = OpTileUnroll<OPA>::value;   // (same-size dtype pairs only: no widening outputs)
  int64_t total = 0;
  for (int i = 0; i < a.nA; i++) total += a.a[i].n_vec;
  for (int i = 0; i < a.nB; i++) total += a.b[i].n_vec;
  if (total > ((int64_t)1 << 21)) return DMXQ_ERR_UNSUPPORTED;   // (beyond 32 MiB: two launches cost nothing there)
  const int64_t totalc = plan_norm(total);
#define DMXQ_STREAM_MULTI2(T_, U_)                                                                                    \
  do {                                                                                                                \
    int64_t tiles = 0;                                                                                                \
    uint32_t first[A::kMaxTensors + 1];                                                                               \
    int n = 0;                                                                                                        \
    for (int i = 0; i < a.nA; i++) { a.a[i].tile0 = tiles; first[n++] = (uint32_t)tiles; tiles += (a.a[i].n_vec + (int64_t)(T_) * (U_) - 1) / ((int64_t)(T_) * (U_)); } \
    constexpr int UB = (U_) < 4 ? (U_) : 4;   /* (OPB: shallow tiles, see the kernel) */                                   \
    for (int i = 0; i < a.nB; i++) { a.b[i].tile0 = tiles; first[n++] = (uint32_t)tiles; tiles += (a.b[i].n_vec + (int64_t)(T_) * UB - 1) / ((int64_t)(T_) * UB); } \
    if (tiles > 0xFFFF) return DMXQ_ERR_UNSUPPORTED;                                                                  \
    uint32_t e[10];                                                                                                   \
    for (int i = 0; i < 10; i++) {                                                                                    \
      const uint32_t lo = 2 * i + 1 < n ? first[2 * i + 1] : 0xFFFFu, hi = 2 * i + 2 < n ? first[2 * i + 2] : 0xFFFFu;  \
      e[i] = lo | (hi << 16);                                                                                         \
    }                                                                                                                 \
    DMXQ_LAUNCH((stream_multi2_kernel<DTI, DTO, U_, UB, T_, OPA, OPB, kMultiSmallNT>), dim3((unsigned)tiles), dim3(T_), 0, s, e[0], e[1], e[2], e[3], \
                e[4], e[5], e[6], e[7], e[8], e[9], a.nA, a);                                                         \
  } while (0)
  if (totalc <= ((int64_t)1 << 17)) {
    DMXQ_STREAM_MULTI2(256, 1);
  } else if constexpr (Elem<DTI>::bytes == 2) {
    DMXQ_STREAM_MULTI2(256, 4);   // (16-bit: the two bodies together spill at 128 x 16 -- 3.4 KiB of scratch per lane; build.py NO_SCRATCH guards the rest)
  } else {
    if (totalc <= ((int64_t)5 << 18)) DMXQ_STREAM_MULTI2(256, 4);
    else DMXQ_STREAM_MULTI2(TT, TU);
  }
#undef DMXQ_STREAM_MULTI2
  return launch_status();
}

// Host side of one launch: a.d[i].{in, out, n_vec, op} filled for i < a.n (n_vec > 0, whole vectors, aligned); fills tile0 and launches.
template <int DTI, int DTO, class OP>
static int launch_stream_multi(StreamMultiArgs<OP>& a, hipStream_t s) {
  if (a.n < 1) return DMXQ_OK;
  constexpr bool WIDE = Elem<DTO>::bytes > Elem<DTI>::bytes;
  constexpr int UB = WIDE ? 8 : 16;
  constexpr int TT = OpTileThreads<OP>::value, TU = OpTileUnroll<OP>::value < UB ? OpTileUnroll<OP>::value : UB;
  int64_t total = 0;
  for (int i = 0; i < a.n; i++) total += a.d[i].n_vec;
#define DMXQ_STREAM_MULTI(T_, U_, N_)                                                                                   \
  do {                                                                                                                \
    int64_t tiles = 0;                                                                                                \
    for (int i = 0; i < a.n; i++) { a.d[i].tile0 = tiles; tiles += (a.d[i].n_vec + (int64_t)(T_) * (U_) - 1) / ((int64_t)(T_) * (U_)); } \
    if (tiles >= ((int64_t)1 << 31)) return DMXQ_ERR_UNSUPPORTED;                                                     \
    uint32_t e[kStreamMultiPre];                                                                                      \
    for (int i = 0; i < kStreamMultiPre; i++) e[i] = i + 1 < a.n ? (uint32_t)a.d[i + 1].tile0 : 0xFFFFFFFFu;           \
    DMXQ_LAUNCH((stream_multi_kernel<DTI, DTO, U_, T_, OP, N_>), dim3((unsigned)tiles), dim3(T_), 0, s, e[0], e[1], e[2], e[3], e[4], e[5], \
                e[6], e[7], e[8], e[9], a.n, a);                                                                     \
  } while (0)
  const int64_t totalc = plan_norm(total);
  if (totalc <= ((int64_t)1 << 17)) DMXQ_STREAM_MULTI(256, 1, kMultiSmallNT);
  else if (totalc <= ((int64_t)5 << 18)) DMXQ_STREAM_MULTI(256, 4, kMultiSmallNT);
  else if (totalc <= ((int64_t)1 << 21)) DMXQ_STREAM_MULTI(TT, TU, kMultiSmallNT);
  else DMXQ_STREAM_MULTI(256, 2, true);
#undef DMXQ_STREAM_MULTI
  return launch_status();
}

template <class OP>
static int dispatch_stream(const void* in, void* out, int dti, int dto, int64_t n, const OP& op, hipStream_t s) {
#define DMXQ_DT(I_, O_) \
  if (dti == I_ && dto == O_) return launch_stream<I_, O_, OP>(in, out, n, op, s);
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

// ---------------------------------------------------------------------------------------------------------
// Channel / group walker for tensors viewed as [outer, C, inner]: group(e) = ((e / inner) % C) / group_size.
// One division per VECTOR (32-bit when the tensor is small enough), then incremental carries per element.
struct ChannelMap {
  int64_t C, inner, group_size;
  int small;  // 1: n < 2^31, 32-bit index arithmetic
  FastDiv31 f_inner, f_C, f_gs;
  // Whole groups of equal length (C % group_size == 0, or a single group): the tensor is a sequence of RUNS of run = group_size * inner
  // contiguous elements, run k belonging to group k % G.  run_align = the largest power of two dividing run (0: not applicable):
  // a power-of-two tile of at most run_align elements starting at a multiple of its length lies inside ONE run, and its group is
  // two scalar multiply-highs away from the tile index -- no vector instruction ahead of the tile's first load.
  uint32_t run_align, G;
  FastDiv31 f_run, f_G;
};

struct ChanIter {
  int64_t c, i, g, r;
  __device__ __forceinline__ void start(const ChannelMap& m, int64_t e) {
    if (m.small) {
      const uint32_t q = m.f_inner.div((uint32_t)e);
      i = (uint32_t)e - q * (uint32_t)m.inner;
      const uint32_t cc = q - m.f_C.div(q) * (uint32_t)m.C;
      const uint32_t gg = m.f_gs.div(cc);
      c = cc; g = gg;
      r = cc - gg * (uint32_t)m.group_size;
    } else {
      const int64_t q = e / m.inner;
      i = e - q * m.inner;
      c = q % m.C;
      g = c / m.group_size;
      r = c - g * m.group_size;
    }
  }
  // advance to the next element; returns true when the group index changed
  __device__ __forceinline__ bool next(const ChannelMap& m) {
    if (++i < m.inner) return false;
    i = 0;
    if (++c == m.C) {
      c = 0; r = 0;
      const bool ch = g != 0;
      g = 0;
      return ch;
    }
    if (++r == m.group_size) { r = 0; ++g; return true; }
    return false;
  }
};

}  // namespace dmxq
