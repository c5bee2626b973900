// csrc/lastdim.hpp — the kernel for per-channel parameters along the CONTIGUOUS dim, shared by the affine integer casts and SmoothQuant's
// scaling (elementwise.hip) and the dense SmoothQuant-scale + BFP weight path (hypernet.hip).
//
// OP interface:
//   template <int N> struct RawParams;  RawParams<N> fetch_params<N>(int64_t c0) const     the table reads of N consecutive channels
//   template <int N> struct ChanParams; ChanParams<N> make_params<N>(const RawParams<N>&)  their arithmetic (reciprocals, conversions)
//   void apply_chan(const float (&x)[N], const ChanParams<N>&, float (&y)[N], int64_t e0)  N elements of one row; called by EVERY lane of
//                                                                                           a wave (cross-lane ops are allowed)
// Optional (round 5), `static constexpr bool kDeferredRedo = true`: the op's straight-line form may not cover every input; then
//   bool apply_chan_flag(x, p, y, e0)   computes y with the straight-line form and returns true when this lane's row needs
//   void apply_chan_exact(x, p, y, e0)  instead -- which the kernel runs in ONE cold loop just before the store burst, from a fresh load
//                                       (every lane of the wave takes part again: cross-lane ops stay legal), the flagged lanes storing.
// In place, behind a wave-uniform branch per row, the 16 rows of a lane could not overlap (a branch fences the scheduling of what
// surrounds it) and the cold blocks sat between the hot ones (stream.hpp has the same finding for its tiles).
#pragma once
#include <type_traits>

#include "common.hpp"

namespace dmxq {

template <class OP, class = void> struct OpDeferredRedo { static constexpr bool value = false; };
template <class OP> struct OpDeferredRedo<OP, std::void_t<decltype(OP::kDeferredRedo)>> { static constexpr bool value = OP::kDeferredRedo; };

// Per-channel parameters along the CONTIGUOUS dim (activations per hidden channel, SmoothQuant's input / weight
// scaling): a lane keeps the parameters of its EPL channels in registers and handles RPI rows of them, instead of
// re-reading the scale / zero-point tables (12 B per element, 3x the data itself) for every lane-vector.
// Layout: lpr = min(cv, THREADS) lanes per row (cv = C / EPL vectors per row), rpp = THREADS / lpr rows side by side in a
// workgroup, column strips of THREADS vectors (grid.y) when rows are longer; a workgroup owns rpp * RPI rows, ONE pass.
// Schedule (round 3, tools/tune_lastdim -> profiles/r03_tune_lastdim.txt): table reads, then all RPI data loads, and only then the
// parameter arithmetic (8 reciprocals, int64 -> float) -- rounds 1-2 finished the parameters first, i.e. a full L2 round trip plus
// ~350 VALU before the first HBM request of a workgroup (48 % of roofline for per-channel INT8).  Every row is converted into
// registers, then the stores go out as one burst (the schedule of bfp_rows.hpp).  One pass per workgroup and 16 rows in flight
// measured best throughout: looping workgroups (2-16 passes) lost 10-50 %.
// IVB = input bytes per lane-vector: 16, or 8 for a 16-bit -> float32 launch of the per-element ops (launch_lastdim below): the lane then
// owns 4 elements, loads 8 bytes and STORES 16 contiguous ones, so that every store instruction of a wave covers whole lines -- with 16-byte
// loads a lane's 8 float32 results are two 16-byte stores 32 bytes apart, half-written lines per instruction (x / s, bf16 -> float32 on
// 4096 x 4096: 30.3 us; the same finding as bfp_rows.hpp's IVB).
template <int IVB>
__device__ __forceinline__ u32x4 lastdim_load(const void* p, int64_t off) {
  if constexpr (IVB == 16) return load_raw16<true>(p, off);
  const u32x2 t = __builtin_nontemporal_load((const u32x2*)((const char*)p + off));
  return u32x4{t.x, t.y, 0u, 0u};
}
template <int DTI, int DTO, class OP, int THREADS, int RPI, int IVB = 16, int PACE = 0>
__global__ __launch_bounds__(THREADS) void lastdim_kernel(const void* __restrict__ in, void* __restrict__ out, int64_t rows,
                                                         int64_t C, int cv, const FastDivU32 lpr_div, int rpp, OP op) {
  constexpr int EPL = IVB / Elem<DTI>::bytes, OVB = EPL * Elem<DTO>::bytes;
  const int t = threadIdx.x;
  const int lpr = (int)lpr_div.d;
  const int sub = (int)lpr_div.div((uint32_t)t), sl = t - sub * lpr;
  const int cb = blockIdx.y * lpr + sl;
  const bool active = sub < rpp && cb < cv;
  const int cbc = cb < cv ? cb : cv - 1;
  const int subc = sub < rpp ? sub : rpp - 1;
  const int64_t r0 = (int64_t)blockIdx.x * rpp * RPI;
  const auto pr = op.template fetch_params<EPL>((int64_t)cbc * EPL);
  // Addresses: ONE scalar base per workgroup (its first row), a scalar step per row group, one 32-bit lane offset (lastdim_plan keeps a
  // workgroup's rpp * RPI rows under 4 GiB in either dtype).  Whole workgroups (all but the last along the rows): no per-load address
  // arithmetic between the loads; the last one clamps its rows (unconditional loads) and predicates its stores.
  const bool whole = r0 + (int64_t)rpp * RPI <= rows;  // wave-uniform
  const uint32_t lane_v = (uint32_t)subc * (uint32_t)cv + (uint32_t)cbc;
  const uint32_t step_v = (uint32_t)rpp * (uint32_t)cv;   // vectors from one of a lane's rows to its next
  const char* src = (const char*)in + r0 * cv * IVB;
  u32x4 raw[RPI];
  if (whole) {
#pragma unroll
    for (int j = 0; j < RPI; j++) {
      const char* rowp = src + (uint64_t)((uint32_t)j * step_v) * IVB;
      if constexpr (IVB == 16) raw[j] = load_raw16<true, uint32_t>(rowp, lane_v * 16u);
      else raw[j] = lastdim_load<IVB>(rowp, (int64_t)(lane_v * (uint32_t)IVB));
      if (j + 1 < RPI) pace_issue<PACE>();
    }
  } else {
    const uint32_t last = (uint32_t)(rows - 1 - r0);  // (the last row of the tensor, counted from this workgroup's first)
#pragma unroll
    for (int j = 0; j < RPI; j++) {
      const uint32_t lr = (uint32_t)j * (uint32_t)rpp + (uint32_t)subc;
      raw[j] = lastdim_load<IVB>(src, (int64_t)(((lr < last ? lr : last) * (uint32_t)cv + (uint32_t)cbc) * (uint32_t)IVB));
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  const auto p = op.template make_params<EPL>(pr);
  __builtin_amdgcn_sched_barrier(0);
#ifdef DMXQ_EXP_LD_WAITALL
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  OutVec<DTO, EPL> o[RPI];
  uint32_t redo = 0u;   // (kDeferredRedo) bit j: this lane's row j needs the op's exact form
  auto widen_row = [&](const u32x4& rw, float (&x)[EPL]) __attribute__((always_inline)) {
    if constexpr (IVB == 16) {
      widen<DTI, EPL>(rw, x);
    } else {
      float xw[16 / Elem<DTI>::bytes];
      widen<DTI, 16 / Elem<DTI>::bytes>(rw, xw);
#pragma unroll
      for (int k = 0; k < EPL; k++) x[k] = xw[k];
    }
  };
#pragma unroll
  for (int j = 0; j < RPI; j++) {
    const int64_t r = r0 + (int64_t)j * rpp + sub;
    float x[EPL], y[EPL];
    widen_row(raw[j], x);
    if constexpr (OpDeferredRedo<OP>::value) redo |= op.apply_chan_flag(x, p, y, r * C + (int64_t)cb * EPL) ? 1u << j : 0u;
    else op.apply_chan(x, p, y, r * C + (int64_t)cb * EPL);
    o[j] = pack_vec<DTO, EPL>(y);
    __builtin_amdgcn_sched_barrier(0);
  }
  if constexpr (OpDeferredRedo<OP>::value) {
    // BEFORE the store burst: nothing of this workgroup's rows has been stored yet, so the fresh load sees the ORIGINAL elements when
    // `out` aliases `in` too (include/dmxq.h allows exact aliasing; round 5 ran this loop after the burst and re-read its own results).
    // The exact result replaces o[j] through a select chain over compile-time indices (o[] stays in registers).
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(redo != 0u) != 0ull, 0)) {
#pragma unroll 1
      for (int j = 0; j < RPI; j++) {
        if (__builtin_amdgcn_ballot_w64(((redo >> j) & 1u) != 0u) == 0ull) continue;   // (wave-uniform: every lane of the wave redoes the row)
        const int64_t r = r0 + (int64_t)j * rpp + sub, rc = r < rows ? r : rows - 1;
        const u32x4 rw = lastdim_load<IVB>(in, (rc * cv + cbc) * IVB);
        float x[EPL], y[EPL];
        widen_row(rw, x);
        op.apply_chan_exact(x, p, y, r * C + (int64_t)cb * EPL);
        const OutVec<DTO, EPL> o1 = pack_vec<DTO, EPL>(y);
        const bool mine = ((redo >> j) & 1u) != 0u;
#pragma unroll
        for (int k = 0; k < RPI; k++)
          if (k == j && mine) o[k] = o1;
      }
    }
  }
  char* dst = (char*)out + r0 * cv * OVB;
  if (whole) {
    if (active) {
#pragma unroll
      for (int j = 0; j < RPI; j++)
        store_out<DTO, EPL, true>(dst + (uint64_t)((uint32_t)j * step_v) * OVB + lane_v * (uint32_t)OVB, o[j]);
    }
  } else {
#pragma unroll
    for (int j = 0; j < RPI; j++) {
      const uint32_t lr = (uint32_t)j * (uint32_t)rpp + (uint32_t)sub;
      if (active && r0 + lr < rows) store_out<DTO, EPL, true>(dst + (lr * (uint32_t)cv + (uint32_t)cb) * (uint32_t)OVB, o[j]);
    }
  }
}

// Geometry of one lastdim launch: THREADS lanes per workgroup, RPI rows per lane, gx workgroups along the rows
struct LastdimPlan { int threads, rpi; int64_t gx; int cv, lpr, rpp, strips; };
static inline bool lastdim_plan(int dti, int dto, int64_t rows, int64_t C, LastdimPlan* pl, int ivb = 16) {
  const int epl = ivb / (dti == DMXQ_F32 ? 4 : 2);
  if (C % epl != 0 || C / epl > 0x7FFFFFFF || rows < 1) return false;
  const int cv = (int)(C / epl);
  const int threads = kThreads;
  const int lpr = cv < threads ? cv : threads, rpp = threads / lpr;
  const int strips = (cv + lpr - 1) / lpr;
  if (strips > 65535 || (int64_t)rpp * 16 * cv * 32 > 0xFFFFFFFFll) return false;  // (32-bit lane offsets: a workgroup's <= 16 row groups < 4 GiB in either dtype)
  // rows per lane: as many (16, 8, 4) as still leave two workgroups per CU
  // (a widening output -- 32 bytes per lane, two half-line stores -- keeps 8: with 16 the partial lines of a row group no longer merge
  //  before they leave the L2, 95 MB written for 67 MB of output and 34.0 instead of 30.3 us, bf16 -> float32 x / s on 4096 x 4096)
  int rpi = (dto == DMXQ_F32 && dti != DMXQ_F32 && ivb == 16) ? 8 : 16;
  while (rpi > 4 && ((rows + (int64_t)rpp * rpi - 1) / ((int64_t)rpp * rpi)) * strips < 512) rpi >>= 1;
  const int64_t gx = (rows + (int64_t)rpp * rpi - 1) / ((int64_t)rpp * rpi);
  if (gx > 0x7FFFFFFF) return false;
  *pl = LastdimPlan{threads, rpi, gx, cv, lpr, rpp, strips};
  return true;
}

// DMXQ_ERR_UNSUPPORTED: not applicable (caller keeps its other kernel)
template <int DTI, int DTO, class OP, int IVB = 16>
static int launch_lastdim_typed(const void* in, void* out, int64_t rows, int64_t C, const OP& op, hipStream_t s) {
  LastdimPlan pl;
  if (!aligned16(in) || !aligned16(out) || !lastdim_plan(DTI, DTO, rows, C, &pl, IVB)) return DMXQ_ERR_UNSUPPORTED;
#ifdef DMXQ_EXP_LD_PACE
  constexpr int kPace = DMXQ_EXP_LD_PACE;
#else
  constexpr int kPace = IVB == 8 ? OpLoadPaceWide<OP>::value : OpLoadPace<OP>::value;   // common.hpp: idle issue cycles between a lane's row loads
#endif
#define DMXQ_LDK(R_)                                                                                                          \
  DMXQ_LAUNCH((lastdim_kernel<DTI, DTO, OP, kThreads, R_, IVB, kPace>), dim3((unsigned)pl.gx, (unsigned)pl.strips), dim3(kThreads), 0, s, in, out, rows, C, \
              pl.cv, make_fastdiv_u32(pl.lpr), pl.rpp, op)
  if (pl.rpi == 16) DMXQ_LDK(16);
  else if (pl.rpi == 8) DMXQ_LDK(8);
  else DMXQ_LDK(4);
#undef DMXQ_LDK
  return launch_status();
}

template <class OP>
static int launch_lastdim(const void* in, void* out, int dti, int dto, int64_t rows, int64_t C, const OP& op, hipStream_t s) {
#define DMXQ_LD(I_, O_) \
  if (dti == I_ && dto == O_) return launch_lastdim_typed<I_, O_, OP>(in, out, rows, C, op, s);
  DMXQ_LD(DMXQ_BF16, DMXQ_BF16)
  DMXQ_LD(DMXQ_F16, DMXQ_F16)
  DMXQ_LD(DMXQ_F32, DMXQ_F32)
#undef DMXQ_LD
  // 16-bit -> float32: 8-byte loads, 16-byte stores (see lastdim_kernel)
  if (dti == DMXQ_BF16 && dto == DMXQ_F32) return launch_lastdim_typed<DMXQ_BF16, DMXQ_F32, OP, 8>(in, out, rows, C, op, s);
  if (dti == DMXQ_F16 && dto == DMXQ_F32) return launch_lastdim_typed<DMXQ_F16, DMXQ_F32, OP, 8>(in, out, rows, C, op, s);
  return DMXQ_ERR_UNSUPPORTED;
}

}  // namespace dmxq
