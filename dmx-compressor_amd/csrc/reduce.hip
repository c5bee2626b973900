// csrc/reduce.hip — calibration-side reductions for gfx950:
//   * per-group min/max over slabs of channels   (numerical/cast.py:179-226 + numerical/observer.py:173-193)
//   * (min,max) -> (scale, zero_point)            (numerical/observer.py:59-115)
//   * per-channel max|x|                          (numerical/smoothquant.py:285-299)
//   * SmoothQuant scale                           (numerical/smoothquant.py:301-321)
//   * torch.histc for the HistogramObserver       (numerical/observer.py:453-510)
// The reference builds one observer nn.Module per group in a Python loop and runs two ATen reductions per
// group; here one launch covers all groups.  Partial results are combined with integer atomics on the float
// bit patterns (order-preserving for non-NaN values), so the outputs are exact and order-independent.
#include <dlfcn.h>
#include <math.h>
#include <stdlib.h>

#include <mutex>
#include <unordered_map>
#include <vector>

#include "common.hpp"
#include "gate_registry.hpp"

namespace dmxq {

// ---------------------------------------------------------------------------------------------------------------------------
// THE INIT GATE (round 4; made placement-independent in round 5): the atomics of a reduction need their destination initialised
// first -- until round 4 by a fill launch in front of the kernel, ~1.6 us of the ~9 (profiles/r03_ops_roofline_table.txt).  A ticket
// (last workgroup reduces partials) costs MORE than the fill on this chip (a RETURNING device-scope atomic per workgroup at the END of
// a one-round kernel: profiles/r04_tune_reduce_tickets.txt).  The gate needs neither: ONE wave stores the identities, makes them
// visible at agent scope (release fence) and publishes the launch's EPOCH in a flag word (release store); thread 0 of every workgroup
// reads that word once its own data has arrived -- the latency hides behind the workgroup's arithmetic -- and polls it only while it
// does not yet hold the epoch, before the workgroup issues its (non-returning) atomics (acquire fence).
//   * WHO initialises is decided by a claim, not by position: `flag[1]` is an election word, and whoever swaps the epoch into it first
//     (a returning exchange) does the job.  A gated launch carries ONE EXTRA workgroup (linear id 0: the grids are one-dimensional,
//     the data workgroups are ids 1 ..) that has no data of its own: its first wave volunteers at once -- uncontended -- so on an
//     idle chip the flag is up ~1.5-2 us into the kernel, before the first data arrives anywhere, and no data wave is delayed by the
//     claim's round trip.  (A data wave as the volunteer -- wave 0 of workgroup (0, 0) -- requests its own rows ~2 us late, behind
//     everybody's: that workgroup then finishes last and the kernel is as slow as with the fill launch, 7.9 -> 9.3 us measured.)
//     A workgroup that has polled kGateTakeover times without seeing the epoch stops assuming that the volunteer is running and
//     claims the job itself: it IS running, so the identities get written whatever order the dispatcher chose and whatever else
//     occupies the chip (MI355X_MICROARCH.md, correctness boundaries: dispatch order is undefined; round 4 had every workgroup wait
//     for (0, 0) unconditionally).  A late volunteer finds the word claimed and exits.
//   * a flag slot belongs to ONE stream (launches of a stream are ordered, so a slot never serves two running kernels: no launch can
//     overwrite the epoch another one still waits for), keyed by (device of the stream, hipStreamGetId where the runtime has it --
//     ids are never reused, a destroyed stream's handle may be --, else the handle); epochs count up per slot and skip 0, the value
//     of a fresh slot;
//   * no gate -- the fill launch as before -- while the stream is being captured (a replayed graph would present the SAME epoch
//     again, already in the flag), on hipStreamPerThread (one handle, many streams), past kGateSlots streams, for more outputs than
//     one wave initialises quickly, on the scalar kernels, and when DMXQ_NO_INIT_GATE is set in the environment.
// on == 0: the destination is initialised already (a fill launch in front, or the accumulate form); 2: test hook, (0, 0) does not
// volunteer (every launch goes through the takeover path: dmxq_internal_gate_mode)
// ORDERING.  What must hold: (1) the identity stores are performed at agent scope before the epoch is; (2) a workgroup's atomics are
// issued after it has read the epoch.  Both sides touch the coherence point directly -- the identities and the epoch are agent-scope
// (sc1, write-through) stores, the epoch is read with agent-scope loads, the contributions are atomic read-modify-writes, which execute
// at the L2 / memory side and never on a cached copy -- so NO cache maintenance is needed for correctness, only ORDER:
//   publish:  identity stores; s_waitcnt vmcnt(0) (every one acknowledged at agent scope; a compiler barrier too); then the epoch store;
//   observe:  epoch load; s_waitcnt vmcnt(0) (the value is there, the branch on it resolved; compiler barrier); then the atomics --
//             a wave issues in program order and nothing after the wait reads ordinary memory that the initialiser wrote.
// The formal spelling -- `fence(release, agent)` before an `atomic_store(release)`, `fence(acquire, agent)` after the load -- compiles
// to the same waits PLUS `buffer_wbl2 sc1` (write back every dirty line of the XCD's L2) and `buffer_inv sc1` (invalidate it), the latter
// once per WORKGROUP at the end of a kernel whose other workgroups still stream through that L2.  Measured (same box, A/B of two
// builds, 4096 x 4096 bf16): per-group min / max 7.9 us with the waits, 9.2 us with the fences; per-column max |x| 7.6 vs 10.4 us --
// the fences cost more than the fill launch the gate exists to remove.  -DDMXQ_GATE_FENCES=1 builds the formal form (profiles/
// r05_gate_ab.txt); the default is the waits.
#ifndef DMXQ_GATE_FENCES
#define DMXQ_GATE_FENCES 0
#endif
__device__ __forceinline__ void gate_release_stores() {
#if DMXQ_GATE_FENCES
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#else
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}
__device__ __forceinline__ void gate_acquire_epoch() {
#if DMXQ_GATE_FENCES
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#else
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}
// (struct InitGate, kGateSlots, kGateStride, kGateMaxOut: gate_registry.hpp)
constexpr int kGateTakeover = 48;  // polls (~1 us each) before a waiting workgroup claims the initialisation

// The calling WAVE (all 64 lanes, wave-uniform control flow) tries to become the initialiser.  true: it was, and the flag is up.
template <typename F>
__device__ __forceinline__ bool gate_claim_and_init(const InitGate& g, int64_t n, F&& init) {
  const int lane = threadIdx.x & (kWave - 1);
  unsigned old = g.epoch;
  if (lane == 0) old = __hip_atomic_exchange(g.flag + 1, g.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
  if (old == g.epoch) return false;   // claimed before: that wave initialises (or has)
  for (int64_t i = lane; i < n; i += kWave) init(i);
  gate_release_stores();   // every lane's identity stores (one wave: one wait covers them all), before ...
  if (lane == 0) __hip_atomic_store(g.flag, g.epoch, DMXQ_GATE_FENCES ? __ATOMIC_RELEASE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... the epoch
  return true;
}
// (GATED is a template parameter of the kernels: the un-gated instances -- the accumulate forms, the fallbacks -- carry none of this;
// as a run-time flag it cost them 0.3 us)
// First statement of a gated kernel: true = this is the volunteer workgroup (linear id 0), the caller returns; else `wid` is the
// linear id among the DATA workgroups.  Ungated instances: wid = blockIdx.x.
template <bool GATED, typename F>
__device__ __forceinline__ bool gate_open(const InitGate& g, int64_t n, F&& init, uint32_t& wid) {
  wid = blockIdx.x;
  if (!GATED) return false;
  if (wid == 0u) {
    // the volunteer workgroup: thread 0 claims, then EVERY thread stores its share of the identities (4096 per-column maxima by one
    // wave were 64 stores per lane, ~1 us of issue: the epoch came up after the first workgroups had asked, and per-column max |x|
    // ran 9.1 us instead of 7.5) -- the workgroup has nothing else to do, so the two barriers are free
    __shared__ unsigned s_claimed;
    if (g.on == 1) {
      if (threadIdx.x == 0) s_claimed = __hip_atomic_exchange(g.flag + 1, g.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      if (s_claimed != g.epoch) {   // (block-uniform) not claimed before: ours
        for (int64_t i = threadIdx.x; i < n; i += blockDim.x) init(i);
        gate_release_stores();      // each wave's own stores
        __syncthreads();
        if (threadIdx.x == 0)
          __hip_atomic_store(g.flag, g.epoch, DMXQ_GATE_FENCES ? __ATOMIC_RELEASE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    return true;
  }
  wid -= 1u;
  return false;
}
__device__ __forceinline__ void gate_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// thread 0 only, and only while it has not seen the epoch.  `dep`: a register of the MIDDLE data load of the caller's batch -- the
// flag's address is made to depend on it, so the read is issued when half the batch has arrived: late enough to find the epoch (the
// volunteer's claim + stores + waits take ~1.5-2 us; a read issued right behind the loads, as in round 4, or at the first arrival,
// finds the old value and costs the workgroup a poll of ~0.7 us at its end: per-column max |x| 7.5 -> 8.3 us), early enough to return
// while the rest of the batch still streams in.
// (Measured in round 4: the same load by EVERY wave costs 0.4 us: 4096 agent-scope reads of one address.)
template <bool GATED>
__device__ __forceinline__ unsigned gate_peek(const InitGate& g, unsigned seen, uint32_t dep) {
  if (GATED && threadIdx.x == 0 && seen != g.epoch) {
    uint32_t off = 0u;
    asm volatile("" : "+v"(off) : "v"(dep));
    seen = __hip_atomic_load(g.flag + off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return seen;
}
// wave 0 of the workgroup (every lane calls it; the others fall through), before the atomics: the threads that issue them follow
// through a barrier or are thread 0 themselves
template <bool GATED, typename F>
__device__ __forceinline__ void gate_wait(const InitGate& g, unsigned seen, int64_t n, F&& init) {
  if (GATED && threadIdx.x < kWave) {
    unsigned s = (unsigned)__builtin_amdgcn_readfirstlane((int)seen);   // thread 0's
    int polls = 0;
    while (s != g.epoch) {
      unsigned v = 0u;
      if (threadIdx.x == 0) v = __hip_atomic_load(g.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s = (unsigned)__builtin_amdgcn_readfirstlane((int)v);
      if (s != g.epoch && ++polls == kGateTakeover && gate_claim_and_init(g, n, init)) s = g.epoch;
    }
    gate_acquire_epoch();   // the epoch read, before this workgroup's atomics
  }
}

// float atomic min/max through integer atomics: non-negative floats order like signed ints, negative floats
// order inversely like unsigned ints.
__device__ __forceinline__ void atomic_max_f32(float* addr, float v) {
  if (!(v < 0.0f) && !(f2u(v) >> 31)) atomicMax((int*)addr, (int)f2u(v));
  else atomicMin((unsigned int*)addr, f2u(v));
}
__device__ __forceinline__ void atomic_min_f32(float* addr, float v) {
  if (!(v < 0.0f) && !(f2u(v) >> 31)) atomicMin((int*)addr, (int)f2u(v));
  else atomicMax((unsigned int*)addr, f2u(v));
}

// ---------------------------------------------------------------------------------------------------------------------------
// min / max of 16-bit floats ON THE PACKED WORDS (round 4).  bf16 and fp16 are sign-magnitude: read as u16, every negative pattern
// lies above every positive one and grows with the magnitude; read as i16, positives lie above negatives and grow with the value.
// With umax = max_u16, imax = max_i16, umin = min_u16 over all words (3 packed operations per dword = 2 elements, instead of two
// widenings, two v_min and two v_max):
//   min = umax if its sign bit is set (some negative value: the one of largest magnitude), else umin (all positive: the smallest);
//   max = imax if imax >= 0 (some positive value: the largest), else umin (all negative: the one of smallest magnitude).
// NaN comes out for free: a +NaN (0x7F81.. / 0x7C01..) is the largest i16, a -NaN the largest u16, and torch.amin / amax -- what
// the reference's observer calls (numerical/observer.py:173-193) -- propagate NaN: either one makes BOTH results NaN.
// (tools/tune_reduce2.hip, profiles/r04_tune_reduce_tickets.txt: 7.44 -> 7.19 us for the reduction pass over 32 MiB.)
struct PkMinMax {
  u16x2 umax, umin;
  i16x2 imax;
  __device__ __forceinline__ void init() { umax = (u16x2){0, 0}; umin = (u16x2){0xFFFF, 0xFFFF}; imax = (i16x2){(int16_t)-32768, (int16_t)-32768}; }
  __device__ __forceinline__ void add(uint32_t w) {
    const u16x2 u = __builtin_bit_cast(u16x2, w);
    umax = __builtin_elementwise_max(umax, u);
    umin = __builtin_elementwise_min(umin, u);
    imax = __builtin_elementwise_max(imax, __builtin_bit_cast(i16x2, w));
  }
  // -> (lo, hi) as floats; `seen` = at least one element was added; a NaN anywhere gives (-NaN, +NaN), which the float atomics
  // below carry to the output (the -NaN pattern wins every unsigned max, the +NaN pattern every signed max)
  template <int DT>
  __device__ __forceinline__ void finish(float& lo, float& hi) const {
    const uint32_t um = max((uint32_t)umax.x, (uint32_t)umax.y), un = min((uint32_t)umin.x, (uint32_t)umin.y);
    const int im = max((int)imax.x, (int)imax.y);
    constexpr uint32_t kInf = DT == DMXQ_BF16 ? 0x7F80u : 0x7C00u;
    const bool nan = (um & 0x7FFFu) > kInf && (um & 0x8000u) ? true : (im > (int)kInf);
    const uint32_t lo16 = (um & 0x8000u) ? um : un, hi16 = im >= 0 ? (uint32_t)im : un;
    if (DT == DMXQ_BF16) { lo = u2f(lo16 << 16); hi = u2f(hi16 << 16); }
    else { lo = half_lo(lo16); hi = half_lo(hi16); }
    if (nan) { lo = u2f(0xFFC00000u); hi = u2f(0x7FC00000u); }
  }
};
// float32 path: fminf / fmaxf drop NaN, so a NaN is tracked on the side (integer max of the |x| patterns: above +Inf's) and
// turned into (-NaN, +NaN) at the end, as above
__device__ __forceinline__ void nan_to_both(uint32_t amax_bits, float& lo, float& hi) {
  if (amax_bits > 0x7F800000u) { lo = u2f(0xFFC00000u); hi = u2f(0x7FC00000u); }
}

__global__ void fill2_kernel(float* a, float va, float* b, float vb, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    a[i] = va;
    if (b) b[i] = vb;
  }
}

// (lo, hi) of every lane -> one pair per workgroup -> the float atomics, on ORDER-PRESERVING KEYS: key(f) = bits ^ (sign ? ~0 : 1 << 31)
// orders like the floats as an unsigned integer, puts -NaN below -Inf and +NaN above +Inf -- so a lane that saw a NaN (and carries
// (-NaN, +NaN)) wins every unsigned min / max of the combine with no special case and no branch, like it wins the atomics.
__device__ __forceinline__ uint32_t fkey(float f) { const uint32_t b = f2u(f); return b ^ ((uint32_t)((int32_t)b >> 31) | 0x80000000u); }
__device__ __forceinline__ float fkey_inv(uint32_t k) { return u2f(k ^ ((k & 0x80000000u) ? 0x80000000u : 0xFFFFFFFFu)); }
__device__ __forceinline__ void wave_minmax_keys(uint32_t& lo, uint32_t& hi) {
  // 16-lane rows by DPP (no LDS crossbar), then two xor shuffles across the rows
#define DMXQ_MM_DPP(ctrl)                                                                                   \
  lo = min(lo, (uint32_t)__builtin_amdgcn_update_dpp((int)lo, (int)lo, ctrl, 0xF, 0xF, false));            \
  hi = max(hi, (uint32_t)__builtin_amdgcn_update_dpp((int)hi, (int)hi, ctrl, 0xF, 0xF, false))
  DMXQ_MM_DPP(0xB1);   // quad_perm 1,0,3,2
  DMXQ_MM_DPP(0x4E);   // quad_perm 2,3,0,1
  DMXQ_MM_DPP(0x141);  // row_half_mirror
  DMXQ_MM_DPP(0x140);  // row_mirror
#undef DMXQ_MM_DPP
  lo = min(lo, (uint32_t)__shfl_xor((int)lo, 16)); hi = max(hi, (uint32_t)__shfl_xor((int)hi, 16));
  lo = min(lo, (uint32_t)__shfl_xor((int)lo, 32)); hi = max(hi, (uint32_t)__shfl_xor((int)hi, 32));
}
template <int T, bool GATED = false, typename F>
__device__ __forceinline__ void block_minmax_finish(float flo, float fhi, float* mn, float* mx, const InitGate& gate,
                                                    unsigned seen, int64_t n_init, F&& init) {
  uint32_t lo = fkey(flo), hi = fkey(fhi);   // (nothing seen: (+Inf, -Inf), which no combine prefers)
  wave_minmax_keys(lo, hi);
  __shared__ uint32_t s_lo[T / kWave], s_hi[T / kWave];
  const int w = threadIdx.x / kWave;
  if ((threadIdx.x & (kWave - 1)) == 0) { s_lo[w] = lo; s_hi[w] = hi; }
  __syncthreads();
  gate_wait<GATED>(gate, seen, n_init, init);   // wave 0
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 1; i < T / kWave; i++) { lo = min(lo, s_lo[i]); hi = max(hi, s_hi[i]); }
    if (lo <= hi || lo < fkey(-INFINITY)) {  // at least one element seen (or a NaN: (-NaN, +NaN) also has lo <= hi as keys)
      atomic_min_f32(mn, fkey_inv(lo));
      atomic_max_f32(mx, fkey_inv(hi));
    }
  }
}

// grid = (splits, G).  Group g owns, for every o, the contiguous run [(o*C + g*gs) * inner, +len_g*inner).
__global__ __launch_bounds__(kThreads) void group_minmax_kernel(const void* __restrict__ in, int dt, int64_t outer,
                                                               int64_t C, int64_t inner, int64_t gs, float* mn,
                                                               float* mx) {
  const int64_t g = blockIdx.y;
  const int64_t c0 = g * gs;
  const int64_t len = ((C - c0 < gs) ? (C - c0) : gs) * inner;  // run length per o
  const int64_t total = outer * len;
  float lo = INFINITY, hi = -INFINITY;
  uint32_t am = 0u;
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x; t < total; t += stride) {
    const int64_t o = t / len, r = t % len;
    const float v = load_rt(in, dt, (o * C + c0) * inner + r);
    lo = fminf(lo, v);
    hi = fmaxf(hi, v);
    am = max(am, f2u(v) & 0x7FFFFFFFu);
  }
  nan_to_both(am, lo, hi);
  block_minmax_finish<kThreads>(lo, hi, &mn[g], &mx[g], InitGate{nullptr, 0u, 0}, 0u, 0, [](int64_t) {});
}

// 8 consecutive elements, compile-time dtype: the RAW 16-byte vectors first (so that a batch of loads is issued back to
// back), widened afterwards.  (The runtime-dtype load8_rt below wraps every load in a dtype branch whose conversion
// waits for that load: one access in flight per lane.)
template <int DT>
struct Raw8 {
  u32x4 a, b;  // b only for fp32
};
template <int DT>
__device__ __forceinline__ Raw8<DT> load8_raw(const void* p, int64_t e) {
  Raw8<DT> r;
  // non-temporal: the reductions read their operand exactly once
  if (DT == DMXQ_F32) { r.a = __builtin_nontemporal_load((const u32x4*)((const float*)p + e)); r.b = __builtin_nontemporal_load((const u32x4*)((const float*)p + e + 4)); }
  else { r.a = __builtin_nontemporal_load((const u32x4*)((const uint16_t*)p + e)); r.b = r.a; }
  return r;
}
template <int DT>
__device__ __forceinline__ void widen8(const Raw8<DT>& r, float (&v)[8]) {
  if (DT == DMXQ_F32) {
#pragma unroll
    for (int j = 0; j < 4; j++) { v[j] = u2f(r.a[j]); v[4 + j] = u2f(r.b[j]); }
  } else {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (DT == DMXQ_BF16) { v[2 * j] = u2f(r.a[j] << 16); v[2 * j + 1] = u2f(r.a[j] & 0xFFFF0000u); }
      else { v[2 * j] = half_lo(r.a[j]); v[2 * j + 1] = half_hi(r.a[j]); }
    }
  }
}

// 8 consecutive elements as fp32 (16-byte accesses; the address must be 16-byte aligned)
__device__ __forceinline__ void load8_rt(const void* p, int dt, int64_t e, float (&v)[8]) {
  if (dt == DMXQ_F32) {
    const f32x4 a = *(const f32x4*)((const float*)p + e), b = *(const f32x4*)((const float*)p + e + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
    const u32x4 t = *(const u32x4*)((const uint16_t*)p + e);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (dt == DMXQ_BF16) { v[2 * j] = u2f(t[j] << 16); v[2 * j + 1] = u2f(t[j] & 0xFFFF0000u); }
      else { v[2 * j] = half_lo(t[j]); v[2 * j + 1] = half_hi(t[j]); }
    }
  }
}

// vectorised twin of group_minmax_kernel: every run of a group is a whole number of aligned 8-element vectors
constexpr int kMinmaxThreads = 1024;  // big workgroups: few contended atomics per group, 16 waves of loads in flight
template <int DT, bool GATED>
__global__ __launch_bounds__(kMinmaxThreads) void group_minmax_vec_kernel(const void* __restrict__ in,
                                                                         int64_t outer, int64_t C, int64_t inner,
                                                                         int64_t gs, float* mn, float* mx, const InitGate gate,
                                                                         const FastDivU32 nx /* splits per group */, uint32_t ng /* groups */) {
  constexpr int kThreads = kMinmaxThreads;  // shadows the namespace constant inside this kernel
  auto init = [&](int64_t i) { gate_store(&mn[i], INFINITY); gate_store(&mx[i], -INFINITY); };
  uint32_t wid;
  if (gate_open<GATED>(gate, ng, init, wid)) return;
  const int64_t g = nx.div(wid);   // (the workgroup id is wave-uniform: two scalar instructions)
  const uint32_t bx = wid - (uint32_t)g * nx.d;
  unsigned seen = ~gate.epoch;   // not the epoch: a workgroup that never peeks must wait
  const int64_t c0 = g * gs;
  const int64_t lenv = ((C - c0 < gs) ? (C - c0) : gs) * inner / 8;  // vectors per run
  const int64_t total = outer * lenv;
  float lo = INFINITY, hi = -INFINITY;
  uint32_t am = 0u;
  PkMinMax pk;
  pk.init();
  const int64_t stride = (int64_t)nx.d * kThreads;
  // (o, r) walked with carries instead of a 64-bit division per vector; 4 independent loads in flight per lane
  int64_t t = (int64_t)bx * kThreads + threadIdx.x;
  const bool any = t < total;
  int64_t o = t / lenv, r = t % lenv;
  const int64_t so = stride / lenv, sr = stride % lenv;
  while (t < total) {
    Raw8<DT> raw[4];
    int nv = 0;
    int64_t last = (o * C + c0) * inner + r * 8;  // out-of-range slots re-read the last valid vector (harmless for min/max)
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (t < total) {
        last = (o * C + c0) * inner + r * 8;
        nv = u + 1;
        t += stride; o += so; r += sr;
        if (r >= lenv) { r -= lenv; o += 1; }
      }
      raw[u] = load8_raw<DT>(in, last);
    }
    seen = gate_peek<GATED>(gate, seen, raw[2].a.x);
    (void)nv;
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (DT != DMXQ_F32) {
#pragma unroll
        for (int j = 0; j < 4; j++) pk.add(raw[u].a[j]);
      } else {
        float v[8];
        widen8<DT>(raw[u], v);
#pragma unroll
        for (int k = 0; k < 8; k++) { lo = fminf(lo, v[k]); hi = fmaxf(hi, v[k]); am = max(am, f2u(v[k]) & 0x7FFFFFFFu); }
      }
    }
  }
  if (DT != DMXQ_F32) { if (any) pk.finish<DT>(lo, hi); }
  else nan_to_both(am, lo, hi);
  block_minmax_finish<kThreads, GATED>(lo, hi, &mn[g], &mx[g], gate, seen, ng, init);
}

// outer == 1 (a weight's row slabs along dim 0, or the whole tensor as one group): the vectors of group g are ONE contiguous run, so a
// workgroup takes workgroup-contiguous tiles of 512 x 16 vectors with all 16 loads of a lane in flight -- the shape that reads fastest
// in tools/tune_reduce (profiles/r02_tune_reduce.txt: 7.3 us for 32 MiB against 8.8 us with 1024-thread workgroups and 4 loads in
// flight, grid-strided) -- and no 64-bit index arithmetic per vector.  One round for a 32 MiB tensor (256 workgroups).
constexpr int kFlatThreads = 512;
template <int DT> struct FlatUnroll { static constexpr int value = DT == DMXQ_F32 ? 8 : 16; };  // 256 bytes in flight per lane either way
template <int DT, bool GATED>
__global__ __launch_bounds__(kFlatThreads) void group_minmax_flat_kernel(const void* __restrict__ in, int64_t C, int64_t inner, int64_t gs,
                                                                        float* mn, float* mx, const InitGate gate, const FastDivU32 nx, uint32_t ng) {
  constexpr int T = kFlatThreads, U = FlatUnroll<DT>::value;
  auto init = [&](int64_t i) { gate_store(&mn[i], INFINITY); gate_store(&mx[i], -INFINITY); };
  uint32_t wid;
  if (gate_open<GATED>(gate, ng, init, wid)) return;
  const int64_t g = nx.div(wid);   // (the workgroup id is wave-uniform: two scalar instructions)
  const uint32_t bx = wid - (uint32_t)g * nx.d;
  unsigned seen = ~gate.epoch;   // not the epoch: a workgroup that never peeks must wait
  const int64_t c0 = g * gs;
  const int64_t lenv = ((C - c0 < gs) ? (C - c0) : gs) * inner / 8;  // vectors of this group
  const int64_t v0 = c0 * inner / 8;
  float lo = INFINITY, hi = -INFINITY;
  uint32_t am = 0u;
  PkMinMax pk;
  pk.init();
  const bool any = (int64_t)bx * (T * U) < lenv;   // (every lane of a workgroup that has a tile adds at least one -- clamped -- vector)
  for (int64_t b = (int64_t)bx * (T * U); b < lenv; b += (int64_t)nx.d * (T * U)) {
    Raw8<DT> raw[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t v = b + (int64_t)u * T + threadIdx.x;
      raw[u] = load8_raw<DT>(in, (v0 + (v < lenv ? v : lenv - 1)) * 8);  // clamped: a repeated vector cannot change a min / max
    }
    seen = gate_peek<GATED>(gate, seen, raw[U / 2].a.x);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (DT != DMXQ_F32) {
#pragma unroll
        for (int j = 0; j < 4; j++) pk.add(raw[u].a[j]);
      } else {
        float v[8];
        widen8<DT>(raw[u], v);
#pragma unroll
        for (int k = 0; k < 8; k++) { lo = fminf(lo, v[k]); hi = fmaxf(hi, v[k]); am = max(am, f2u(v[k]) & 0x7FFFFFFFu); }
      }
    }
  }
  if (DT != DMXQ_F32) { if (any) pk.finish<DT>(lo, hi); }
  else nan_to_both(am, lo, hi);
  block_minmax_finish<T, GATED>(lo, hi, &mn[g], &mx[g], gate, seen, ng, init);
}

// vectorised twin of channel_maxabs_kernel: a workgroup covers a strip of 64 x 8 = 512 consecutive columns of the
// (C x inner) plane; a lane owns 8 columns, the 4 waves take different rows (4 row loads in flight each), their
// partial maxima are combined through LDS and ONE wave issues the integer atomics.
constexpr int kMaxabsThreads = 1024;  // 16 waves over rows per column strip: parallelism without more atomics
constexpr int kMaxabsRows = 8;        // rows in flight per lane
template <int DT, int U, bool GATED>
__global__ __launch_bounds__(kMaxabsThreads) void channel_maxabs_vec_kernel(const void* __restrict__ in,
                                                                           int64_t outer, int64_t C, int64_t inner,
                                                                           float* out, const InitGate gate, const FastDivU32 nx /* column strips */,
                                                                           uint32_t ny /* row splits */) {
  constexpr int W = kMaxabsThreads / kWave;
  auto init = [&](int64_t i) { gate_store(&out[i], 0.0f); };
  uint32_t wid;
  if (gate_open<GATED>(gate, C, init, wid)) return;
  const uint32_t by = nx.div(wid), bx = wid - by * nx.d;
  unsigned seen = ~gate.epoch;   // not the epoch: a workgroup that never peeks must wait
  const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
  const int64_t col0 = ((int64_t)bx * kWave + lane) * 8;
  const int64_t plane = C * inner;
  const bool ok = col0 < plane;
  // a workgroup takes W * U consecutive rows per pass: wave w rows w, w + W, ...; the U row loads of a lane are all in flight
  // before the first is consumed (tools/tune_reduce.hip: W16 x U8 = 9.4 us on 4096 x 4096 bf16 against 11.6 us with 4 loads per
  // batch and 16 row splits).  Rows past the end re-read the last row: harmless for a maximum.
  // (U < 8: activations of a few thousand rows -- fewer rows per workgroup, more workgroups; see dmxq_channel_maxabs)
  // 16-bit inputs (round 4): max |x| on the PACKED words -- v_and clears both signs, v_pk_max_u16 -- 2 operations per dword instead of
  // 4, half the LDS traffic in the combine, and a NaN (the largest magnitude pattern) propagates like torch.amax's
  // (tools/tune_reduce2.hip: 7.67 -> 7.23 us for the pass over 32 MiB).
  constexpr bool PK = DT != DMXQ_F32;
  constexpr int NW = PK ? 4 : 8;   // dwords a lane keeps: 8 columns as 4 packed pairs, or 8 float patterns
  uint32_t m[NW];
#pragma unroll
  for (int k = 0; k < NW; k++) m[k] = 0u;
  if (ok) {
    for (int64_t o = (int64_t)by * (W * U) + w; o < outer; o += (int64_t)ny * (W * U)) {
      Raw8<DT> raw[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int64_t r = o + u * W < outer ? o + u * W : outer - 1;
        raw[u] = load8_raw<DT>(in, r * plane + col0);
      }
      seen = gate_peek<GATED>(gate, seen, raw[U / 2].a.x);
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (PK) {
#pragma unroll
          for (int j = 0; j < 4; j++)
            m[j] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, m[j]), __builtin_bit_cast(u16x2, raw[u].a[j] & 0x7FFF7FFFu)));
        } else {
#pragma unroll
          for (int j = 0; j < 4; j++) { m[j] = max(m[j], raw[u].a[j] & 0x7FFFFFFFu); m[4 + j] = max(m[4 + j], raw[u].b[j] & 0x7FFFFFFFu); }
        }
      }
    }
  }
  __shared__ uint32_t sm[W][NW][kWave];  // [wave][dword of the lane][lane]: conflict-free writes and reads
#pragma unroll
  for (int k = 0; k < NW; k++) sm[w][k][lane] = m[k];
  gate_wait<GATED>(gate, seen, C, init);   // wave 0; the barrier carries it to the threads that issue the atomics
  __syncthreads();
  // the strip's 512 columns over the first 512 threads (8 waves), each combining the W partial maxima of its column
  if (threadIdx.x < 8 * kWave) {
    const int c = threadIdx.x;                 // column c of the strip = element c % 8 of lane c / 8
    const int64_t col = (int64_t)bx * (kWave * 8) + c;
    if (col < plane) {
      const int l = c >> 3, e = c & 7;
      uint32_t r;
      if (PK) {
        uint32_t p = sm[0][e >> 1][l];
#pragma unroll
        for (int i = 1; i < W; i++) p = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, p), __builtin_bit_cast(u16x2, sm[i][e >> 1][l])));
        const uint32_t h = (e & 1) ? (p >> 16) : (p & 0xFFFFu);
        r = DT == DMXQ_BF16 ? (h << 16) : f2u(half_lo(h));
      } else {
        r = sm[0][e][l];
#pragma unroll
        for (int i = 1; i < W; i++) r = max(r, sm[i][e][l]);
      }
      const int64_t ch = inner == 1 ? col : (plane < (1ll << 31) ? (int64_t)((uint32_t)col / (uint32_t)inner) : col / inner);
      atomicMax((int*)&out[ch], (int)r);  // |x| patterns: int order == float order, a NaN pattern is above +Inf's
    }
  }
}

__global__ void qparams_kernel(const float* mn, const float* mx, int64_t G, int qmin, int qmax, int sym, float* scale,
                               int64_t* zp) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  const float eps = 1.1920928955078125e-07f;  // torch.finfo(torch.float32).eps (observer.py:37)
  const float min_neg = fminf(mn[g], 0.0f);
  const float max_pos = fmaxf(mx[g], 0.0f);
  if (sym) {
    const float m = fmaxf(-min_neg, max_pos);
    scale[g] = fmaxf(m / ((float)(qmax - qmin) / 2.0f), eps);
    zp[g] = 0;
  } else {
    const float s = fmaxf((max_pos - min_neg) / (float)(qmax - qmin), eps);
    float z = (float)qmin - rintf(min_neg / s);
    z = fminf(fmaxf(z, (float)qmin), (float)qmax);
    scale[g] = s;
    zp[g] = (int64_t)z;
  }
}

// grid = (ceil(C*inner / 256), splits over outer).  Lanes run along the contiguous (c, i) plane.
__global__ __launch_bounds__(kThreads) void channel_maxabs_kernel(const void* __restrict__ in, int dt, int64_t outer,
                                                                 int64_t C, int64_t inner, float* out) {
  const int64_t col = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  const int64_t plane = C * inner;
  if (col >= plane) return;
  uint32_t m = 0u;  // |x| as bit patterns: integer order == float order, and a NaN pattern (above +Inf's) propagates like torch.amax's
  for (int64_t o = blockIdx.y; o < outer; o += gridDim.y) m = max(m, f2u(load_rt(in, dt, o * plane + col)) & 0x7FFFFFFFu);
  atomicMax((int*)&out[col / inner], (int)m);
}

__global__ void smoothquant_scale_kernel(const float* a, const float* b, int64_t C, float alpha, float smin,
                                         float* scale) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float bb = fmaxf(b[c], smin);
  const float s = powf(a[c], alpha) / powf(bb, 1.0f - alpha);
  scale[c] = fmaxf(s, smin);
}


// torch.histc (numerical/observer.py:470-472, 489-491): one pass, workgroup-private LDS histogram of integer
// counts, flushed with one global atomic per non-empty bin per workgroup; the counts become fp32 in a second tiny
// launch that reuses the output buffer.  The bin of an element is ATen's fp32 expression, evaluated with IEEE
// multiply and divide: (int64)((x - lo) * bins / (hi - lo)), right edge into the last bin, out-of-range and NaN dropped.
constexpr int kHistThreads = 1024;
constexpr int kHistMaxBins = 8192;  // 32 KiB of LDS counters
// FAST: width in [2^-20, 2^20] (checked on the host): the quotient comes from div_for_clamped_int (common.hpp), which IS the IEEE
// quotient for 2^-100 <= |n| <= 2^100; n = (v - lo) * bins is below 2^34 for an in-range v, and a smaller |n| truncates to bin 0
// whatever its last bit is.  3 FMA-class operations instead of the ~14 of v_div_scale/fmas/fixup per element.
// (Tried and measured slower on the same box, profiles/r02_histc_variants.txt: a branch-free add of 0 for out-of-range elements;
// 2 / 4 / 8 interleaved copies of the counters against same-bin collisions; 128 or 512 workgroups.)
template <bool FAST>
__device__ __forceinline__ void hist_add(uint32_t* s, float v, float lo, float hi, float fb, const Recip& width, int bins) {
  if (v >= lo && v <= hi) {
    const float n = (v - lo) * fb;
    int pos = (int)(FAST ? div_for_clamped_int(n, width) : n / width.d);
    pos = pos < bins ? pos : bins - 1;
    atomicAdd(&s[pos], 1u);
  }
}
template <int DT, bool FAST, bool GATED>
__global__ __launch_bounds__(kHistThreads) void histc_kernel(const void* __restrict__ in, int64_t n, int bins,
                                                             float lo, float hi, int vec, uint32_t* counts, const InitGate gate) {
  constexpr int dt = DT;
  extern __shared__ uint32_t s_hist[];
  auto init = [&](int64_t i) { __hip_atomic_store(&counts[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  uint32_t wid;
  if (gate_open<GATED>(gate, bins, init, wid)) return;
  const uint32_t nb = gridDim.x - (GATED ? 1u : 0u);   // data workgroups
  unsigned seen = ~gate.epoch;
  for (int b = threadIdx.x; b < bins; b += kHistThreads) s_hist[b] = 0;
  __syncthreads();
  const float fb = (float)bins;
  const Recip width = make_recip(hi - lo);
  const int64_t stride = (int64_t)nb * kHistThreads;
  const int64_t t0 = (int64_t)wid * kHistThreads + threadIdx.x;
  if (vec) {
    constexpr int U = 4;  // 16-byte loads in flight per lane
    const int64_t nv = n / 8;
    for (int64_t t = t0; t < nv; t += U * stride) {
      Raw8<DT> raw[U];
#pragma unroll
      for (int u = 0; u < U; u++) raw[u] = load8_raw<DT>(in, (t + u * stride < nv ? t + u * stride : t) * 8);
      seen = gate_peek<GATED>(gate, seen, raw[U / 2].a.x);
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (u == 0 || t + u * stride < nv) {
          float a[8];
          widen8<DT>(raw[u], a);
#pragma unroll
          for (int k = 0; k < 8; k++) hist_add<FAST>(s_hist, a[k], lo, hi, fb, width, bins);
        }
      }
    }
    for (int64_t e = nv * 8 + t0; e < n; e += stride) hist_add<FAST>(s_hist, load_rt(in, dt, e), lo, hi, fb, width, bins);
  } else {
    for (int64_t e = t0; e < n; e += stride) hist_add<FAST>(s_hist, load_rt(in, dt, e), lo, hi, fb, width, bins);
  }
  gate_wait<GATED>(gate, seen, bins, init);   // wave 0; the barrier carries it to everyone
  __syncthreads();
  for (int b = threadIdx.x; b < bins; b += kHistThreads) {
    const uint32_t c = s_hist[b];
    if (c) atomicAdd(&counts[b], c);
  }
}
__global__ void hist_to_float_kernel(uint32_t* counts, int bins) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < bins) ((float*)counts)[b] = (float)counts[b];
}
}  // namespace dmxq

using namespace dmxq;

// FILL: mn / mx are initialised to +inf / -inf by a first launch (dmxq_group_minmax); without it the kernel's atomics fold this
// tensor's extrema INTO the values already there (dmxq_group_minmax_accumulate: a running min / max updated in ONE launch)
static int group_minmax_impl(const void* in, int dtype_in, int64_t outer, int64_t C, int64_t inner, int64_t group_size, float* mn, float* mx,
                             void* stream, bool fill) {
  if (!valid_dtype(dtype_in) || outer < 0 || C < 0 || inner < 0 || group_size < 1) return DMXQ_ERR_BAD_ARG;
  if (C == 0) return DMXQ_OK;
  if (!mn || !mx) return DMXQ_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int64_t G = (C + group_size - 1) / group_size;
  if (G > 65535) return DMXQ_ERR_UNSUPPORTED;
  const bool vec = outer * inner > 0 && aligned16(in) && (group_size * inner) % 8 == 0 && (C * inner) % 8 == 0;
  // the identities: by the kernel's own first workgroup behind the init gate (vector kernels), else by a launch in front of it
  InitGate gate{nullptr, 0u, 0};
  if (fill && vec && in) gate = take_gate(s, G);
  if (fill && !gate.on) DMXQ_LAUNCH(fill2_kernel, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, s, mn, INFINITY, mx, -INFINITY, G);
  if (outer * inner > 0) {
    if (!in) return DMXQ_ERR_BAD_ARG;
    const int64_t per_group = outer * group_size * inner;
    int64_t splits = (per_group + kThreads * 16 - 1) / (kThreads * 16);
    const int64_t cap = (kMaxBlocks + G - 1) / G;
    if (splits > cap) splits = cap;
    if (splits < 1) splits = 1;
    // (vec: every run [(o*C + g*gs) * inner, + len*inner) starts 16-byte aligned and is a whole number of 8-element vectors)
    if (vec && outer == 1) {
      const int64_t tile = (int64_t)kFlatThreads * (dtype_in == DMXQ_F32 ? 8 : 16);
      int64_t sv = (per_group / 8 + tile - 1) / tile;
      // <= 512 workgroups in all, and <= 160 per group: every workgroup ends in two atomics on its group's pair of addresses, and
      // same-address atomics serialise (~1.4 ns each) -- per-tensor on 32 MiB: 512 x 1 tile 9.3 us, 256 x 2 9.3, 192 8.8, 160 x 3-4
      // tiles 8.6, 128 x 4 8.8, 96 10.5 (too few loads in flight)
      int64_t capv = (512 + G - 1) / G;
      if (capv > 160) capv = 160;
      if (sv > capv) sv = capv;
      if (sv < 1) sv = 1;
#define DMXQ_MF1(D_, G_) DMXQ_LAUNCH((group_minmax_flat_kernel<D_, G_>), dim3((unsigned)(sv * G + ((G_) ? 1 : 0))), dim3(kFlatThreads), 0, s, in, C, inner, group_size, mn, mx, gate, make_fastdiv_u32(sv), (uint32_t)G)
#define DMXQ_MF(D_) do { if (gate.on) DMXQ_MF1(D_, true); else DMXQ_MF1(D_, false); } while (0)
      if (dtype_in == DMXQ_F32) DMXQ_MF(DMXQ_F32); else if (dtype_in == DMXQ_F16) DMXQ_MF(DMXQ_F16); else DMXQ_MF(DMXQ_BF16);
#undef DMXQ_MF
#undef DMXQ_MF1
    } else if (vec) {
      int64_t sv = (per_group / 8 + kMinmaxThreads * 8 - 1) / (kMinmaxThreads * 8);  // ~8 vectors per lane
      const int64_t capv = (512 + G - 1) / G;  // ~512 workgroups of 1024 threads in total
      if (sv > capv) sv = capv;
      if (sv < 1) sv = 1;
#define DMXQ_MM1(D_, G_) DMXQ_LAUNCH((group_minmax_vec_kernel<D_, G_>), dim3((unsigned)(sv * G + ((G_) ? 1 : 0))), dim3(kMinmaxThreads), 0, s, in, outer, C, inner, group_size, mn, mx, gate, make_fastdiv_u32(sv), (uint32_t)G)
#define DMXQ_MM(D_) do { if (gate.on) DMXQ_MM1(D_, true); else DMXQ_MM1(D_, false); } while (0)
      if (dtype_in == DMXQ_F32) DMXQ_MM(DMXQ_F32); else if (dtype_in == DMXQ_F16) DMXQ_MM(DMXQ_F16); else DMXQ_MM(DMXQ_BF16);
#undef DMXQ_MM
#undef DMXQ_MM1
    } else
      DMXQ_LAUNCH(group_minmax_kernel, dim3((unsigned)splits, (unsigned)G), dim3(kThreads), 0, s, in, dtype_in,
                         outer, C, inner, group_size, mn, mx);
  }
  return launch_status();
}

extern "C" int dmxq_group_minmax(const void* in, int dtype_in, int64_t outer, int64_t C, int64_t inner,
                                 int64_t group_size, float* mn, float* mx, void* stream) {
  return group_minmax_impl(in, dtype_in, outer, C, inner, group_size, mn, mx, stream, true);
}
extern "C" int dmxq_group_minmax_accumulate(const void* in, int dtype_in, int64_t outer, int64_t C, int64_t inner,
                                            int64_t group_size, float* mn, float* mx, void* stream) {
  return group_minmax_impl(in, dtype_in, outer, C, inner, group_size, mn, mx, stream, false);
}

extern "C" int dmxq_qparams(const float* mn, const float* mx, int64_t n_groups, int qmin, int qmax,
                            int symmetric_qscheme, float* scale, int64_t* zero_point, void* stream) {
  if (n_groups < 0 || qmax <= qmin) return DMXQ_ERR_BAD_ARG;
  if (n_groups == 0) return DMXQ_OK;
  if (!mn || !mx || !scale || !zero_point) return DMXQ_ERR_BAD_ARG;
  DMXQ_LAUNCH(qparams_kernel, dim3((unsigned)((n_groups + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mn,
                     mx, n_groups, qmin, qmax, symmetric_qscheme, scale, zero_point);
  return launch_status();
}

extern "C" int dmxq_channel_maxabs(const void* in, int dtype_in, int64_t outer, int64_t C, int64_t inner, float* out,
                                   void* stream) {
  if (!valid_dtype(dtype_in) || outer < 0 || C < 0 || inner < 0) return DMXQ_ERR_BAD_ARG;
  if (C == 0) return DMXQ_OK;
  if (!out) return DMXQ_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int64_t plane = C * inner;
  const bool vec = outer * plane > 0 && in && aligned16(in) && plane % 8 == 0;
  InitGate gate{nullptr, 0u, 0};
  if (vec) gate = take_gate(s, C);   // zeros by the kernel's first workgroup (the init gate above), else by a launch in front
  if (!gate.on) DMXQ_LAUNCH(fill2_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, s, out, 0.0f, (float*)nullptr, 0.0f, C);
  if (outer * plane > 0) {
    if (!in) return DMXQ_ERR_BAD_ARG;
    const int64_t gx = vec ? (plane / 8 + kWave - 1) / kWave : (plane + kThreads - 1) / kThreads;
    int64_t gy = kMaxBlocks / gx;
    if (gy < 1) gy = 1;
    if (gy > outer) gy = outer;
    int rows_in_flight = kMaxabsRows;
    if (vec) {  // one pass of 16 waves x 8 rows per workgroup, at most 64 row splits (each costs one atomic per channel)
      // a [1500, 768] activation (SmoothQuant calibration of Whisper-small) is 2 x 12 workgroups that way, 24 of 256 CUs, 10.2 us:
      // 4 / 2 rows per lane while that leaves the chip short of workgroups and stays within the 64 splits
      while (rows_in_flight > 2 && gx * ((outer + (kMaxabsThreads / kWave) * rows_in_flight - 1) / ((kMaxabsThreads / kWave) * rows_in_flight)) < 256 &&
             (outer + (kMaxabsThreads / kWave) * (rows_in_flight / 2) - 1) / ((kMaxabsThreads / kWave) * (rows_in_flight / 2)) <= 64)
        rows_in_flight /= 2;
      const int64_t rows_per_pass = (kMaxabsThreads / kWave) * rows_in_flight;
      gy = (outer + rows_per_pass - 1) / rows_per_pass;
      if (gy > 64) gy = 64;
    }
    if (gy > 65535) gy = 65535;
    if (vec && gx * gy > 0x7FFFFFF0ll) gy = 0x7FFFFFF0ll / gx > 0 ? 0x7FFFFFF0ll / gx : 1;   // (one-dimensional grid: strips x row splits)
    if (vec)
#define DMXQ_MAU1(D_, U_, G_) DMXQ_LAUNCH((channel_maxabs_vec_kernel<D_, U_, G_>), dim3((unsigned)(gx * gy + ((G_) ? 1 : 0))), dim3(kMaxabsThreads), 0, s, in, outer, C, inner, out, gate, make_fastdiv_u32(gx), (uint32_t)gy)
#define DMXQ_MAU(D_, U_) do { if (gate.on) DMXQ_MAU1(D_, U_, true); else DMXQ_MAU1(D_, U_, false); } while (0)
#define DMXQ_MA(D_) do { if (rows_in_flight == 8) DMXQ_MAU(D_, 8); else if (rows_in_flight == 4) DMXQ_MAU(D_, 4); else DMXQ_MAU(D_, 2); } while (0)
    { if (dtype_in == DMXQ_F32) DMXQ_MA(DMXQ_F32); else if (dtype_in == DMXQ_F16) DMXQ_MA(DMXQ_F16); else DMXQ_MA(DMXQ_BF16); }
#undef DMXQ_MA
#undef DMXQ_MAU
#undef DMXQ_MAU1
    else
      DMXQ_LAUNCH(channel_maxabs_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(kThreads), 0, s, in, dtype_in,
                         outer, C, inner, out);
  }
  return launch_status();
}

extern "C" int dmxq_smoothquant_scale(const float* a_maxabs, const float* b_maxabs, int64_t C, float alpha,
                                      float scale_min, float* scale, void* stream) {
  if (C < 0) return DMXQ_ERR_BAD_ARG;
  if (C == 0) return DMXQ_OK;
  if (!a_maxabs || !b_maxabs || !scale) return DMXQ_ERR_BAD_ARG;
  DMXQ_LAUNCH(smoothquant_scale_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     a_maxabs, b_maxabs, C, alpha, scale_min, scale);
  return launch_status();
}

extern "C" int dmxq_histc(const void* in, int dtype_in, int64_t n, int64_t bins, float lo, float hi, float* hist,
                          void* stream) {
  if (!valid_dtype(dtype_in) || n < 0 || bins < 1 || !hist || !(lo < hi) || isinf(lo) || isinf(hi))
    return DMXQ_ERR_BAD_ARG;
  if (bins > kHistMaxBins) return DMXQ_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  // the zeros: by the kernel's first workgroup behind the init gate (above), else by a memset in front of it
  InitGate gate{nullptr, 0u, 0};
  if (n > 0 && in) gate = take_gate(s, bins);
  if (!gate.on && hipMemsetAsync(hist, 0, (size_t)bins * sizeof(float), s) != hipSuccess) return DMXQ_ERR_LAUNCH;
  if (n > 0) {
    if (!in) return DMXQ_ERR_BAD_ARG;
    // one workgroup of 16 waves per CU: every workgroup ends with one global atomic per non-empty bin (512 workgroups x 2048
    // bins were 1 M atomics), and 128 leave half the CUs idle
    int64_t blocks = (n + kHistThreads * 32 - 1) / (kHistThreads * 32);
    if (blocks > 256) blocks = 256;
    const bool fast = recip_ok(hi - lo);
#define DMXQ_HC1(D_, F_, G_) DMXQ_LAUNCH((histc_kernel<D_, F_, G_>), dim3((unsigned)blocks + ((G_) ? 1u : 0u)), dim3(kHistThreads), (size_t)bins * sizeof(uint32_t), s, in, n, (int)bins, lo, hi, aligned16(in) ? 1 : 0, (uint32_t*)hist, gate)
#define DMXQ_HC(D_)                                                                  \
  do {                                                                               \
    if (fast) { if (gate.on) DMXQ_HC1(D_, true, true); else DMXQ_HC1(D_, true, false); }   \
    else { if (gate.on) DMXQ_HC1(D_, false, true); else DMXQ_HC1(D_, false, false); }      \
  } while (0)
    if (dtype_in == DMXQ_F32) DMXQ_HC(DMXQ_F32); else if (dtype_in == DMXQ_F16) DMXQ_HC(DMXQ_F16); else DMXQ_HC(DMXQ_BF16);
#undef DMXQ_HC
#undef DMXQ_HC1
    DMXQ_LAUNCH(hist_to_float_kernel, dim3((unsigned)((bins + 255) / 256)), dim3(256), 0, s, (uint32_t*)hist, (int)bins);
  }
  return launch_status();
}
