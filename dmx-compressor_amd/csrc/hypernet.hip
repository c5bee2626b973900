// csrc/hypernet.hip — the fused weight hypernet.
//
// DmxModule.weight_hypernet (modeling/nn/core.py:178-198) runs, on EVERY forward until the weights are folded,
//     w -> weight_sparsifier (N:M mask, x * mask) -> smoothquant.scale_weight (x * scale[c_in]) -> weight_storage_cast
//       -> weight_cast (BFP)
// as separate passes over the weight (score 4 B + w 2 B + y 4 B, then 4+4 B, then 4+4 B ... per element).  This kernel
// does mask -> scale -> BFP Q->DQ in ONE pass for the Linear layout (everything along the contiguous last dim):
// reads w (+ score, + the [C_in] scale vector), writes the quantised weight once.
//
// Bit-exactness with the unfused chain is kept by reproducing its dtype flow:
//   T1 = promote(dtype(w), dtype(score)) after the mask multiply (exact: x or +-0);
//   scale: fl32(x * s) rounded to T1 (ActivationWeightSmoothQuant.scale_weight's `.to(wgt.dtype)`);
//   BFP nearest-even on the T1 value (bfp_math.hpp), result rounded to T1, then to dtype_out (`.to(input dtype)`).
// Scope: inner == 1, L % B == 0, B = 2^k in [8, 512], L % 8 == 0, M in {0 (dense), 2, 4, 8}, nearest rounding.
#include <stdlib.h>

#include "hypernet_rows.hpp"
#include "lastdim.hpp"

namespace dmxq {

// units per lane of the single-tensor kernel: 2 for the masked BFP chain (hypernet_rows.hpp kHnUnitsSmall), 4 for the rest
constexpr int hn_units_of(int M, bool bfp) { return (M != 0 && bfp) ? kHnUnitsSmall : kHnUnits; }
template <int DTW, int DTS, int DTO, int M, bool HAS_SCALE, bool BFP = true, bool DIVIDE = false>
__global__ __launch_bounds__(kThreads) void hypernet_rows_kernel(HnArgs a) {
  // BFP16_64 (8 lanes per block: the BASIC rule's weight format) gets the branch-free form; other block sizes the runtime one
  // symmetric / asymmetric codes: chosen once per launch as well, not once per unit
  const bool asym = BFP && __builtin_amdgcn_readfirstlane(a.asym) != 0;
  constexpr int UN = hn_units_of(M, BFP);
  if (BFP && __builtin_amdgcn_readfirstlane(a.lpb) == 8) {
    if (asym) hypernet_rows_body<DTW, DTS, DTO, M, HAS_SCALE, BFP, 8, true, DIVIDE, UN>(a);
    else hypernet_rows_body<DTW, DTS, DTO, M, HAS_SCALE, BFP, 8, false, DIVIDE, UN>(a);
  } else {
    if (asym) hypernet_rows_body<DTW, DTS, DTO, M, HAS_SCALE, BFP, 0, true, DIVIDE, UN>(a);
    else hypernet_rows_body<DTW, DTS, DTO, M, HAS_SCALE, BFP, 0, false, DIVIDE, UN>(a);
  }
}

// Dense weight path with a SmoothQuant scale (M == 0, scale along the contiguous dim): w * s[c] -> BFP as an op of lastdim_kernel
// (lastdim.hpp) -- a lane keeps the scales of its 8 (float32: 4) columns in registers and handles up to 16 rows of them, one pass per
// workgroup, instead of re-reading 32 bytes of scale table for every 16 bytes of weight (hypernet_rows_body: 13.6 us on 4096 x 4096
// bf16, 62 % of the roofline).  Same arithmetic, element for element, as hypernet_rows_body<.., M = 0, HAS_SCALE, BFP>.
template <int DTW, bool ASYM, int LPBC>
struct HnLastOp {
  const float* scale;
  int lpb, wl;
  template <int N> struct RawParams { f32x4 sc[N / 4]; };
  template <int N> struct ChanParams { float s[N]; };
  template <int N>
  __device__ __forceinline__ RawParams<N> fetch_params(int64_t c0) const {
    RawParams<N> r;
#pragma unroll
    for (int k = 0; k < N / 4; k++) r.sc[k] = *(const f32x4*)(scale + c0 + 4 * k);
    return r;
  }
  template <int N>
  __device__ __forceinline__ ChanParams<N> make_params(const RawParams<N>& r) const {
    ChanParams<N> p;
#pragma unroll
    for (int k = 0; k < N; k += 4) {
      const f32x4 t = r.sc[k / 4];
      p.s[k] = t.x; p.s[k + 1] = t.y; p.s[k + 2] = t.z; p.s[k + 3] = t.w;
    }
    return p;
  }
  template <int N>
  __device__ __forceinline__ void apply_chan(const float (&xin)[N], const ChanParams<N>& p, float (&y)[N], int64_t) const {
    const int lanes = LPBC > 0 ? LPBC : __builtin_amdgcn_readfirstlane(lpb);
    float x[N];
    uint32_t mb = 0u;
#pragma unroll
    for (int k = 0; k < N; k++) {
      x[k] = round_to<DTW>(xin[k] * p.s[k]);
      mb = max(mb, f2u(x[k]) & 0x7FFFFFFFu);
    }
    mb = group_max_u32(mb, lanes);
    const bool fast_ok = bfp_fast_ok(mb, wl);
    {
      const BfpBlockParams bp = bfp_block_params<ASYM, true>(mb, wl);
#pragma unroll
      for (int k = 0; k < N; k++) y[k] = bfp_q1_fast<false, ASYM>(x[k], bp);
    }
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!fast_ok) != 0ull, 0)) {
      if (!fast_ok) {
        const BfpBlockParams bp = bfp_block_params<ASYM, false>(mb, wl);
#pragma unroll
        for (int k = 0; k < N; k++) y[k] = bfp_q1<DMXQ_ROUND_NEAREST, ASYM>(x[k], bp, wl, DMXQ_ROUND_NEAREST, 0u);
      }
    }
#pragma unroll
    for (int k = 0; k < N; k++) y[k] = round_to<DTW>(y[k]);   // CastTo's `.to(physical_dtype)`, then the caller's dtype
  }
};

// The ACTIVATION twin (round 5): x / s[c] -> BFP in float32 (dmxq_input_hypernet) as an op of lastdim_kernel too.  The tiled kernel
// below gives every lane-vector its own four channels, i.e. an IEEE division per element (~13 VALU cycles' worth); here a lane keeps
// its four channels for all its rows, so their RECIPROCALS are formed once per workgroup and a quotient is common.hpp's div_by_recip:
// RN(x / s) exactly, 6 operations, with the IEEE division as the cold redo of lanes holding a tiny / huge / Inf / NaN element or an
// out-of-range scale.  Then the float32 tile arithmetic of the hot kernel (double-rounding magic add, literal redo).
template <bool ASYM, int LPBC>
struct InLastOp {
  const float* scale;
  int lpb, wl;
  template <int N> struct RawParams { f32x4 sc[N / 4]; };
  template <int N> struct ChanParams { float s[N], rs[N]; bool fast; };
  template <int N>
  __device__ __forceinline__ RawParams<N> fetch_params(int64_t c0) const {
    RawParams<N> r;
#pragma unroll
    for (int k = 0; k < N / 4; k++) r.sc[k] = *(const f32x4*)(scale + c0 + 4 * k);
    return r;
  }
  template <int N>
  __device__ __forceinline__ ChanParams<N> make_params(const RawParams<N>& r) const {
    ChanParams<N> p;
    p.fast = true;
#pragma unroll
    for (int k = 0; k < N; k += 4) {
      const f32x4 t = r.sc[k / 4];
      p.s[k] = t.x; p.s[k + 1] = t.y; p.s[k + 2] = t.z; p.s[k + 3] = t.w;
    }
#pragma unroll
    for (int k = 0; k < N; k++) { p.rs[k] = 1.0f / p.s[k]; p.fast = p.fast && recip_ok(p.s[k]); }
    return p;
  }
  // lastdim.hpp OpDeferredRedo: straight-line for every row (reciprocal quotient, magic-add BFP); a row with an out-of-range element or
  // scale, or a block the magic add does not cover, is flagged -- by every lane of its block, the flag is taken on the block's
  // maximum and on the wave's ballot -- and redone with the IEEE division and the literal bit path after the stores
  static constexpr bool kDeferredRedo = true;
  template <int N>
  __device__ __forceinline__ bool apply_chan_flag(const float (&xin)[N], const ChanParams<N>& p, float (&y)[N], int64_t) const {
    const int lanes = LPBC > 0 ? LPBC : __builtin_amdgcn_readfirstlane(lpb);
    float x[N];
    bool ok = p.fast;
#pragma unroll
    for (int k = 0; k < N; k++) { x[k] = div_by_recip(xin[k], p.s[k], p.rs[k]); ok = ok && div_by_recip_ok(xin[k]); }
    uint32_t mb = 0u;
#pragma unroll
    for (int k = 0; k < N; k++) mb = max(mb, f2u(x[k]) & 0x7FFFFFFFu);
    // a lane whose quotients are not exact marks its block's maximum as unusable (an all-ones pattern: the largest): the whole block
    // is then flagged through the same DPP reduction
    mb = group_max_u32(ok ? mb : 0xFFFFFFFFu, lanes);
    const BfpBlockParams bp = bfp_block_params<ASYM, true>(mb, wl);
#pragma unroll
    for (int k = 0; k < N; k++) y[k] = bfp_q1_fast<false, ASYM>(x[k], bp);
    return mb == 0xFFFFFFFFu || !bfp_fast_ok(mb, wl);
  }
  template <int N>
  __device__ __forceinline__ void apply_chan_exact(const float (&xin)[N], const ChanParams<N>& p, float (&y)[N], int64_t) const {
    const int lanes = LPBC > 0 ? LPBC : __builtin_amdgcn_readfirstlane(lpb);
    float x[N];
    uint32_t mb = 0u;
#pragma unroll
    for (int k = 0; k < N; k++) {
      x[k] = xin[k] / p.s[k];   // smoothquant.py:255-268 `a / scale`, fp32
      mb = max(mb, f2u(x[k]) & 0x7FFFFFFFu);
    }
    mb = group_max_u32(mb, lanes);
    const BfpBlockParams bp = bfp_block_params<ASYM, false>(mb, wl);
#pragma unroll
    for (int k = 0; k < N; k++) y[k] = bfp_q1<DMXQ_ROUND_NEAREST, ASYM>(x[k], bp, wl, DMXQ_ROUND_NEAREST, 0u);
  }
  template <int N>
  __device__ __forceinline__ void apply_chan(const float (&xin)[N], const ChanParams<N>& p, float (&y)[N], int64_t e0) const {
    apply_chan_exact(xin, p, y, e0);
  }
};
template <int DTX>
static int launch_in_lastdim(const void* x, void* out, const float* scale, int64_t rows, int64_t L, int64_t B, int wl, bool asym, hipStream_t s) {
  constexpr int IVB = 4 * Elem<DTX>::bytes;   // a lane owns 4 elements: 16 contiguous bytes of float32 output
  const int64_t lpb = B / 4;
  if (lpb < 1 || lpb > 64 || (lpb & (lpb - 1)) != 0 || L % B != 0) return DMXQ_ERR_UNSUPPORTED;  // a block is lpb adjacent lanes of ONE wave
#define DMXQ_INL(L_) (asym ? launch_lastdim_typed<DTX, DMXQ_F32, InLastOp<true, L_>, IVB>(x, out, rows, L, InLastOp<true, L_>{scale, (int)lpb, wl}, s) \
                           : launch_lastdim_typed<DTX, DMXQ_F32, InLastOp<false, L_>, IVB>(x, out, rows, L, InLastOp<false, L_>{scale, (int)lpb, wl}, s))
  if (lpb == 16) return DMXQ_INL(16);   // BFP16_64, the BASIC rules' activation format
  if (lpb == 4) return DMXQ_INL(4);
  return DMXQ_INL(0);
#undef DMXQ_INL
}

// DMXQ_ERR_UNSUPPORTED: geometry the lastdim kernel does not take (the caller falls back to hypernet_rows_kernel)
template <int DTW, int DTO>
static int launch_hn_lastdim(const void* w, void* out, const float* scale, int64_t rows, int64_t L, int64_t B, int wl, bool asym, hipStream_t s) {
  constexpr int EPL = 16 / Elem<DTW>::bytes;
  const int64_t lpb = B / EPL;
  if (lpb < 1 || lpb > 64) return DMXQ_ERR_UNSUPPORTED;  // a block is lpb adjacent lanes of ONE wave
  if (lpb == 8) {
    return asym ? launch_lastdim_typed<DTW, DTO>(w, out, rows, L, HnLastOp<DTW, true, 8>{scale, 8, wl}, s)
                : launch_lastdim_typed<DTW, DTO>(w, out, rows, L, HnLastOp<DTW, false, 8>{scale, 8, wl}, s);
  }
  return asym ? launch_lastdim_typed<DTW, DTO>(w, out, rows, L, HnLastOp<DTW, true, 0>{scale, (int)lpb, wl}, s)
              : launch_lastdim_typed<DTW, DTO>(w, out, rows, L, HnLastOp<DTW, false, 0>{scale, (int)lpb, wl}, s);
}

template <int DTW, int DTS, int DTO>
static int launch_hn(const HnArgs& a, int M, bool has_scale, hipStream_t s) {
  // masked chains (workgroup-contiguous tiles of kThreads x 4 units): ONE pass per workgroup -- a grid capped at 2048 looping
  // workgroups measured 233 us on the seven Llama-3-8B weights against 205 us for one-pass workgroups (the multi-tensor kernel on
  // the same tensors, profiles/r04_shard_sets.txt: 70 -> 80 % of the roofline); the dense path keeps its strided, capped grid
  const int64_t un = hn_units_of(M, true);
  const int64_t tiles = (a.n_units + (int64_t)kThreads * un - 1) / ((int64_t)kThreads * un);
  const int grid = (M != 0 && tiles < ((int64_t)1 << 31)) ? (int)tiles : grid_for((a.n_units + 3) / 4);
#define DMXQ_HN(M_, S_) DMXQ_LAUNCH((hypernet_rows_kernel<DTW, DTS, DTO, M_, S_>), dim3(grid), dim3(kThreads), 0, s, a)
  if (has_scale) { switch (M) { case 0: DMXQ_HN(0, true); break; case 2: DMXQ_HN(2, true); break; case 4: DMXQ_HN(4, true); break; default: DMXQ_HN(8, true); } }
  else { switch (M) { case 0: DMXQ_HN(0, false); break; case 2: DMXQ_HN(2, false); break; case 4: DMXQ_HN(4, false); break; default: DMXQ_HN(8, false); } }
#undef DMXQ_HN
  return launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------------
// Activation path (dmxq_input_hypernet): x / scale[c] -> BFP, tiled like the hot kernel (bfp_rows.hpp).  A lane owns 4 ELEMENTS
// per vector -- 8 bytes of a 16-bit input, 16 of an fp32 one -- so that every fp32 store instruction of a wave covers one contiguous
// KiB (two strided 16-byte halves per lane cost ~40 % of the bandwidth, bfp_rows.hpp).  The quotients (IEEE fp32 divisions: they
// ARE the result, common.hpp's reciprocal form does not apply) replace the raw vector in registers and from there on the tile is
// an fp32 -> fp32 tile of bfp_rows_vector: block maximum by DPP, magic-add codes (double rounding) with a cold literal redo.
// The 4 scales of a vector are requested BEFORE the tile's own loads (vector memory returns in order).
constexpr int kInThreads = 256, kInUnroll = 2;  // measured on 4096x4096 bf16: 256x8 21.6, 256x4 21.1, 256x2 19.5, 512x2 20.6 us (the IEEE divisions make this VALU-heavy: small tiles stagger)
struct InArgs {
  const void* x; const float* scale; void* out;
  int64_t n_vec, L;   // 4-element vectors
  int lpb, wl, asym, small;
  FastDiv31 f_L;
};
template <int DTX, bool ASYM, int LPBC>
__device__ __forceinline__ void input_rows_tile(const InArgs& a) {
  constexpr int U = kInUnroll, T = kInThreads;
  constexpr int IVB = 4 * Elem<DTX>::bytes;  // 8 or 16 input bytes per lane-vector
  const int lpb = LPBC > 0 ? LPBC : __builtin_amdgcn_readfirstlane(a.lpb);
  const int in_blk = threadIdx.x & (lpb - 1);
  const int64_t base = (int64_t)blockIdx.x * (T * U) + threadIdx.x;
  f32x4 sc[U];
  int64_t vc[U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    const int64_t v = base + u * T;
    vc[u] = v < a.n_vec ? v : a.n_vec - lpb + in_blk;  // past the end: the same lane position of the last block (not stored)
    const int64_t e = vc[u] * 4;
    const int64_t c = a.small ? (int64_t)((uint32_t)e - a.f_L.div((uint32_t)e) * (uint32_t)a.L) : e % a.L;
    sc[u] = *(const f32x4*)(a.scale + c);
  }
  __builtin_amdgcn_sched_barrier(0);
  u32x4 raw[U];
#pragma unroll
  for (int u = 0; u < U; u++) raw[u] = load_rawv<IVB>((const char*)a.x + vc[u] * IVB, 0u);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int u = 0; u < U; u++) {
    float xw[16 / Elem<DTX>::bytes];
    widen<DTX, 16 / Elem<DTX>::bytes>(raw[u], xw);
    u32x4 q;
#pragma unroll
    for (int k = 0; k < 4; k++) q[k] = f2u(xw[k] / sc[u][k]);  // smoothquant.py:255-268 `a / scale`, fp32
    const uint32_t mb = group_max_u32(absmax_bits<DMXQ_F32>(q), lpb);
    const bool ok = bfp_fast_ok(mb, a.wl);
    OutVec<DMXQ_F32, 4> o = bfp_rows_vector<DMXQ_F32, DMXQ_F32, DMXQ_ROUND_NEAREST, ASYM, 1, true, 4>(q, mb, 0, a.wl, DMXQ_ROUND_NEAREST, false, 0ull);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!ok) != 0ull, 0)) {
      if (!ok) o = bfp_rows_vector<DMXQ_F32, DMXQ_F32, DMXQ_ROUND_NEAREST, ASYM, 1, false, 4>(q, mb, 0, a.wl, DMXQ_ROUND_NEAREST, false, 0ull);
    }
    raw[u] = u32x4{o.w[0], o.w[1], o.w[2], o.w[3]};
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int u = 0; u < U; u++) {
    const int64_t v = base + u * T;
    if (v < a.n_vec) __builtin_nontemporal_store(raw[u], (u32x4*)((char*)a.out + v * 16));
  }
}
template <int DTX>
__global__ __launch_bounds__(kInThreads) void input_rows_kernel(InArgs a) {
  const int lpb = __builtin_amdgcn_readfirstlane(a.lpb);
  const bool asym = __builtin_amdgcn_readfirstlane(a.asym) != 0;
#define DMXQ_IN(L_) do { if (asym) input_rows_tile<DTX, true, L_>(a); else input_rows_tile<DTX, false, L_>(a); } while (0)
  switch (lpb) {  // the usual block sizes with a compile-time lane count (branch-free DPP maxima), as bfp_rows_kernel
    case 4: DMXQ_IN(4); break;
    case 16: DMXQ_IN(16); break;
    default: DMXQ_IN(0); break;
  }
#undef DMXQ_IN
}

// ---------------------------------------------------------------------------------------------------------------------------
// The weight chain for layouts whose blocked dimension is NOT the contiguous one (dmxq_weight_hypernet_strided): w = [outer, L,
// inner] with N:M groups, SmoothQuant channels and BFP blocks all along L -- Conv1d / Conv2d weights [out, in, k...] with block_dim = 1
// (modeling/nn/torch_modules.py:582-585, 674-677).  These tensors are small (Whisper conv2 [768,768,3]: 7 MB of fp32; LeNet conv2:
// 2400 elements) and the call is launch-bound, so the kernel is the plain statement of the chain: one lane per (outer, block, inner
// position) walks its <= B rows twice -- the block maximum of the masked, scaled values, then the codes -- adjacent lanes on adjacent
// inner positions.  Any block size and a ragged last block (torch.split), runtime dtypes, the dtype flow of the header comment.
struct HsArgs {
  const void* w; const void* score; const float* scale; void* out;
  int64_t outer, L, inner, nblk;   // nblk = ceil(L / B)
  int B, K, M, wl, dtw, dts, dto, t1;
};
__device__ __forceinline__ float round_rt(int dt, float v) {
  if (dt == DMXQ_BF16) return round_to<DMXQ_BF16>(v);
  if (dt == DMXQ_F16) return round_to<DMXQ_F16>(v);
  return v;
}
// rows [j0, j0 + 8) of a lane's block after mask and scale, as T1 values (rows past `len` are 0: they cannot raise a maximum)
__device__ __forceinline__ void hs_rows8(const HsArgs& a, int64_t base, int64_t c0, int j0, int len, float (&v)[8]) {
  float s[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const bool in = j0 + j < len;
    v[j] = in ? load_rt(a.w, a.dtw, base + (int64_t)(j0 + j) * a.inner) : 0.0f;
    s[j] = (in && a.M) ? load_rt(a.score, a.dts, base + (int64_t)(j0 + j) * a.inner) : 0.0f;
  }
  if (a.M) {  // ranks inside the groups of M rows (M in {2, 4, 8} divides 8 and the block size)
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int g0 = j & ~(a.M - 1);
      const int32_t kj = hn_sort_key(s[j]);
      int rank = 0;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const int32_t ki = hn_sort_key(s[i]);
        const bool same_group = i >= g0 && i < g0 + a.M && i != j;
        rank += (same_group && (ki < kj || (ki == kj && i < j))) ? 1 : 0;
      }
      v[j] = v[j] * (rank >= a.M - a.K ? 1.0f : 0.0f);  // a real multiply: -w * 0 = -0 (sparse.py:300)
    }
  }
  if (a.scale) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int64_t c = c0 + j0 + j < a.L ? c0 + j0 + j : a.L - 1;
      v[j] = round_rt(a.t1, v[j] * a.scale[c]);   // scale_weight's `.to(wgt.dtype)`
    }
  }
}
template <bool ASYM>
__global__ __launch_bounds__(kThreads) void hypernet_strided_kernel(const HsArgs a) {
  const int64_t n = a.outer * a.nblk * a.inner;
  for (int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x; t < n; t += (int64_t)gridDim.x * kThreads) {
    const int64_t q = t / a.inner, i = t - q * a.inner, o = q / a.nblk, b = q - o * a.nblk;
    const int64_t c0 = b * a.B;
    const int len = (int)(a.L - c0 < a.B ? a.L - c0 : a.B);
    const int64_t base = (o * a.L + c0) * a.inner + i;
    uint32_t mb = 0u;
    for (int j0 = 0; j0 < len; j0 += 8) {
      float v[8];
      hs_rows8(a, base, c0, j0, len, v);
#pragma unroll
      for (int j = 0; j < 8; j++) mb = max(mb, f2u(v[j]) & 0x7FFFFFFFu);  // integer max of |x| patterns: NaN propagates like torch.max
    }
    const BfpBlockParams p = bfp_block_params<ASYM, false>(mb, a.wl);
    for (int j0 = 0; j0 < len; j0 += 8) {
      float v[8];
      hs_rows8(a, base, c0, j0, len, v);
#pragma unroll
      for (int j = 0; j < 8; j++) {
        if (j0 + j < len) {
          const float y = round_rt(a.t1, bfp_q1<DMXQ_ROUND_NEAREST, ASYM>(v[j], p, a.wl, DMXQ_ROUND_NEAREST, 0u));  // CastTo's `.to(physical dtype)`
          store_rt(a.out, a.dto, base + (int64_t)(j0 + j) * a.inner, y);
        }
      }
    }
  }
}

}  // namespace dmxq

using namespace dmxq;

// Typed N:M mask-and-multiply over a flat [n] stream of whole M-groups (inner == 1), y = x * mask(score); called by
// dmxq_nm_mask (nm_mask.hip) before its generic kernels.  DMXQ_ERR_UNSUPPORTED: not one of the instantiated dtype triples.
extern "C" int dmxq_internal_nm_sparsify_typed(const void* score, int dtype_score, const void* x, int dtype_x, void* y, int dtype_y,
                                               int64_t n, int K, int M, void* stream) {
  if (!(M == 2 || M == 4 || M == 8) || n % 8 != 0 || !aligned16(score) || !aligned16(x) || !aligned16(y)) return DMXQ_ERR_UNSUPPORTED;
  const HnArgs a{x, score, nullptr, y, n / 8, n, K, 1, 8, 0, n < ((int64_t)1 << 31) ? 1 : 0, make_fastdiv31(n)};
  hipStream_t s = (hipStream_t)stream;
  const int64_t tiles = (a.n_units + (int64_t)kThreads * kHnUnits - 1) / ((int64_t)kThreads * kHnUnits);
  const int grid = tiles < ((int64_t)1 << 31) ? (int)tiles : grid_for((a.n_units + 3) / 4);   // one pass per workgroup (launch_hn)
#define DMXQ_NMT(W_, S_, O_)                                                                                          \
  if (dtype_x == W_ && dtype_score == S_ && dtype_y == O_) {                                                          \
    switch (M) {                                                                                                      \
      case 2: DMXQ_LAUNCH((hypernet_rows_kernel<W_, S_, O_, 2, false, false>), dim3(grid), dim3(kThreads), 0, s, a); break; \
      case 4: DMXQ_LAUNCH((hypernet_rows_kernel<W_, S_, O_, 4, false, false>), dim3(grid), dim3(kThreads), 0, s, a); break; \
      default: DMXQ_LAUNCH((hypernet_rows_kernel<W_, S_, O_, 8, false, false>), dim3(grid), dim3(kThreads), 0, s, a); break; \
    }                                                                                                                 \
    return launch_status();                                                                                           \
  }
  DMXQ_NMT(DMXQ_BF16, DMXQ_BF16, DMXQ_BF16)
  DMXQ_NMT(DMXQ_F16, DMXQ_F16, DMXQ_F16)
  DMXQ_NMT(DMXQ_F32, DMXQ_F32, DMXQ_F32)
  DMXQ_NMT(DMXQ_BF16, DMXQ_F32, DMXQ_F32)
  DMXQ_NMT(DMXQ_F16, DMXQ_F32, DMXQ_F32)
  DMXQ_NMT(DMXQ_BF16, DMXQ_F32, DMXQ_BF16)
#undef DMXQ_NMT
  return DMXQ_ERR_UNSUPPORTED;
}

extern "C" int dmxq_weight_hypernet(const void* w, int dtype_w, const void* score, int dtype_score, int K, int M,
                                    const float* sq_scale, void* out, int dtype_out, int64_t rows, int64_t L,
                                    int64_t block_size, int precision, int symmetric, void* stream) {
  if (!valid_dtype(dtype_w) || !valid_dtype(dtype_out) || rows < 0 || L < 0 || block_size < 1) return DMXQ_ERR_BAD_ARG;
  if (M != 0 && (!score || !valid_dtype(dtype_score) || K < 1 || K > M)) return DMXQ_ERR_BAD_ARG;
  const int64_t B = block_size;
  // fusable geometry; anything else is the caller's job to run unfused
  if (!(M == 0 || M == 2 || M == 4 || M == 8) || (B & (B - 1)) != 0 || B < 8 || B > 512 || L % B != 0 || L % 8 != 0 ||
      precision < 2 || precision > 20)
    return DMXQ_ERR_UNSUPPORTED;
  if (rows * L == 0) return DMXQ_OK;
  if (!w || !out || !aligned16(w) || !aligned16(out) || (score && !aligned16(score)) || (sq_scale && !aligned16(sq_scale)))
    return w && out ? DMXQ_ERR_UNSUPPORTED : DMXQ_ERR_BAD_ARG;
  const HnArgs a{w, score, sq_scale, out, rows * L / 8, L, K, (int)(B / 8), precision, symmetric ? 0 : 1,
                 rows * L < ((int64_t)1 << 31) ? 1 : 0, make_fastdiv31(L)};
  hipStream_t s = (hipStream_t)stream;
  if (M == 0 && sq_scale) {  // dense + SmoothQuant scale: the per-column scales in registers (HnLastOp)
    int rc = DMXQ_ERR_UNSUPPORTED;
#define DMXQ_HL(W_, O_) \
  if (dtype_w == W_ && dtype_out == O_) rc = launch_hn_lastdim<W_, O_>(w, out, sq_scale, rows, L, B, precision, !symmetric, s);
    DMXQ_HL(DMXQ_BF16, DMXQ_BF16)
    DMXQ_HL(DMXQ_BF16, DMXQ_F32)
    DMXQ_HL(DMXQ_F16, DMXQ_F16)
    DMXQ_HL(DMXQ_F16, DMXQ_F32)
    DMXQ_HL(DMXQ_F32, DMXQ_F32)
#undef DMXQ_HL
    if (rc != DMXQ_ERR_UNSUPPORTED) return rc;
  }
  const int ds = M ? dtype_score : dtype_w;
#define DMXQ_DT(W_, S_, O_) \
  if (dtype_w == W_ && ds == S_ && dtype_out == O_) return launch_hn<W_, S_, O_>(a, M, sq_scale != nullptr, s);
  DMXQ_DT(DMXQ_BF16, DMXQ_F32, DMXQ_BF16)   // bf16 weight, fp32 score Parameter, result straight to the matmul dtype
  DMXQ_DT(DMXQ_BF16, DMXQ_F32, DMXQ_F32)    // ... or fp32, the reference's `_weight` dtype in this case
  DMXQ_DT(DMXQ_BF16, DMXQ_BF16, DMXQ_BF16)  // dense / |w| score
  DMXQ_DT(DMXQ_F16, DMXQ_F32, DMXQ_F16)
  DMXQ_DT(DMXQ_F16, DMXQ_F32, DMXQ_F32)
  DMXQ_DT(DMXQ_F16, DMXQ_F16, DMXQ_F16)
  DMXQ_DT(DMXQ_F32, DMXQ_F32, DMXQ_F32)
#undef DMXQ_DT
  return DMXQ_ERR_UNSUPPORTED;
}

// The activation twin: SmoothQuant input scaling -> input cast in one pass over x[rows, L] (channels and BFP blocks along the
// contiguous last dim): out = BFP_QDQ(x / sq_scale[c]) with the quotient in fp32 (torch's promotion of the input dtype and the fp32
// scale: what `a / scale` yields in smoothquant.py:255-268 and what CastTo then takes as its physical dtype, cast.py:262,306).
extern "C" int dmxq_input_hypernet(const void* x, int dtype_x, const float* sq_scale, void* out, int dtype_out, int64_t rows,
                                   int64_t L, int64_t block_size, int precision, int symmetric, void* stream) {
  if (!valid_dtype(dtype_x) || !valid_dtype(dtype_out) || rows < 0 || L < 0 || block_size < 1) return DMXQ_ERR_BAD_ARG;
  const int64_t B = block_size;
  if ((B & (B - 1)) != 0 || B < 8 || B > 512 || L % B != 0 || L % 8 != 0 || precision < 2 || precision > 20 || dtype_out != DMXQ_F32)
    return DMXQ_ERR_UNSUPPORTED;
  if (rows * L == 0) return DMXQ_OK;
  if (!x || !out || !sq_scale) return DMXQ_ERR_BAD_ARG;
  if (!aligned16(x) || !aligned16(out) || !aligned16(sq_scale)) return DMXQ_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  // per-lane channels with hoisted reciprocals (InLastOp, round 5); DMXQ_INPUT_HYPERNET_TILED=1 keeps the tiled kernel (A/B runs)
  static const bool force_tiled = [] { const char* e = getenv("DMXQ_INPUT_HYPERNET_TILED"); return e && e[0] == '1'; }();
  if (!force_tiled) {
    int rc = DMXQ_ERR_UNSUPPORTED;
    if (dtype_x == DMXQ_BF16) rc = launch_in_lastdim<DMXQ_BF16>(x, out, sq_scale, rows, L, B, precision, !symmetric, s);
    else if (dtype_x == DMXQ_F16) rc = launch_in_lastdim<DMXQ_F16>(x, out, sq_scale, rows, L, B, precision, !symmetric, s);
    else rc = launch_in_lastdim<DMXQ_F32>(x, out, sq_scale, rows, L, B, precision, !symmetric, s);
    if (rc != DMXQ_ERR_UNSUPPORTED) return rc;
  }
  if (B / 4 <= 64) {  // the tiled kernel: a block = B / 4 adjacent lanes of one wave
    const int64_t n_vec = rows * L / 4, tiles = (n_vec + kInThreads * kInUnroll - 1) / (kInThreads * kInUnroll);
    if (tiles <= 0x7FFFFFFF) {
      const InArgs ia{x, sq_scale, out, n_vec, L, (int)(B / 4), precision, symmetric ? 0 : 1, rows * L < ((int64_t)1 << 31) ? 1 : 0,
                      make_fastdiv31(L)};
      if (dtype_x == DMXQ_BF16) DMXQ_LAUNCH(input_rows_kernel<DMXQ_BF16>, dim3((unsigned)tiles), dim3(kInThreads), 0, s, ia);
      else if (dtype_x == DMXQ_F16) DMXQ_LAUNCH(input_rows_kernel<DMXQ_F16>, dim3((unsigned)tiles), dim3(kInThreads), 0, s, ia);
      else DMXQ_LAUNCH(input_rows_kernel<DMXQ_F32>, dim3((unsigned)tiles), dim3(kInThreads), 0, s, ia);
      return launch_status();
    }
  }
  const HnArgs a{x, nullptr, sq_scale, out, rows * L / 8, L, 0, (int)(B / 8), precision, symmetric ? 0 : 1,
                 rows * L < ((int64_t)1 << 31) ? 1 : 0, make_fastdiv31(L)};
  const int grid = grid_for((a.n_units + 3) / 4);
#define DMXQ_IH(W_) DMXQ_LAUNCH((hypernet_rows_kernel<W_, W_, DMXQ_F32, 0, true, true, true>), dim3(grid), dim3(kThreads), 0, s, a)
  if (dtype_x == DMXQ_BF16) DMXQ_IH(DMXQ_BF16); else if (dtype_x == DMXQ_F16) DMXQ_IH(DMXQ_F16); else DMXQ_IH(DMXQ_F32);
#undef DMXQ_IH
  return launch_status();
}

// The chain of dmxq_weight_hypernet for [outer, L, inner] layouts (conv weights blocked along in-channels): see hypernet_strided_kernel.
extern "C" int dmxq_weight_hypernet_strided(const void* w, int dtype_w, const void* score, int dtype_score, int K, int M,
                                            const float* sq_scale, void* out, int dtype_out, int64_t outer, int64_t L, int64_t inner,
                                            int64_t block_size, int precision, int symmetric, void* stream) {
  if (!valid_dtype(dtype_w) || !valid_dtype(dtype_out) || outer < 0 || L < 0 || inner < 1 || block_size < 1) return DMXQ_ERR_BAD_ARG;
  if (M != 0 && (!score || !valid_dtype(dtype_score) || K < 1 || K > M)) return DMXQ_ERR_BAD_ARG;
  const int64_t B = block_size;
  if (!(M == 0 || M == 2 || M == 4 || M == 8) || B < 2 || B > (1 << 20) || (M && (B % M != 0 || L % M != 0)) || precision < 2 || precision > 22)
    return DMXQ_ERR_UNSUPPORTED;
  if (outer * L * inner == 0) return DMXQ_OK;
  if (!w || !out) return DMXQ_ERR_BAD_ARG;
  // T1: dtype after the mask multiply = torch's promotion of (w, score); the weight's own dtype without a mask
  const int t1 = M == 0 ? dtype_w : ((dtype_w == DMXQ_F32 || dtype_score == DMXQ_F32 || dtype_w != dtype_score) ? DMXQ_F32 : dtype_w);
  const int64_t nblk = (L + B - 1) / B;
  const HsArgs a{w, M ? score : nullptr, sq_scale, out, outer, L, inner, nblk, (int)B, K, M, precision, dtype_w, M ? dtype_score : dtype_w, dtype_out, t1};
  const int grid = grid_for(outer * nblk * inner);
  if (symmetric) DMXQ_LAUNCH(hypernet_strided_kernel<false>, dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, a);
  else DMXQ_LAUNCH(hypernet_strided_kernel<true>, dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, a);
  return launch_status();
}
