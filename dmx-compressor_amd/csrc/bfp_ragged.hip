// csrc/bfp_ragged.hip — BFP Q->DQ along the contiguous dim for row lengths the flat-stream kernel cannot take:
// L % B != 0 (torch.split's ragged last block, numerical/format.py:324-326 — LeNet fc in=400, attention rows of
// 1500, ...) and/or rows that are not 16-byte aligned.
//
// LDS staging: a workgroup copies a batch of row segments from HBM into LDS with the widest access the row
// pitch allows, each segment into a slot whose pitch is rounded UP to a multiple of B and zero-padded.  Inside
// LDS the data then looks exactly like the aligned case: 16-byte vectors, a block spans B/EPL adjacent lanes,
// block max by DPP, same arithmetic (bfp_math.hpp).  Results go to a second LDS region and are copied back to
// HBM coalesced.  HBM traffic stays 1 read + 1 write per element; LDS carries the re-alignment.
#include "bfp_math.hpp"

namespace dmxq {

constexpr int kRaggedLdsBytes = 32 * 1024;  // per workgroup: 4-5 workgroups per CU of the 160 KiB, phases of different workgroups overlap
constexpr int64_t kRaggedSeg = 4096;        // elements per row segment (rows longer than this are cut at multiples of B)

// all slots of a batch: item (row, segment) <-> LDS slot j.  Threads form a (slot, chunk) grid whose chunk extent is
// a power of two, so the mapping needs shifts only (no per-access division); nseg == 1 (rows that fit one segment,
// the common case) skips the item -> (row, segment) division as well.
template <int CW, bool TO_LDS>
__device__ __forceinline__ void copy_slots_w(char* lds, char* glb, int eb, int64_t it0, int64_t nit, int64_t nseg,
                                             int64_t seg, int64_t segp, int64_t L) {
  const uint32_t cps = (uint32_t)((seg * eb + CW - 1) / CW);  // chunks per full segment
  uint32_t cp_log = 0;
  while ((1u << cp_log) < cps && cp_log < 8) cp_log++;         // chunk lanes = 2^cp_log <= 256
  const uint32_t tc = threadIdx.x & ((1u << cp_log) - 1u), tj = threadIdx.x >> cp_log, nj = kThreads >> cp_log;
  for (int64_t j = tj; j < nit; j += nj) {
    const int64_t item = it0 + j;
    const int64_t row = nseg == 1 ? item : item / nseg, sg = nseg == 1 ? 0 : item % nseg;
    const int64_t len = (L - sg * seg < seg) ? (L - sg * seg) : seg;
    const uint32_t nch = (uint32_t)((len * eb + CW - 1) / CW);
    char* l = lds + j * segp * eb;
    char* g = glb + (row * L + sg * seg) * eb;
    for (uint32_t c = tc; c < nch; c += (1u << cp_log)) {
      const uint32_t off = c * CW;
      if (CW == 16) { if (TO_LDS) *(u32x4*)(l + off) = __builtin_nontemporal_load((const u32x4*)(g + off)); else __builtin_nontemporal_store(*(const u32x4*)(l + off), (u32x4*)(g + off)); }
      else if (CW == 8) { if (TO_LDS) *(u32x2*)(l + off) = *(const u32x2*)(g + off); else *(u32x2*)(g + off) = *(const u32x2*)(l + off); }
      else if (CW == 4) { if (TO_LDS) *(uint32_t*)(l + off) = *(const uint32_t*)(g + off); else *(uint32_t*)(g + off) = *(const uint32_t*)(l + off); }
      else { if (TO_LDS) *(uint16_t*)(l + off) = *(const uint16_t*)(g + off); else *(uint16_t*)(g + off) = *(const uint16_t*)(l + off); }
    }
  }
}
template <bool TO_LDS>
__device__ __forceinline__ void copy_slots(int cw, char* lds, char* glb, int eb, int64_t it0, int64_t nit, int64_t nseg,
                                           int64_t seg, int64_t segp, int64_t L) {
  if (cw == 16) copy_slots_w<16, TO_LDS>(lds, glb, eb, it0, nit, nseg, seg, segp, L);
  else if (cw == 8) copy_slots_w<8, TO_LDS>(lds, glb, eb, it0, nit, nseg, seg, segp, L);
  else if (cw == 4) copy_slots_w<4, TO_LDS>(lds, glb, eb, it0, nit, nseg, seg, segp, L);
  else copy_slots_w<2, TO_LDS>(lds, glb, eb, it0, nit, nseg, seg, segp, L);
}

template <int DTI, int DTO, int RND, bool ASYM, int FAST>
__global__ __launch_bounds__(kThreads) void bfp_lds_rows_kernel(const void* __restrict__ in, void* __restrict__ out,
                                                               int64_t rows, int64_t L, int B, int wl, int rounding,
                                                               uint64_t seed, int cw_in, int cw_out) {
  constexpr int EPL = 16 / Elem<DTI>::bytes;
  constexpr int IB = Elem<DTI>::bytes, OB = Elem<DTO>::bytes;
  constexpr int OVB = EPL * OB;
  constexpr bool kFast = FAST != 0 && RND == DMXQ_ROUND_NEAREST;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const bool stoch = (RND == kRuntimeRounding) && rounding == DMXQ_ROUND_STOCHASTIC;
  const int64_t seg = L < kRaggedSeg ? L : kRaggedSeg;               // segment length (last one of a row may be shorter)
  const int64_t nseg = (L + seg - 1) / seg;
  const int64_t segp = (seg + B - 1) / B * B;                        // LDS slot pitch: whole blocks
  const int64_t items = rows * nseg;
  int64_t R = kRaggedLdsBytes / (segp * (IB + OB));                  // slots per batch
  if (R < 1) R = 1;
  char* lds_in = smem;
  char* lds_out = smem + R * segp * IB;
  const int lpb = __builtin_amdgcn_readfirstlane(B / EPL);
  const int64_t vps = segp / EPL;                                    // vectors per slot
  for (int64_t it0 = (int64_t)blockIdx.x * R; it0 < items; it0 += (int64_t)gridDim.x * R) {
    const int64_t nit = (items - it0 < R) ? (items - it0) : R;
    // 1. HBM -> LDS (threads spread over all slots of the batch), zero padding up to the slot pitch
    copy_slots<true>(cw_in, lds_in, (char*)in, IB, it0, nit, nseg, seg, segp, L);
    {  // pad [len, segp) of every slot with zeros: at most B-1 elements, plus the shortfall of a row's last segment
      uint32_t b_log = 0;
      while ((1u << b_log) < (uint32_t)B) b_log++;               // B is a power of two <= 512
      const uint32_t bl = b_log > 8 ? 8 : b_log;
      const uint32_t tk = threadIdx.x & ((1u << bl) - 1u), tj = threadIdx.x >> bl, nj = kThreads >> bl;
      for (int64_t j = tj; j < nit; j += nj) {
        const int64_t sg = nseg == 1 ? 0 : (it0 + j) % nseg;
        const int64_t len = (L - sg * seg < seg) ? (L - sg * seg) : seg;
        for (int64_t e = len + tk; e < segp; e += (1u << bl)) {
          if (IB == 2) *(uint16_t*)(lds_in + (j * segp + e) * 2) = 0; else *(uint32_t*)(lds_in + (j * segp + e) * 4) = 0u;
        }
      }
    }
    __syncthreads();
    // 2. quantise inside LDS: identical to the aligned flat-stream case
    for (int64_t v = threadIdx.x; v < nit * vps; v += kThreads) {
      const u32x4 raw = *(const u32x4*)(lds_in + v * 16);
      const uint32_t mb = group_max_u32(absmax_bits<DTI>(raw), lpb);
      float x[EPL], y[EPL];
      widen<DTI, EPL>(raw, x);
      if (kFast && __builtin_amdgcn_ballot_w64(!bfp_fast_ok(mb, wl)) == 0ull) {
        const BfpBlockParams p = bfp_block_params<ASYM, true>(mb, wl);
#pragma unroll
        for (int k = 0; k < EPL; k++) y[k] = bfp_q1_fast<FAST == 2, ASYM>(x[k], p);
      } else {
        const BfpBlockParams p = bfp_block_params<ASYM, false>(mb, wl);
        const int64_t j = v / vps, pos = (v % vps) * EPL;
        const int64_t item = it0 + j, row = item / nseg, sg = item % nseg;
        const int64_t e0 = row * L + sg * seg + pos;                 // flat element index (numbers the random draws)
#pragma unroll
        for (int k = 0; k < EPL; k++)
          y[k] = bfp_q1<RND, ASYM>(x[k], p, wl, rounding, rnd_if(stoch, seed, (uint64_t)(e0 + k)));
      }
      store_out<DTO, EPL, false>(lds_out + v * OVB, pack_vec<DTO, EPL>(y));
    }
    __syncthreads();
    // 3. LDS -> HBM (only the real elements)
    copy_slots<false>(cw_out, lds_out, (char*)out, OB, it0, nit, nseg, seg, segp, L);
    __syncthreads();
  }
}

// widest power-of-two access (<= 16 bytes) that every row start honours
static inline int copy_width(const void* p, int64_t row_bytes) {
  uintptr_t a = reinterpret_cast<uintptr_t>(p) | (uintptr_t)row_bytes | 16u;
  int w = 16;
  while (a & (uintptr_t)(w - 1)) w >>= 1;
  return w < 2 ? 2 : w;
}

template <int DTI, int DTO, int RND, bool ASYM>
static int launch_ragged(const void* in, void* out, int64_t rows, int64_t L, int64_t B, int wl, int rounding,
                         uint64_t seed, hipStream_t s) {
  constexpr int IB = Elem<DTI>::bytes, OB = Elem<DTO>::bytes;
  const int64_t seg = L < kRaggedSeg ? L : kRaggedSeg, nseg = (L + seg - 1) / seg, segp = (seg + B - 1) / B * B;
  int64_t R = kRaggedLdsBytes / (segp * (IB + OB));
  if (R < 1) R = 1;
  const size_t lds = (size_t)(R * segp * (IB + OB));
  if (lds > 150 * 1024) return DMXQ_ERR_UNSUPPORTED;
  const int64_t batches = (rows * nseg + R - 1) / R;
  const int grid = (int)(batches < kMaxBlocks ? (batches < 1 ? 1 : batches) : kMaxBlocks);
  const int cwi = copy_width(in, L * IB), cwo = copy_width(out, L * OB);
  const int fast = (RND == DMXQ_ROUND_NEAREST && wl <= 20) ? (bfp_single_rounding_ok<DTI>(wl) ? 2 : 1) : 0;
#define DMXQ_RAG(F_)                                                                                             \
  hipLaunchKernelGGL((bfp_lds_rows_kernel<DTI, DTO, RND, ASYM, F_>), dim3(grid), dim3(kThreads), lds, s, in, out, rows, \
                     L, (int)B, wl, rounding, seed, cwi, cwo)
  constexpr bool in16 = IB == 2;
  if constexpr (RND == kRuntimeRounding) {
    DMXQ_RAG(0);
  } else {
    if (in16 && fast == 2) {
      if constexpr (in16) DMXQ_RAG(2);
    } else {
      DMXQ_RAG(1);
    }
  }
#undef DMXQ_RAG
  return launch_status();
}

}  // namespace dmxq

using namespace dmxq;

// internal entry used by dmxq_bfp_qdq (bfp.hip) for inner == 1 tensors the flat-stream kernel cannot take.
// DMXQ_ERR_UNSUPPORTED = not applicable (caller falls back to the generic kernel).
extern "C" int dmxq_internal_bfp_ragged(const void* in, void* out, int dtype_in, int dtype_out, int64_t rows, int64_t L,
                                        int64_t B, int wl, int rounding, int symmetric, uint64_t seed, void* stream) {
  const int epl = dtype_in == DMXQ_F32 ? 4 : 8;
  if (B < epl || B > 64 * epl || (B & (B - 1)) != 0 || wl > 22) return DMXQ_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const bool asym = !symmetric;
#define DMXQ_DT(I_, O_)                                                                                          \
  if (dtype_in == I_ && dtype_out == O_) {                                                                       \
    if (rounding == DMXQ_ROUND_NEAREST && wl <= 20)                                                              \
      return asym ? launch_ragged<I_, O_, DMXQ_ROUND_NEAREST, true>(in, out, rows, L, B, wl, rounding, seed, s)   \
                  : launch_ragged<I_, O_, DMXQ_ROUND_NEAREST, false>(in, out, rows, L, B, wl, rounding, seed, s); \
    return asym ? launch_ragged<I_, O_, kRuntimeRounding, true>(in, out, rows, L, B, wl, rounding, seed, s)       \
                : launch_ragged<I_, O_, kRuntimeRounding, false>(in, out, rows, L, B, wl, rounding, seed, s);     \
  }
  DMXQ_DT(DMXQ_BF16, DMXQ_BF16)
  DMXQ_DT(DMXQ_F16, DMXQ_F16)
  DMXQ_DT(DMXQ_F32, DMXQ_F32)
  DMXQ_DT(DMXQ_BF16, DMXQ_F32)
  DMXQ_DT(DMXQ_F16, DMXQ_F32)
  DMXQ_DT(DMXQ_F32, DMXQ_BF16)
  DMXQ_DT(DMXQ_F32, DMXQ_F16)
#undef DMXQ_DT
  return DMXQ_ERR_UNSUPPORTED;
}
