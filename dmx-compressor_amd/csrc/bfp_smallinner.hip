// csrc/bfp_smallinner.hip — BFP Q->DQ with blocks along a dim whose INNER extent is small (2 .. 64 elements): conv weights
// [out, in, kh, kw] blocked along `in` (inner = kh * kw = 9, 3, 49 ...), 7 x 7 feature maps blocked along channels.  The elements of a
// block are then `inner` apart, a row of the column kernel (bfp_cols.hip) is only a few bytes long, and one lane per block with
// strided global accesses moves a 64-byte line per element (5.8 % of the roofline on [512, 512, 3, 3], profiles/r01_conv_shapes.txt).
//
// Here the B x inner elements that hold `inner` interleaved blocks -- a SUB-SLAB -- are contiguous in memory, and so is the
// sequence of all sub-slabs of the tensor (L % B == 0).  A workgroup copies G consecutive sub-slabs to the LDS with coalesced
// 16-byte accesses, one lane per block walks its B elements there (stride `inner` halfwords; sub-slabs padded by one dword so
// that the lanes of different sub-slabs fall into different banks), writes the results back in place, and the tile leaves with
// coalesced 16-byte stores.  HBM traffic is 1 read + 1 write per element; arithmetic and results are those of every other BFP
// kernel (bfp_math.hpp).  Scope: 16-bit tensors with the same dtype in and out, nearest rounding, B = 2^k in [8, 256], L % B == 0 --
// or a ragged last block per outer index (attention operands blocked along a sequence of 1500) with one sub-slab per workgroup.
#include "bfp_math.hpp"

namespace dmxq {

constexpr int kSiThreads = 256;
struct SiArgs {
  const void* in; void* out;
  int64_t n_sub;          // sub-slabs in the tensor: outer * (L / B)
  int S8, K, B, G, wl;    // 16-byte vectors per sub-slab (B * K / 8), inner extent, block size, sub-slabs per tile, precision
  int nib, tail;          // ragged L (G == 1 only): sub-slabs per outer index, rows of the last one (0: L % B == 0)
  int64_t LK8;            // ... and 16-byte vectors per outer index (L * K / 8)
};

template <int DT>
__device__ __forceinline__ float si_widen(uint32_t h) {
  if (DT == DMXQ_BF16) return u2f(h << 16);
  return half_lo(h);
}

// FAST: 1 = magic-add double rounding, 2 = single rounding (bfp_single_rounding_ok<DT>(wl))
// LPB: lanes per block (1, or 4 = the lanes of one quad, each walking a quarter of the block; the block maximum through two DPP
// quad permutes): four times as many lanes per sub-slab, so four times smaller tiles for the same lane count -- [512,512,3,3]
// is then 585 workgroups of 8 KB instead of 147 of 32 KB whose lanes each walk 64 elements (8.4 us by rocprofv3)
template <int DT, bool ASYM, int FAST, int LPB>
__global__ __launch_bounds__(kSiThreads) void bfp_smallinner_kernel(const SiArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t si_lds[];
  uint16_t* lds = (uint16_t*)si_lds;
  const int S = a.S8 * 8, SP = S + 2;  // sub-slab length and its padded pitch in halfwords (pitch in dwords is odd + ... : see header)
  const int64_t s0 = (int64_t)blockIdx.x * a.G;
  const int g_cnt = (int)((a.n_sub - s0 < a.G) ? (a.n_sub - s0) : a.G);
  int nv = g_cnt * a.S8, rows_here = a.B;   // rows (block members) of this tile's sub-slabs
  int64_t v_start = s0 * a.S8;
  if (a.tail) {  // ragged L: one sub-slab per tile, the last of every outer index is shorter (torch.split's ragged block)
    const int64_t o = s0 / a.nib;
    const int ib = (int)(s0 - o * a.nib);
    v_start = o * a.LK8 + (int64_t)ib * a.S8;
    if (ib == a.nib - 1) { rows_here = a.tail; nv = a.tail * a.K / 8; }
  }
  const u32x4* src = (const u32x4*)a.in + v_start;
  u32x4* dst = (u32x4*)a.out + v_start;
  // the tile's global loads in batches of 4 per lane, all of a batch issued before its LDS writes (one load at a time exposed a
  // full HBM round trip per iteration of the copy loop)
  for (int v0 = threadIdx.x; v0 < nv; v0 += 4 * kSiThreads) {
    u32x4 raw[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int v = v0 + j * kSiThreads;
      raw[j] = __builtin_nontemporal_load(src + (v < nv ? v : v0));  // clamped: unconditional, back to back
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int v = v0 + j * kSiThreads;
      if (v < nv) {
        const int g = v / a.S8, off = (v - g * a.S8) * 8;
        uint32_t* p = si_lds + (g * SP + off) / 2;  // g * SP + off is even
        p[0] = raw[j].x; p[1] = raw[j].y; p[2] = raw[j].z; p[3] = raw[j].w;
      }
    }
  }
  __syncthreads();
  const int per = a.B / LPB;  // elements of a block per lane (a multiple of 8)
  // (the trip count is rounded up to whole quads so that the DPP exchange below always has its four lanes active)
  for (int t = threadIdx.x; t < g_cnt * a.K * LPB; t += kSiThreads) {
    const int bidx = t / LPB, part = t - bidx * LPB;
    const int g = bidx / a.K, k = bidx - g * a.K;
    uint16_t* blk = lds + g * SP + k + part * per * a.K;
    const int cnt = rows_here - part * per < per ? (rows_here - part * per > 0 ? rows_here - part * per : 0) : per;  // rows of this lane
    // 8 elements at a time: the 8 LDS reads are independent and in flight together (one by one each would expose its latency)
    uint32_t m16 = 0u;
    for (int i = 0; i < cnt; i += 8) {
      uint32_t h[8];
#pragma unroll
      for (int j = 0; j < 8; j++) h[j] = i + j < cnt ? blk[(i + j) * a.K] : 0u;
#pragma unroll
      for (int j = 0; j < 8; j++) m16 = max(m16, h[j] & 0x7FFFu);  // abs bit patterns order like the values
    }
    if (LPB == 4) {
      m16 = max(m16, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m16, 0xB1, 0xF, 0xF, false));  // quad_perm 1,0,3,2
      m16 = max(m16, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m16, 0x4E, 0xF, 0xF, false));  // quad_perm 2,3,0,1
    }
    const uint32_t mb = f2u(si_widen<DT>(m16));
    if (bfp_fast_ok(mb, a.wl)) {
      const BfpBlockParams p = bfp_block_params<ASYM, true>(mb, a.wl);
      for (int i = 0; i < cnt; i += 8) {
        uint32_t h[8];
#pragma unroll
        for (int j = 0; j < 8; j++) h[j] = i + j < cnt ? blk[(i + j) * a.K] : 0u;
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const float y = bfp_q1_fast<FAST == 2, ASYM>(si_widen<DT>(h[j]), p);
          if (i + j < cnt) blk[(i + j) * a.K] = (uint16_t)(pack2<DT>(y, 0.0f) & 0xFFFFu);
        }
      }
    } else {
      const BfpBlockParams p = bfp_block_params<ASYM, false>(mb, a.wl);
      for (int i = 0; i < cnt; i++) {
        const float y = bfp_q1<DMXQ_ROUND_NEAREST, ASYM>(si_widen<DT>(blk[i * a.K]), p, a.wl, DMXQ_ROUND_NEAREST, 0u);
        blk[i * a.K] = (uint16_t)(pack2<DT>(y, 0.0f) & 0xFFFFu);
      }
    }
  }
  __syncthreads();
  for (int v = threadIdx.x; v < nv; v += kSiThreads) {
    const int g = v / a.S8, off = (v - g * a.S8) * 8;
    const uint32_t* p = si_lds + (g * SP + off) / 2;
    __builtin_nontemporal_store(u32x4{p[0], p[1], p[2], p[3]}, dst + v);
  }
}

}  // namespace dmxq

using namespace dmxq;

// internal entry used by dmxq_bfp_qdq (bfp.hip) before the column kernel.  DMXQ_ERR_UNSUPPORTED = not applicable.
extern "C" int dmxq_internal_bfp_smallinner(const void* in, void* out, int dtype_in, int dtype_out, int64_t outer, int64_t L,
                                            int64_t inner, int64_t B, int wl, int rounding, int symmetric, void* stream) {
  if (dtype_in != dtype_out || (dtype_in != DMXQ_BF16 && dtype_in != DMXQ_F16) || rounding != DMXQ_ROUND_NEAREST) return DMXQ_ERR_UNSUPPORTED;
  // inner = 64 (rows of 8 vectors) is the column kernel's: measured 10.1 us vs 13.5 us here on [8,12,1500,64] along the sequence
  if (inner < 2 || inner > 63 || L < B || (B & (B - 1)) != 0 || B < 8 || B > 256 || wl > 20) return DMXQ_ERR_UNSUPPORTED;
  const int64_t tail = L % B;  // ragged last block per outer index: supported when a tile is ONE sub-slab and its length is whole vectors
  if (tail && (tail * inner) % 8 != 0) return DMXQ_ERR_UNSUPPORTED;
  if (!aligned16(in) || !aligned16(out)) return DMXQ_ERR_UNSUPPORTED;
  const int64_t S = B * inner;                       // halfwords per sub-slab (a multiple of 8)
  const int64_t lds_cap = 48 * 1024;
  const int lpb = B >= 32 ? 4 : 1;                // a quad per block when each of its lanes still gets >= 8 elements
  int64_t G = kSiThreads / (inner * lpb);            // G * inner * lpb <= 256 lanes: one pass over the tile's blocks
  if (G < 1) G = 1;
  const int64_t fit = lds_cap / ((S + 2) * 2);
  if (fit < 1) return DMXQ_ERR_UNSUPPORTED;
  if (G > fit) G = fit;
  if (tail) { if (inner < 16 || (L * inner) % 8 != 0) return DMXQ_ERR_UNSUPPORTED; G = 1; }
  const int64_t nib = (L + B - 1) / B;
  const int64_t n_sub = outer * nib;
  const int64_t tiles = (n_sub + G - 1) / G;
  if (tiles > 0x7FFFFFFF || S / 8 > 0x7FFFFFF) return DMXQ_ERR_UNSUPPORTED;
  if (nib > 0x7FFFFFFF) return DMXQ_ERR_UNSUPPORTED;
  const SiArgs a{in, out, n_sub, (int)(S / 8), (int)inner, (int)B, (int)G, wl, (int)nib, (int)tail, L * inner / 8};
  const size_t lds = (size_t)(G * (S + 2) * 2);
  hipStream_t s = (hipStream_t)stream;
  const bool asym = !symmetric;
#define DMXQ_SI2(D_, A_, F_)                                                                                                          \
  do {                                                                                                                                \
    if (lpb == 4) DMXQ_LAUNCH((bfp_smallinner_kernel<D_, A_, F_, 4>), dim3((unsigned)tiles), dim3(kSiThreads), lds, s, a);            \
    else DMXQ_LAUNCH((bfp_smallinner_kernel<D_, A_, F_, 1>), dim3((unsigned)tiles), dim3(kSiThreads), lds, s, a);                     \
  } while (0)
#define DMXQ_SI(D_)                                                                                                                   \
  do {                                                                                                                                \
    const bool single = bfp_single_rounding_ok<D_>(wl);                                                                               \
    if (single) { if (asym) DMXQ_SI2(D_, true, 2); else DMXQ_SI2(D_, false, 2); }                                                     \
    else { if (asym) DMXQ_SI2(D_, true, 1); else DMXQ_SI2(D_, false, 1); }                                                            \
  } while (0)
  if (dtype_in == DMXQ_BF16) DMXQ_SI(DMXQ_BF16); else DMXQ_SI(DMXQ_F16);
#undef DMXQ_SI
#undef DMXQ_SI2
  return launch_status();
}
