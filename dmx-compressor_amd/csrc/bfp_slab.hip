// csrc/bfp_slab.hip — BFP Q->DQ with blocks along a NON-contiguous dim whose rows (the contiguous inner extent) are NOT whole 128-byte
// lines: conv activations [N, C, H, W] blocked along C (modeling/nn/torch_modules.py:582-585; numerical/format.py:304-343 transposes,
// splits and transposes back) on 14 x 14 / 28 x 28 maps -- rows of 392 / 1568 bytes.  Round 6: the register-tiled
// column kernel (bfp_cols.hip) reads such tensors row piece by row piece, every piece starting and ending inside a line that the
// neighbouring piece -- another wave, usually another workgroup on another XCD with its own L2 -- also touches: 50 % of the roofline on
// [256, 1024, 14, 14] and [64, 512, 28, 28] where whole-line rows ([64, 256, 56, 56]) run at 69 %.
//
// Here the B rows x inner elements that hold `inner` interleaved blocks -- a SLAB, contiguous in memory and (B a multiple of 64 elements
// of 2 bytes) a whole number of lines -- travel through the LDS as ONE contiguous run per tile, read and written with aligned 16-byte
// accesses whatever the row length; no line is shared between tiles.  In the LDS a lane owns a PAIR of adjacent columns (one dword per
// row: two blocks) and every Q-th row of it; the Q lanes of a pair are neighbours (block maxima through quad DPP permutes).  Row pitch
// = 32 / Q (mod 32) dwords: the 32 lanes of a half-wave -- 32 / Q column pairs x Q rows -- then fall into 32 different banks for the b32
// reads and writes.  Slabs up to 64 KiB: workgroups of 256 lanes, several per CU; up to 150 KiB (28 x 28 maps, B = 64: 98 KiB): ONE
// workgroup of 1024 lanes per CU.  A tile's loads are all in flight at once (NV per lane).  The 256-lane form runs one tile per
// workgroup; the 1024-lane form is PERSISTENT (the grid is one resident round) and holds the NEXT tile's loads in registers while it
// works on the current one in the LDS (the launcher's `persistent`: measured both ways, profiles/r06_slab_ab4 / ab6).
// HBM traffic is one read and one write per element; arithmetic and results are those of every other BFP kernel (bfp_math.hpp).
// Scope: 16-bit tensors, same dtype in and out, nearest rounding, B = 2^k in [8, 256], even inner; in place allowed (a tile is read
// completely before it is written, tiles are disjoint).  (A SEGMENTED form -- tiles of B rows x a column range for slabs that do not
// fit -- was built and measured: its row segments split lines between tiles again, 39-47 % on 28 x 28 / 56 x 56 against the column
// kernel's 51 / 69 %; profiles/r06_slab_ab*.txt.  Not kept.)
#include <mutex>
#include <vector>

#include "bfp_math.hpp"

namespace dmxq {

struct SlabArgs {
  const void* in; void* out;
  int64_t L, inner;          // block dim, contiguous dim (elements)
  int B, wl, logQ;           // block size, precision, log2 of the lanes per column pair
  int nblk;                  // blocks per outer index
  int pitch;                 // LDS row pitch in dwords
  FastDiv31 f_nblk;          // tile -> (outer, block)
  FastDiv31 f_inner;         // element of the slab -> row
};

template <int DT>
__device__ __forceinline__ float slab_lo(uint32_t w) { return DT == DMXQ_BF16 ? u2f(w << 16) : half_lo(w); }
template <int DT>
__device__ __forceinline__ float slab_hi(uint32_t w) { return DT == DMXQ_BF16 ? u2f(w & 0xFFFF0000u) : half_hi(w); }

// one tile's place in the tensor
struct SlabTile { int rows_here, nv; int64_t e_base; };
__device__ __forceinline__ SlabTile slab_tile(const SlabArgs& a, uint32_t tile) {
  SlabTile t;
  const uint32_t o = a.f_nblk.div(tile), blk = tile - o * (uint32_t)a.nblk;
  const int rows_left = (int)(a.L - (int64_t)blk * a.B);
  t.rows_here = rows_left < a.B ? rows_left : a.B;
  t.e_base = ((int64_t)o * a.L + (int64_t)blk * a.B) * a.inner;   // first element of the tile
  t.nv = (t.rows_here * (int)a.inner) >> 3;
  return t;
}

// T: lanes per workgroup.  PER: rows per lane (B = PER << logQ).  FAST: 1 = magic-add double rounding, 2 = single rounding
// (bfp_single_rounding_ok<DT>(wl)).  NV: 16-byte vectors of a tile per lane (tile <= NV * T vectors).  PERSISTENT workgroups:
//     registers -> LDS | barrier | issue the next tile's loads | blocks in the LDS | barrier | LDS -> global
template <int DT, bool ASYM, int FAST, int PER, int NV, int T>
__global__ __launch_bounds__(T) void bfp_slab_kernel(const SlabArgs a, uint32_t n_tiles) {
  extern __shared__ __attribute__((aligned(16))) uint32_t slab_lds[];
  const int pitch = a.pitch;
  const int wd = (int)(a.inner >> 1);   // dwords per row
  u32x4 raw[NV];
  auto issue = [&](const SlabTile& t) __attribute__((always_inline)) {
    const u32x4* src = (const u32x4*)((const uint16_t*)a.in + t.e_base);
#pragma unroll
    for (int j = 0; j < NV; j++) {
      int v = (int)threadIdx.x + j * T;
      v = v < t.nv ? v : t.nv - 1;   // clamped: unconditional loads, back to back
      raw[j] = __builtin_nontemporal_load(src + v);
    }
  };
  uint32_t tile = blockIdx.x;
  if (tile >= n_tiles) return;
  SlabTile t = slab_tile(a, tile);
  issue(t);
  for (;;) {
    // ---- registers -> LDS (a vector may straddle two rows: dword by dword)
#pragma unroll
    for (int j = 0; j < NV; j++) {
      const int v = (int)threadIdx.x + j * T;
      if (v < t.nv) {
        const uint32_t e = (uint32_t)v << 3;
        int r = (int)a.f_inner.div(e);
        int cd = (int)(e - (uint32_t)r * (uint32_t)a.inner) >> 1;   // dword column in the row
#pragma unroll
        for (int k = 0; k < 4; k++) {
          if (cd >= wd) { cd -= wd; r++; }
          slab_lds[r * pitch + cd] = raw[j][k];
          cd++;
        }
      }
    }
    __syncthreads();
    // ---- the next tile's loads travel while this one is worked on
    const uint32_t next = tile + gridDim.x;
    const bool more = next < n_tiles;   // (workgroup-uniform)
    const SlabTile cur = t;
    if (more) {
      t = slab_tile(a, next);
      issue(t);
    }
    // ---- the blocks: lane (pair, q) owns rows q, q + Q, q + 2 Q ... of column pair `pair`
    const int Q = 1 << a.logQ;
    const int rows_here = cur.rows_here;
    for (int it = threadIdx.x; it < (wd << a.logQ); it += T) {
      const int pair = it >> a.logQ, q = it & (Q - 1);
      uint32_t* col = slab_lds + q * pitch + pair;
      const int step = pitch << a.logQ;
      uint32_t h[PER];
      uint32_t m_lo = 0u, m_hi = 0u;
#pragma unroll
      for (int i = 0; i < PER; i++) h[i] = (q + (i << a.logQ)) < rows_here ? col[i * step] : 0u;
#pragma unroll
      for (int i = 0; i < PER; i++) {   // abs bit patterns order like the values
        m_lo = max(m_lo, h[i] & 0x7FFFu);
        m_hi = max(m_hi, (h[i] >> 16) & 0x7FFFu);
      }
      if (a.logQ >= 1) {
        m_lo = max(m_lo, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m_lo, 0xB1, 0xF, 0xF, false));  // quad_perm 1,0,3,2
        m_hi = max(m_hi, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m_hi, 0xB1, 0xF, 0xF, false));
      }
      if (a.logQ >= 2) {
        m_lo = max(m_lo, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m_lo, 0x4E, 0xF, 0xF, false));  // quad_perm 2,3,0,1
        m_hi = max(m_hi, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m_hi, 0x4E, 0xF, 0xF, false));
      }
      const uint32_t mb_lo = f2u(slab_lo<DT>(m_lo)), mb_hi = f2u(slab_lo<DT>(m_hi));
      const bool fast = bfp_fast_ok(mb_lo, a.wl) && bfp_fast_ok(mb_hi, a.wl);
      if (__builtin_amdgcn_ballot_w64(!fast) == 0ull) {   // wave-uniform
        const BfpBlockParams p0 = bfp_block_params<ASYM, true>(mb_lo, a.wl), p1 = bfp_block_params<ASYM, true>(mb_hi, a.wl);
#pragma unroll
        for (int i = 0; i < PER; i++) {
          const float y0 = bfp_q1_fast<FAST == 2, ASYM>(slab_lo<DT>(h[i]), p0);
          const float y1 = bfp_q1_fast<FAST == 2, ASYM>(slab_hi<DT>(h[i]), p1);
          if ((q + (i << a.logQ)) < rows_here) col[i * step] = pack2<DT>(y0, y1);
        }
      } else {
        const BfpBlockParams p0 = bfp_block_params<ASYM, false>(mb_lo, a.wl), p1 = bfp_block_params<ASYM, false>(mb_hi, a.wl);
#pragma unroll
        for (int i = 0; i < PER; i++) {   // (cold: blocks with a denormal / NaN / huge maximum; unrolled so that h[] stays in registers)
          const float y0 = bfp_q1<DMXQ_ROUND_NEAREST, ASYM>(slab_lo<DT>(h[i]), p0, a.wl, DMXQ_ROUND_NEAREST, 0u);
          const float y1 = bfp_q1<DMXQ_ROUND_NEAREST, ASYM>(slab_hi<DT>(h[i]), p1, a.wl, DMXQ_ROUND_NEAREST, 0u);
          if ((q + (i << a.logQ)) < rows_here) col[i * step] = pack2<DT>(y0, y1);
        }
      }
    }
    __syncthreads();
    // ---- LDS -> global.  (A FIXED number of predicated stores: the wait for the next tile's loads at the top of the loop is then "all but
    // the last NV vector-memory operations" -- vmcnt counts loads and stores together on gfx9, and after a loop of unknown length the
    // compiler could only wait for everything, i.e. for the write acknowledgements of this tile.)
    u32x4* dst = (u32x4*)((uint16_t*)a.out + cur.e_base);
#pragma unroll
    for (int j = 0; j < NV; j++) {
      const int v = (int)threadIdx.x + j * T;
      if (v < cur.nv) {
        const uint32_t e = (uint32_t)v << 3;
        int r = (int)a.f_inner.div(e);
        int cd = (int)(e - (uint32_t)r * (uint32_t)a.inner) >> 1;
        u32x4 w;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          if (cd >= wd) { cd -= wd; r++; }
          w[k] = slab_lds[r * pitch + cd];
          cd++;
        }
        __builtin_nontemporal_store(w, dst + v);
      }
    }
    if (!more) break;
    tile = next;
    __syncthreads();   // (the LDS is rewritten by the next tile: every lane's reads above must be done)
  }
}

}  // namespace dmxq

using namespace dmxq;

// workgroups of `kernel` a CU holds at once with `lds` bytes of dynamic LDS (hipOccupancyMaxActiveBlocksPerMultiprocessor), remembered per
// (kernel, LDS size): the persistent grid is exactly one resident round.  Slabs beyond 64 KiB need the kernel's dynamic-LDS limit raised.
static int slab_resident(const void* kernel, int threads, size_t lds) {
  struct Key { const void* k; size_t l; int n; };
  static std::mutex mu;
  static std::vector<Key> seen;
  std::lock_guard<std::mutex> g(mu);
  for (const Key& e : seen)
    if (e.k == kernel && e.l == lds) return e.n;
  int n = 0;
  if (lds > 64 * 1024 && hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
    (void)hipGetLastError();
    n = -1;   // (this slab size cannot run here: the caller falls back)
  } else if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, threads, lds) != hipSuccess || n < 1) {
    (void)hipGetLastError();
    n = threads >= 1024 ? 1 : 2;
  }
  if (n > 8) n = 8;
  seen.push_back(Key{kernel, lds, n});
  return n;
}

// internal entry used by dmxq_bfp_qdq (bfp.hip) before the column kernel.  DMXQ_ERR_UNSUPPORTED = not applicable.
extern "C" int dmxq_internal_bfp_slab(const void* in, void* out, int dtype_in, int dtype_out, int64_t outer, int64_t L, int64_t inner,
                                      int64_t B, int wl, int rounding, int symmetric, void* stream) {
  if (dtype_in != dtype_out || (dtype_in != DMXQ_BF16 && dtype_in != DMXQ_F16) || rounding != DMXQ_ROUND_NEAREST) return DMXQ_ERR_UNSUPPORTED;
  if (inner < 64 || (inner & 1) || inner > (1 << 20) || L < 1 || L > (1 << 30) || (B & (B - 1)) != 0 || B < 8 || B > 256 || wl > 20 || wl < 2)
    return DMXQ_ERR_UNSUPPORTED;
  if (!aligned16(in) || !aligned16(out)) return DMXQ_ERR_UNSUPPORTED;
  // fewer rows than half a block (RGB input of a first conv layer: L = 3): one ragged block per column whatever the nominal size
  while (B > 8 && L <= B / 2) B /= 2;
  const int per = B >= 128 ? (int)(B / 4) : (B >= 64 ? 16 : 8);
  const int Q = (int)(B / per);
  const int logQ = Q == 4 ? 2 : (Q == 2 ? 1 : 0);
  int64_t pitch = (inner / 2 + 3) & ~(int64_t)3;   // dwords: >= inner / 2, a multiple of 4, = 32 / Q (mod 32) for Q > 1 (see the header)
  if (Q > 1) while (pitch % 32 != 32 / Q) pitch += 4;
  const int64_t nblk = (L + B - 1) / B, tail = L % B;
  const size_t lds = (size_t)(B * pitch * 4);
  if (lds > 150 * 1024) return DMXQ_ERR_UNSUPPORTED;
  // whole slabs as ONE contiguous run each: every tile must start at a 16-byte boundary and hold whole vectors
  if ((B * inner) % 8 != 0 || (outer > 1 && (L * inner) % 8 != 0) || (tail * inner) % 8 != 0) return DMXQ_ERR_UNSUPPORTED;
  const int64_t tiles = outer * nblk;
  if (tiles < 1 || tiles > 0x7FFFFFFF) return DMXQ_ERR_UNSUPPORTED;
  SlabArgs a{in, out, L, inner, (int)B, wl, logQ, (int)nblk, (int)pitch, make_fastdiv31(nblk), make_fastdiv31(inner)};
  hipStream_t s = (hipStream_t)stream;
  const bool asym = !symmetric;
  const bool big = lds > 64 * 1024;                 // one workgroup of 1024 lanes per CU
  const int threads = big ? 1024 : 256;
  const int nvl = (int)(((B * inner) / 8 + threads - 1) / threads);   // vectors of a full tile per lane
  if (nvl > 16 || (big && (nvl > 8 || per == 64))) return DMXQ_ERR_UNSUPPORTED;   // (B = 256 at 1024 lanes would spill: 128 VGPRs)
  // persistent workgroups only where ONE workgroup fits a CU (the 1024-lane form: 54.7 % against 48.2 % with one tile per workgroup on
  // [64,512,28,28]); with several workgroups per CU their phases overlap by themselves and one tile per workgroup is as fast (14 x 14
  // maps: 65.6 vs 66.0 %) or faster ([8,32,2048,128] along the sequence, 16 KiB slabs: 76.3 vs 67.7 %) -- profiles/r06_slab_ab4*.txt.
  // -DDMXQ_SLAB_PERSIST / env DMXQ_SLAB_PERSIST=0|1 force either (A/B runs).
  static const int forced_persist = [] { const char* e = getenv("DMXQ_SLAB_PERSIST"); return e ? atoi(e) : -1; }();
  const bool persistent = forced_persist >= 0 ? forced_persist != 0 : big;
  int rc_unsupported = 0;
#define DMXQ_SL5(K_, T_)                                                                                                            \
  do {                                                                                                                              \
    int per_cu = slab_resident((const void*)(K_), T_, lds);                                                                         \
    if (per_cu < 1) { rc_unsupported = 1; break; }                                                                                  \
    int64_t grid = (int64_t)per_cu * plan_cus();                                                                                    \
    if (grid > tiles || !persistent) grid = tiles;                                                                                  \
    DMXQ_LAUNCH((K_), dim3((unsigned)grid), dim3(T_), lds, s, a, (uint32_t)tiles);                                                  \
  } while (0)
#define DMXQ_SL3(D_, A_, F_, P_)                                                                                                    \
  do {                                                                                                                              \
    if (big) { if constexpr (P_ != 64) DMXQ_SL5((bfp_slab_kernel<D_, A_, F_, P_, 8, 1024>), 1024); }                                \
    else if (nvl <= 4) DMXQ_SL5((bfp_slab_kernel<D_, A_, F_, P_, 4, 256>), 256);                                                    \
    else if (nvl <= 8) DMXQ_SL5((bfp_slab_kernel<D_, A_, F_, P_, 8, 256>), 256);                                                    \
    else DMXQ_SL5((bfp_slab_kernel<D_, A_, F_, P_, 16, 256>), 256);                                                                 \
  } while (0)
#define DMXQ_SL2(D_, A_, F_)                                                                                                        \
  do {                                                                                                                              \
    if (per == 8) DMXQ_SL3(D_, A_, F_, 8);                                                                                          \
    else if (per == 16) DMXQ_SL3(D_, A_, F_, 16);                                                                                   \
    else if (per == 32) DMXQ_SL3(D_, A_, F_, 32);                                                                                   \
    else DMXQ_SL3(D_, A_, F_, 64);                                                                                                  \
  } while (0)
#define DMXQ_SL(D_)                                                                                                                 \
  do {                                                                                                                              \
    const bool single = bfp_single_rounding_ok<D_>(wl);                                                                             \
    if (single) { if (asym) DMXQ_SL2(D_, true, 2); else DMXQ_SL2(D_, false, 2); }                                                   \
    else { if (asym) DMXQ_SL2(D_, true, 1); else DMXQ_SL2(D_, false, 1); }                                                          \
  } while (0)
  if (dtype_in == DMXQ_BF16) DMXQ_SL(DMXQ_BF16); else DMXQ_SL(DMXQ_F16);
#undef DMXQ_SL
#undef DMXQ_SL2
#undef DMXQ_SL3
#undef DMXQ_SL5
  if (rc_unsupported) return DMXQ_ERR_UNSUPPORTED;
  return launch_status();
}
