// csrc/bfp_urows.hip — BFP Q->DQ along the contiguous dim for rows the flat-stream kernel (bfp_rows.hpp) cannot take:
// L % B != 0 (torch.split's ragged last block, numerical/format.py:324-326 — LeNet fc in = 400, attention rows of
// 1500, ...) and/or rows or bases that are not 16-byte aligned — WITHOUT staging through LDS.
//
// gfx950 serves 16-byte global accesses at any element alignment at 93-97 % of the aligned rate (measured:
// tools/unaligned_access.hip, 2- / 4- / 8-byte offsets), so a row is read directly as ceil(L / EPL) lane-vectors starting
// at its first element, whatever the row pitch.  The vectors of all rows are numbered in a VIRTUAL flat space in which
// every row is padded to a multiple of the block's lane count (nvrp vectors per row): a block is then B / EPL adjacent
// lanes of one wave exactly as in the aligned kernel (block max by DPP, same arithmetic, bfp_math.hpp), padding
// vectors read nothing and count as zeros (= the reference's shorter last block), and the one partial vector at the
// end of a row (L % EPL elements) is read and written in 8 / 4 / 2-byte pieces so that nothing outside the row is
// touched (element by element: one lane per row).  HBM traffic is 1 read + 1 write per element.
#include "bfp_math.hpp"

namespace dmxq {

typedef u32x4 u32x4_u __attribute__((aligned(2)));
typedef u32x2 u32x2_u __attribute__((aligned(2)));

// the first t elements (t < EPL, wave-uniform) of the lane-vector at element offset e, rest zero: element-wise loads,
// packed into the raw layout with compile-time positions (only the one tail lane of a row runs this)
template <int DTI, int EPL>
__device__ __forceinline__ u32x4 tail_load(const void* in, int64_t e, int t) {
  uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int k = 0; k < EPL; k++) {
    if (k < t) {
      if (Elem<DTI>::bytes == 4) w[k] = ((const uint32_t*)in)[e + k];
      else w[k / 2] |= (uint32_t)((const uint16_t*)in)[e + k] << (16 * (k & 1));
    }
  }
  return u32x4{w[0], w[1], w[2], w[3]};
}

// EPL consecutive outputs at element offset e, any alignment: 16-byte accesses (one, or two for fp32 outputs of a 16-bit
// lane-vector), 8 bytes for 16-bit outputs of an fp32 lane-vector
template <int DTO, int EPL>
__device__ __forceinline__ void store_unaligned(void* out, int64_t e, const float (&y)[EPL]) {
  char* p = (char*)out + e * Elem<DTO>::bytes;
  if (DTO == DMXQ_F32) {
#pragma unroll
    for (int k = 0; k < EPL; k += 4)
      __builtin_nontemporal_store(u32x4{f2u(y[k]), f2u(y[k + 1]), f2u(y[k + 2]), f2u(y[k + 3])}, (u32x4_u*)(p + 4 * k));
  } else if (EPL == 8) {
    __builtin_nontemporal_store(u32x4{pack2<DTO>(y[0], y[1]), pack2<DTO>(y[2], y[3]), pack2<DTO>(y[4], y[5]), pack2<DTO>(y[6], y[7])},
                                (u32x4_u*)p);
  } else {
    __builtin_nontemporal_store(u32x2{pack2<DTO>(y[0], y[1]), pack2<DTO>(y[2], y[3])}, (u32x2_u*)p);
  }
}

template <int DTI, int DTO, int RND, bool ASYM, int FAST, int UNROLL>
__global__ __launch_bounds__(kThreads) void bfp_urows_kernel(const void* __restrict__ in, void* __restrict__ out,
                                                            int64_t rows, int64_t L, int nvr /*vectors per row*/,
                                                            int nvrp /*... padded to whole blocks*/, int tail, int back,
                                                            int lpb_arg, int wl, int rounding, uint64_t seed) {
  constexpr int EB = Elem<DTI>::bytes, EPL = 16 / EB;
  constexpr bool kFast = FAST != 0 && RND == DMXQ_ROUND_NEAREST;
  constexpr int64_t TILE = (int64_t)kThreads * UNROLL;
  const bool stoch = (RND == kRuntimeRounding) && rounding == DMXQ_ROUND_STOCHASTIC;
  const int lpb = __builtin_amdgcn_readfirstlane(lpb_arg);
  const int64_t total = rows * nvrp;
  const int64_t n_tiles = (total + TILE - 1) / TILE;
  const int dq = kThreads / nvrp, dr = kThreads % nvrp;  // (row, vector) step between a lane's consecutive vectors
  const int last = nvr - 1;
  // the last vector of a row is partial.  back > 0: it is read and written as the 16 bytes that END at the row end instead (its
  // first `back` elements repeat the previous lane's last ones -- same block, same values, a benign duplicate store), so the
  // row has no partial access at all; the host allows it when the ragged block holds a whole vector and out != in
  const bool has_tail = tail < EPL && back == 0;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t u0 = tile * TILE + threadIdx.x;
    int64_t r = total < ((int64_t)1 << 31) ? (int64_t)((uint32_t)u0 / (uint32_t)nvrp) : u0 / nvrp;
    int v = (int)(u0 - r * nvrp);
    u32x4 raw[UNROLL];
    int64_t eoff[UNROLL];  // element offset of the vector; -1: nothing to read or write (padding, past the end)
    bool part[UNROLL];     // the partial last vector of its row
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const bool real = r < rows && v < nvr;
      part[u] = real && has_tail && v == last;
      eoff[u] = real ? r * L + (int64_t)v * EPL - (v == last ? back : 0) : -1;
      // whole vectors: one 16-byte access at any alignment; everything else reads vector 0 of the tensor (in bounds,
      // discarded) so that the loads stay unconditional and are issued back to back
      const bool whole = real && !part[u];
      raw[u] = __builtin_nontemporal_load((const u32x4_u*)((const char*)in + (whole ? eoff[u] : 0) * EB));
      if (!whole) raw[u] = u32x4{0u, 0u, 0u, 0u};
      v += dr; r += dq;
      if (v >= nvrp) { v -= nvrp; r += 1; }
    }
    if (has_tail) {
#pragma unroll
      for (int u = 0; u < UNROLL; u++)
        if (part[u]) raw[u] = tail_load<DTI, EPL>(in, eoff[u], tail);
    }
    float y[UNROLL][EPL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const uint32_t mb = group_max_u32(absmax_bits<DTI>(raw[u]), lpb);
      float x[EPL];
      widen<DTI, EPL>(raw[u], x);
      if constexpr (kFast) {
        // magic-add form for every lane (straight-line), literal redo of the blocks it does not cover behind one cold branch
        const bool ok = bfp_fast_ok(mb, wl);
        {
          const BfpBlockParams p = bfp_block_params<ASYM, true>(mb, wl);
#pragma unroll
          for (int k = 0; k < EPL; k++) y[u][k] = bfp_q1_fast<FAST == 2, ASYM>(x[k], p);
        }
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(!ok) != 0ull, 0)) {
          if (!ok) {
            const BfpBlockParams p = bfp_block_params<ASYM, false>(mb, wl);
#pragma unroll
            for (int k = 0; k < EPL; k++) y[u][k] = bfp_q1<RND, ASYM>(x[k], p, wl, rounding, 0u);
          }
        }
      } else {
        const BfpBlockParams p = bfp_block_params<ASYM, false>(mb, wl);
#pragma unroll
        for (int k = 0; k < EPL; k++)
          y[u][k] = bfp_q1<RND, ASYM>(x[k], p, wl, rounding, bfp_rnd_if(stoch, seed, (uint64_t)(eoff[u] + k)));
      }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; u++)
      if (eoff[u] >= 0 && !part[u]) store_unaligned<DTO, EPL>(out, eoff[u], y[u]);
    if (has_tail) {
#pragma unroll
      for (int u = 0; u < UNROLL; u++)
        if (part[u]) {
#pragma unroll
          for (int k = 0; k < EPL; k++)
            if (k < tail) store1<DTO>(out, eoff[u] + k, y[u][k]);
        }
    }
  }
}

template <int DTI, int DTO, int RND, bool ASYM>
static int launch_urows(const void* in, void* out, int64_t rows, int64_t L, int64_t B, int wl, int rounding,
                        uint64_t seed, hipStream_t s) {
  constexpr int EPL = 16 / Elem<DTI>::bytes;
  // vectors per lane: 3 (2..4 are within 3 % of each other on rows of 1500 / 4088 / 4100 at 100+ MB, 8 is 10 % slower), and 6 for
  // tensors of up to 24 MiB of lane-vectors -- the rule of every streaming kernel here, more of a small tensor in flight per
  // workgroup: [8, 12, 64, 1500] bf16 (Whisper's attention probabilities, 37 MB moved) 9.73 -> 9.03 us, [2048, 1500] 10.4 -> 10.0,
  // while 18000 x 1500 LOSES 5 % with 6 (same-lease A/B of four builds, profiles/r06_ab_stragglers.txt).  -DDMXQ_EXP_UROWS_UNROLL=N: A/B builds.
#ifdef DMXQ_EXP_UROWS_UNROLL
  constexpr int U_BIG = DMXQ_EXP_UROWS_UNROLL, U_SMALL = DMXQ_EXP_UROWS_UNROLL;
#else
  constexpr int U_BIG = 3, U_SMALL = 6;
#endif
  const int lpb = (int)(B / EPL);
  const int64_t nvr = (L + EPL - 1) / EPL, nvrp = (nvr + lpb - 1) / lpb * lpb;
  const int tail = (int)(L - (nvr - 1) * EPL);  // 1..EPL elements in the last vector of a row
  const int back = (tail < EPL && L % B >= EPL && in != out) ? EPL - tail : 0;
  const bool small = plan_norm(rows * nvrp) <= ((int64_t)3 << 19);
  const int unroll = small ? U_SMALL : U_BIG;
  const int64_t tiles = (rows * nvrp + (int64_t)kThreads * unroll - 1) / ((int64_t)kThreads * unroll);
  const int grid = (int)(tiles < (1 << 20) ? tiles : (1 << 20));
  const int fast = (RND == DMXQ_ROUND_NEAREST && wl <= 20) ? (bfp_single_rounding_ok<DTI>(wl) ? 2 : 1) : 0;
#define DMXQ_UR1(F_, U_)                                                                                           \
  DMXQ_LAUNCH((bfp_urows_kernel<DTI, DTO, RND, ASYM, F_, U_>), dim3(grid), dim3(kThreads), 0, s, in, out,          \
                     rows, L, (int)nvr, (int)nvrp, tail, back, lpb, wl, rounding, seed)
#define DMXQ_UR(F_) do { if (small) DMXQ_UR1(F_, U_SMALL); else DMXQ_UR1(F_, U_BIG); } while (0)
  constexpr bool in16 = Elem<DTI>::bytes == 2;
  if constexpr (RND == kRuntimeRounding) {
    DMXQ_UR(0);
  } else {
    if (in16 && fast == 2) {
      if constexpr (in16) DMXQ_UR(2);
    } else {
      DMXQ_UR(1);
    }
  }
#undef DMXQ_UR
#undef DMXQ_UR1
  return launch_status();
}

}  // namespace dmxq

using namespace dmxq;

// internal entry used by dmxq_bfp_qdq (bfp.hip) for inner == 1 tensors the flat-stream kernel cannot take.
// DMXQ_ERR_UNSUPPORTED = not applicable (caller goes on to the generic one-lane-per-block kernel).
extern "C" int dmxq_internal_bfp_urows(const void* in, void* out, int dtype_in, int dtype_out, int64_t rows, int64_t L,
                                       int64_t B, int wl, int rounding, int symmetric, uint64_t seed, void* stream) {
  const int epl = dtype_in == DMXQ_F32 ? 4 : 8;
  if (B < epl || B > 64 * epl || (B & (B - 1)) != 0 || wl > 22 || L < epl) return DMXQ_ERR_UNSUPPORTED;
  if ((L + epl - 1) / epl + 64 >= ((int64_t)1 << 31)) return DMXQ_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(in) & (dtype_in == DMXQ_F32 ? 3u : 1u)) ||
      (reinterpret_cast<uintptr_t>(out) & (dtype_out == DMXQ_F32 ? 3u : 1u)))
    return DMXQ_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const bool asym = !symmetric;
#define DMXQ_DT(I_, O_)                                                                                          \
  if (dtype_in == I_ && dtype_out == O_) {                                                                       \
    if (rounding == DMXQ_ROUND_NEAREST && wl <= 20)                                                              \
      return asym ? launch_urows<I_, O_, DMXQ_ROUND_NEAREST, true>(in, out, rows, L, B, wl, rounding, seed, s)    \
                  : launch_urows<I_, O_, DMXQ_ROUND_NEAREST, false>(in, out, rows, L, B, wl, rounding, seed, s);  \
    return asym ? launch_urows<I_, O_, kRuntimeRounding, true>(in, out, rows, L, B, wl, rounding, seed, s)        \
                : launch_urows<I_, O_, kRuntimeRounding, false>(in, out, rows, L, B, wl, rounding, seed, s);      \
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
