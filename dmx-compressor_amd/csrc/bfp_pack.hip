// csrc/bfp_pack.hip — packed on-wire block floating point: the data format on either side of
// the fake-quant path.  The reference only SIMULATES BFP in fp32; its export path names the packed pair
// com.microsoft::QuantizeBFP / DequantizeBFP (numerical/cast.py:34-55) with the frozen type ids of
// numerical/onnx.py (DMX_BFP_16_64 = "8-bit signed mantissa + 8-bit shared exponent, block 64").
//
//   pack   : x[rows, L]  ->  mant int8 [rows, L] (two's-complement codes) + exps uint8 [rows, ceil(L/B)]
//            code = Q(x) / quantum, quantum = 2^(e - (p-2)), exps = biased fp32 exponent of the block max (1..254)
//   unpack : mant, exps  ->  code * 2^(exps - 127 - (p-2))
// Contract: unpack(pack(x)) == dmxq_bfp_qdq(x) bit for bit (same arithmetic: bfp_math.hpp, nearest-even), for
// every block whose maximum is a normal finite number.  Blocks with a denormal or zero maximum pack to all-zero
// codes with exps = 0 (their fp32 simulation keeps p mantissa bits of a denormal, which a p-bit code cannot hold);
// Inf/NaN maxima pack to exps = 255 and unpack to NaN.  precision p <= 8 (codes fit int8).
#include "bfp_math.hpp"

namespace dmxq {

// codes of one 16-byte input vector.  Nearest-even through the magic add of bfp_math.hpp (2): with t = fl(x + base),
// fl(t + M) lies in M's binade, whose ulp is the quantum, and so does K = M + base: the INTEGER difference of the two
// bit patterns is the code.  Clamp and asymmetric rule as bfp_q1_fast (double-rounding form, any input dtype).
// SINGLE (16-bit inputs: bfp_single_rounding_ok, precision <= 8): one rounding, fl(x + K) already lies in K's binade -- the form of the
// hot kernel (bfp_q1_fast<true>), where the asymmetric format is just an asymmetric clamp.  3 operations per element instead of 6.
template <int EPL, bool SINGLE, bool ASYM>
__device__ __forceinline__ void pack_codes_fast(const float (&x)[EPL], uint32_t mb, int wl, int (&code)[EPL]) {
  const BfpBlockParams p = bfp_block_params<ASYM, true>(mb, wl);
  const int cmax = (1 << (wl - 1)) - 1;
  const uint32_t kb = f2u(p.K);
#pragma unroll
  for (int k = 0; k < EPL; k++) {
    if constexpr (SINGLE) {
      const int ci = (int)(f2u(x[k] + p.K) - kb);
      code[k] = ci < (ASYM ? -cmax - 1 : -cmax) ? (ASYM ? -cmax - 1 : -cmax) : (ci > cmax ? cmax : ci);
    } else {
      const int ci = (int)(f2u((x[k] + p.base) + p.M) - kb);
      const int c = ci < -cmax ? -cmax : (ci > cmax ? cmax : ci);
      code[k] = (ASYM && x[k] <= p.thr) ? -cmax - 1 : c;
    }
  }
}
// four codes (each within int8) -> one dword: two byte selections and an OR
__device__ __forceinline__ uint32_t pack4_codes(int c0, int c1, int c2, int c3) {
  return __builtin_amdgcn_perm((uint32_t)c1, (uint32_t)c0, 0x0c0c0400u) | __builtin_amdgcn_perm((uint32_t)c3, (uint32_t)c2, 0x04000c0cu);
}
// the literal form, for the blocks bfp_fast_ok does not cover and the zero / denormal / Inf / NaN maxima (all-zero codes)
template <int EPL>
__device__ __forceinline__ void pack_codes_literal(const float (&x)[EPL], uint32_t mb, int wl, int asym, int (&code)[EPL]) {
  const uint32_t Eb = (mb & 0x7F800000u) >> 23;
  if (Eb == 0u || Eb == 255u) {
#pragma unroll
    for (int k = 0; k < EPL; k++) code[k] = 0;
    return;
  }
  const float inv_quantum = u2f((uint32_t)(127 - ((int)Eb - 127 - (wl - 2))) << 23);  // 2^-(e-(p-2))
  const BfpBlockParams p = bfp_block_params<true, false>(mb, wl);
#pragma unroll
  for (int k = 0; k < EPL; k++) {
    const float q = asym ? bfp_q1<DMXQ_ROUND_NEAREST, true>(x[k], p, wl, DMXQ_ROUND_NEAREST, 0u)
                         : bfp_q1<DMXQ_ROUND_NEAREST, false>(x[k], p, wl, DMXQ_ROUND_NEAREST, 0u);
    code[k] = (int)(q * inv_quantum);  // exact: q is a multiple of the quantum, |code| <= 2^(p-1)
  }
}

// rows of whole power-of-two blocks: a workgroup takes a contiguous tile of kPackThreads x kPackUnroll 16-byte vectors, all of a
// lane's loads in flight before the first block maximum; the magic-add codes for every lane as straight-line code, the blocks
// that form does not cover redone behind one cold branch (the structure of bfp_rows.hpp)
// (idle issue cycles between a lane's loads, common.hpp pace_issue; -DDMXQ_EXP_PACK_PACE=N for A/B builds.  Round 6, same-lease A/B of
//  0 / 2 / 4 on 4096 x 4096 bf16: see profiles/r06_ab_stragglers.txt)
#ifdef DMXQ_EXP_PACK_PACE
constexpr int kPackPace = DMXQ_EXP_PACK_PACE;
#else
constexpr int kPackPace = 0;
#endif
constexpr int kPackThreads = 256, kPackUnroll = 4;  // (round 3, 4096 x 4096 bf16: 256 x 2 and 256 x 4 12.0 us, 256 x 8 12.8, 256 x 16 and 128 x 16 14.5)
// VAR bit 0 (16-bit inputs, n_vec even, mant 16-byte aligned): 16-byte code stores -- neighbouring lanes swap the codes of two unroll
//   slots (one DPP quad_perm each way), so the even lane stores vectors (v, v + 1) of slot 2k and the odd lane those of slot 2k + 1;
// VAR bit 1 (exps 16-byte aligned): the tile's shared exponents (kPackThreads * kPackUnroll / lpb contiguous bytes) go through LDS and
//   leave as 16-byte stores by the first lanes of the workgroup instead of one byte store per block.
template <int DTI, int VAR, int T = kPackThreads, int U = kPackUnroll, bool ASYM = false>
__global__ __launch_bounds__(T) void bfp_pack_rows_kernel(const void* __restrict__ in, int8_t* __restrict__ mant,
                                                                    uint8_t* __restrict__ exps, int64_t n_vec, int lpb_arg,
                                                                    int lpb_log, int wl, int asym) {
  constexpr int EPL = 16 / Elem<DTI>::bytes;
  constexpr bool PAIR = (VAR & 1) != 0 && EPL == 8, QUAD = (VAR & 1) != 0 && EPL == 4, LDSE = (VAR & 2) != 0;
  static_assert(U % 2 == 0 && (!QUAD || U % 4 == 0), "slots are stored in pairs (float32 inputs: in fours)");
  __shared__ __attribute__((aligned(16))) uint8_t se[LDSE ? T * U : 16];
  const int lpb = __builtin_amdgcn_readfirstlane(lpb_arg);
  const int in_blk = threadIdx.x & (lpb - 1);  // v = threadIdx (mod lpb): tile bases and T are multiples of 64
  const int64_t tile0 = (int64_t)blockIdx.x * (T * U);
  const int64_t base = tile0 + threadIdx.x;
  u32x4 raw[U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    const int64_t v = base + u * T;
    // past the end: the same lane position of the LAST block (whole blocks: n_vec % lpb == 0), so that every lane of a
    // wave takes part in the block maxima with defined data; nothing is stored for it
    raw[u] = load_raw16<true>(in, (v < n_vec ? v : n_vec - lpb + in_blk) * 16);
    if (u + 1 < U) pace_issue<kPackPace>();
  }
  __builtin_amdgcn_sched_barrier(0);
  uint32_t w[U][EPL / 4];
#pragma unroll
  for (int u = 0; u < U; u++) {
    const int64_t v = base + u * T;
    const uint32_t mb = group_max_u32(absmax_bits<DTI>(raw[u]), lpb);
    const uint32_t Eb = (mb & 0x7F800000u) >> 23;
    const bool fast = bfp_fast_ok(mb, wl) && Eb != 0u && Eb != 255u;
    float x[EPL];
    widen<DTI, EPL>(raw[u], x);
    int code[EPL];
    pack_codes_fast<EPL, DTI != DMXQ_F32, ASYM>(x, mb, wl, code);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!fast) != 0ull, 0)) {
      if (!fast) pack_codes_literal<EPL>(x, mb, wl, asym, code);
    }
#pragma unroll
    for (int j = 0; j < EPL / 4; j++) w[u][j] = pack4_codes(code[4 * j], code[4 * j + 1], code[4 * j + 2], code[4 * j + 3]);
    if constexpr (!PAIR && !QUAD) {
      if (v < n_vec) {
        if (EPL == 8) __builtin_nontemporal_store(u32x2{w[u][0], w[u][EPL / 4 - 1]}, (u32x2*)(mant + v * 8));
        else __builtin_nontemporal_store(w[u][0], (uint32_t*)(mant + v * 4));
      }
    }
    if constexpr (LDSE) { if (in_blk == 0) se[(u * T + threadIdx.x) >> lpb_log] = (uint8_t)Eb; }
    else { if (v < n_vec && in_blk == 0) exps[v >> lpb_log] = (uint8_t)Eb; }
  }
  if constexpr (PAIR) {
    const bool odd = (threadIdx.x & 1) != 0;
#pragma unroll
    for (int u = 0; u < U; u += 2) {
      // the even lane gives away its slot u + 1 codes, the odd lane its slot u codes
      const uint32_t s0 = odd ? w[u][0] : w[u + 1][0], s1 = odd ? w[u][1] : w[u + 1][1];
      const uint32_t r0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s0, 0xB1, 0xF, 0xF, false);  // quad_perm [1, 0, 3, 2]
      const uint32_t r1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s1, 0xB1, 0xF, 0xF, false);
      const u32x4 o = odd ? u32x4{r0, r1, w[u + 1][0], w[u + 1][1]} : u32x4{w[u][0], w[u][1], r0, r1};
      const int64_t v = base + (odd ? (u + 1) * T - 1 : u * T);  // the even vector of the pair, in slot u (even lane) / u + 1 (odd lane)
      if (v < n_vec) __builtin_nontemporal_store(o, (u32x4*)(mant + v * 8));  // (n_vec even: the pair is in or out as a whole)
    }
  }
  if constexpr (QUAD) {
    // float32 inputs: 4 codes = one dword per lane and slot.  A 4 x 4 transpose inside each quad of lanes (two exchange steps: lanes
    // xor 1, then lanes xor 2) leaves lane j of the quad with the dwords of all four lanes for slot 4k + j: one 16-byte store (n_vec % 4 == 0)
    const bool b1 = (threadIdx.x & 1) != 0, b2 = (threadIdx.x & 2) != 0;
#pragma unroll
    for (int u = 0; u < U; u += 4) {
      uint32_t p0[2], p1[2];  // after step 1: slot u + b1 (p0) and slot u + 2 + b1 (p1), each from lanes (2h, 2h + 1) of this half quad
      {
        const uint32_t sa = b1 ? w[u][0] : w[u + 1][0], sb = b1 ? w[u + 2][0] : w[u + 3][0];
        const uint32_t ra = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sa, 0xB1, 0xF, 0xF, false);  // quad_perm [1, 0, 3, 2]
        const uint32_t rb = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sb, 0xB1, 0xF, 0xF, false);
        p0[0] = b1 ? ra : w[u][0];      p0[1] = b1 ? w[u + 1][0] : ra;
        p1[0] = b1 ? rb : w[u + 2][0];  p1[1] = b1 ? w[u + 3][0] : rb;
      }
      // step 2: the lower half quad gives away its p1, the upper its p0
      const uint32_t s0 = b2 ? p0[0] : p1[0], s1 = b2 ? p0[1] : p1[1];
      const uint32_t r0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s0, 0x4E, 0xF, 0xF, false);  // quad_perm [2, 3, 0, 1]
      const uint32_t r1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s1, 0x4E, 0xF, 0xF, false);
      const u32x4 o = b2 ? u32x4{r0, r1, p1[0], p1[1]} : u32x4{p0[0], p0[1], r0, r1};
      const int j = threadIdx.x & 3;
      const int64_t v = base - j + (int64_t)(u + j) * T;  // the first vector of the quad, in slot u + j
      if (v < n_vec) __builtin_nontemporal_store(o, (u32x4*)(mant + v * 4));
    }
  }
  if constexpr (LDSE) {
    __syncthreads();
    const int64_t blk0 = tile0 >> lpb_log, nblk = n_vec >> lpb_log;
    const int cnt = (T * U) >> lpb_log;  // >= 16 (checked by the dispatcher)
    const int i = threadIdx.x * 16;
    if (i < cnt) {
      if (blk0 + i + 16 <= nblk) {
        __builtin_nontemporal_store(*(const u32x4*)(se + i), (u32x4*)(exps + blk0 + i));
      } else {
        for (int k = 0; k < 16 && blk0 + i + k < nblk; k++) exps[blk0 + i + k] = se[i + k];
      }
    }
  }
}

// generic: one lane per block (any L / B, ragged tails, unaligned)
template <int DTI>
__global__ __launch_bounds__(kThreads) void bfp_pack_generic_kernel(const void* __restrict__ in, int8_t* __restrict__ mant,
                                                                   uint8_t* __restrict__ exps, int64_t rows, int64_t L,
                                                                   int64_t B, int wl, int asym) {
  const int64_t nblk = (L + B - 1) / B;
  const int64_t total = rows * nblk;
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x; t < total; t += stride) {
    const int64_t r = t / nblk, k = t % nblk;
    const int64_t e0 = r * L + k * B, len = (L - k * B < B) ? (L - k * B) : B;
    uint32_t mb = 0u;
    for (int64_t i = 0; i < len; i++) mb = max(mb, f2u(load1<DTI>(in, e0 + i)) & 0x7FFFFFFFu);
    const uint32_t Eb = (mb & 0x7F800000u) >> 23;
    exps[t] = (uint8_t)Eb;
    if (Eb == 0u || Eb == 255u) {
      for (int64_t i = 0; i < len; i++) mant[e0 + i] = 0;
      continue;
    }
    const float inv_quantum = u2f((uint32_t)(127 - ((int)Eb - 127 - (wl - 2))) << 23);
    const BfpBlockParams ps = bfp_block_params<false, false>(mb, wl), pa = bfp_block_params<true, false>(mb, wl);
    for (int64_t i = 0; i < len; i++) {
      const float x = load1<DTI>(in, e0 + i);
      const float q = asym ? bfp_q1<DMXQ_ROUND_NEAREST, true>(x, pa, wl, DMXQ_ROUND_NEAREST, 0u)
                           : bfp_q1<DMXQ_ROUND_NEAREST, false>(x, ps, wl, DMXQ_ROUND_NEAREST, 0u);
      mant[e0 + i] = (int8_t)(int)(q * inv_quantum);
    }
  }
}

__global__ __launch_bounds__(kThreads) void bfp_unpack_kernel(const int8_t* __restrict__ mant, const uint8_t* __restrict__ exps,
                                                             void* __restrict__ out, int dto, int64_t rows, int64_t L,
                                                             int64_t B, int wl) {
  const int64_t nblk = (L + B - 1) / B;
  const int64_t n = rows * L;
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < n; e += stride) {
    const int64_t r = e / L, c = e % L;
    const int Eb = exps[r * nblk + c / B];
    float v;
    if (Eb == 255) v = u2f(0x7FC00000u);
    else if (Eb == 0) v = 0.0f;
    else v = ldexpf((float)mant[e], Eb - 127 - (wl - 2));
    store_rt(out, dto, e, v);
  }
}

// vector twin: rows of whole power-of-two blocks of >= 8 codes.  A lane turns 8 codes (one 8-byte load) of ONE block
// into 8 outputs (one or two 16-byte stores); the block index is a shift of the flat element index.
// value = (code * 2^-(p-2)) * 2^(E-127): the first product is exact, the second rounds once (only a denormal result
// rounds at all) -- the same value as the scalar kernel's ldexpf -- and cannot overflow.
constexpr int kUnpackUnroll = 8;
// PAIR (n_vec even, mant 16-byte aligned): 16-byte code loads -- the even lane loads the codes of vectors (v, v + 1) of slot 2k, the odd
// lane those of slot 2k + 1, and the two swap halves (one DPP quad_perm each way): the mirror image of bfp_pack_rows_kernel's stores
template <int DTO, bool PAIR, int U = kUnpackUnroll>
__global__ __launch_bounds__(kThreads) void bfp_unpack_vec_kernel(const int8_t* __restrict__ mant,
                                                                 const uint8_t* __restrict__ exps, void* __restrict__ out,
                                                                 int64_t n_vec, int b_shift /*log2(B / 8)*/, int wl) {
  const float down = u2f((uint32_t)(127 - (wl - 2)) << 23);
  // workgroup-contiguous tiles of kThreads x kUnpackUnroll code vectors, one pass per workgroup, all loads first (round 3: the
  // grid-strided 4-in-flight form measured 11.0 us on 4096 x 4096, 58 %)
  static_assert(U % 2 == 0, "slots are loaded in pairs");
  const int64_t stride = kThreads;
  {
    const int64_t v0 = (int64_t)blockIdx.x * (kThreads * U) + threadIdx.x;
    u32x2 m[U];
    uint32_t eb[U];
    if constexpr (PAIR) {
      const bool odd = (threadIdx.x & 1) != 0;
      u32x4 q[U / 2];
#pragma unroll
      for (int u = 0; u < U; u += 2) {
        const int64_t p = v0 + (odd ? (u + 1) * stride - 1 : u * stride);  // the even vector of this lane's pair
        q[u / 2] = __builtin_nontemporal_load((const u32x4*)(mant + (p < n_vec ? p : n_vec - 2) * 8));  // clamped: unconditional loads
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int64_t v = v0 + u * stride < n_vec ? v0 + u * stride : n_vec - 1;
        eb[u] = exps[v >> b_shift];
      }
#pragma unroll
      for (int u = 0; u < U; u += 2) {
        const u32x4 o = q[u / 2];
        // the even lane gives away its second half (vector v + 1 of slot u), the odd lane its first (vector v - 1 of slot u + 1)
        const uint32_t s0 = odd ? o[0] : o[2], s1 = odd ? o[1] : o[3];
        const uint32_t r0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s0, 0xB1, 0xF, 0xF, false);  // quad_perm [1, 0, 3, 2]
        const uint32_t r1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s1, 0xB1, 0xF, 0xF, false);
        m[u] = odd ? u32x2{r0, r1} : u32x2{o[0], o[1]};
        m[u + 1] = odd ? u32x2{o[2], o[3]} : u32x2{r0, r1};
      }
    } else {
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t v = v0 + u * stride < n_vec ? v0 + u * stride : n_vec - 1;  // clamped: unconditional loads
      m[u] = __builtin_nontemporal_load((const u32x2*)(mant + v * 8));
      eb[u] = exps[v >> b_shift];
    }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t v = v0 + u * stride;
      if (v < n_vec) {
        const float up = eb[u] == 255u ? u2f(0x7FC00000u) : u2f(eb[u] << 23);
        if (eb[u] == 0u) m[u] = u32x2{0u, 0u};  // zero / denormal block: +0 whatever the codes (as the scalar kernel)
        float y[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const uint32_t w = k < 4 ? m[u].x : m[u].y;
          const int c = (int)(w << (24 - 8 * (k & 3))) >> 24;  // sign-extended byte k
          y[k] = ((float)c * down) * up;                       // NaN block: c * NaN = NaN (also for c = 0)
        }
        store_vec<DTO, 8, true>(out, v * 8, y);
      }
    }
  }
}

// float32 outputs: 8 codes become 32 bytes, and two 16-byte stores per lane at a 32-byte lane stride leave every store instruction with
// half-used lines (measured 44 % of roofline whatever the unroll, tools/tune_pack.hip).  Here a wave takes 512 codes per slot as two
// regions of 256: the even lane of a pair loads 8 bytes of region 0, the odd lane 8 bytes of region 1, they swap one dword (DPP
// quad_perm), and lane l then holds dword l of BOTH regions -- 4 codes each -> two 16-byte stores, each instruction one contiguous KiB.
// Same values as bfp_unpack_vec_kernel.  n_codes: a multiple of 8.
template <int U>
__global__ __launch_bounds__(kThreads) void bfp_unpack_f32_kernel(const int8_t* __restrict__ mant, const uint8_t* __restrict__ exps,
                                                                 float* __restrict__ out, int64_t n_codes, int b_log /*log2 B*/, int wl) {
  const float down = u2f((uint32_t)(127 - (wl - 2)) << 23);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool odd = (lane & 1) != 0;
  const int64_t tile0 = (int64_t)blockIdx.x * (kThreads * U * 8);
  u32x2 m[U];
  uint32_t ea[U], eb[U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    const int64_t chunk = tile0 + (int64_t)(u * kThreads + wave * 64) * 8;
    const int64_t ld = chunk + (odd ? 256 : 0) + 8 * (lane >> 1);
    m[u] = __builtin_nontemporal_load((const u32x2*)(mant + (ld + 8 <= n_codes ? ld : n_codes - 8)));  // clamped: unconditional loads
    const int64_t ca = chunk + 4 * lane, cb = ca + 256;
    ea[u] = exps[(ca < n_codes ? ca : n_codes - 1) >> b_log];
    eb[u] = exps[(cb < n_codes ? cb : n_codes - 1) >> b_log];
  }
#pragma unroll
  for (int u = 0; u < U; u++) {
    const int64_t chunk = tile0 + (int64_t)(u * kThreads + wave * 64) * 8;
    // even lane: has region-0 dwords (2i, 2i + 1), gives away the second; odd lane: has region-1 dwords (2i, 2i + 1), gives away the first
    const uint32_t r = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(odd ? m[u].x : m[u].y), 0xB1, 0xF, 0xF, false);  // quad_perm [1, 0, 3, 2]
    const uint32_t wa = odd ? r : m[u].x, wb = odd ? m[u].y : r;
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int64_t c0 = chunk + 4 * lane + 256 * h;
      if (c0 < n_codes) {
        const uint32_t e = h ? eb[u] : ea[u];
        const uint32_t w = e == 0u ? 0u : (h ? wb : wa);  // zero / denormal block: +0 whatever the codes
        const float up = e == 255u ? u2f(0x7FC00000u) : u2f(e << 23);
        float y[4];
#pragma unroll
        for (int k = 0; k < 4; k++) y[k] = ((float)((int)(w << (24 - 8 * k)) >> 24) * down) * up;
        __builtin_nontemporal_store(u32x4{f2u(y[0]), f2u(y[1]), f2u(y[2]), f2u(y[3])}, (u32x4*)(out + c0));
      }
    }
  }
}

}  // namespace dmxq

using namespace dmxq;

extern "C" int dmxq_bfp_pack(const void* in, int dtype_in, int8_t* mant, uint8_t* exps, int64_t rows, int64_t L,
                             int64_t block_size, int precision, int symmetric, void* stream) {
  if (!valid_dtype(dtype_in) || rows < 0 || L < 0 || block_size < 1) return DMXQ_ERR_BAD_ARG;
  if (precision < 2 || precision > 8) return DMXQ_ERR_UNSUPPORTED;
  if (rows * L == 0) return DMXQ_OK;
  if (!in || !mant || !exps) return DMXQ_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int epl = dtype_in == DMXQ_F32 ? 4 : 8;
  const int64_t B = block_size, n = rows * L;
  const bool pow2 = (B & (B - 1)) == 0;
  const int asym = symmetric ? 0 : 1;
  if (L % B == 0 && pow2 && B >= epl && B <= 64 * epl && aligned16(in) && (reinterpret_cast<uintptr_t>(mant) & 7u) == 0) {
    const int64_t n_vec = n / epl;
    const bool small = n_vec <= (3 << 18);  // (tools/tune_pack.hip: up to 12 MiB of 16-bit input two vectors per lane, four beyond)
    const int tile = kPackThreads * (small ? 2 : kPackUnroll);
    const int64_t tiles = (n_vec + tile - 1) / tile;
    if (tiles > 0x7FFFFFFF) return DMXQ_ERR_UNSUPPORTED;
    const unsigned grid = (unsigned)tiles;
    int lpb_log = 0;
    while (((int64_t)epl << lpb_log) < B) lpb_log++;
    // (the LDS form writes a tile's exponents as whole 16-byte vectors: at least 16 blocks per tile -- not B = 64 lane-vectors on the two-slot tiles)
    const bool pair = epl == 8 && n_vec % 2 == 0 && aligned16(mant), ldse = aligned16(exps) && (tile >> lpb_log) >= 16;
#define DMXQ_PKL(DT_, V_, U_, A_) DMXQ_LAUNCH((bfp_pack_rows_kernel<DT_, V_, kPackThreads, U_, A_>), dim3(grid), dim3(kPackThreads), 0, s, in, mant, exps, n_vec, (int)(B / epl), lpb_log, precision, asym)
#define DMXQ_PKU(DT_, V_, U_) do { if (asym) DMXQ_PKL(DT_, V_, U_, true); else DMXQ_PKL(DT_, V_, U_, false); } while (0)
#define DMXQ_PK(DT_, V_) do { if (small) DMXQ_PKU(DT_, V_, 2); else DMXQ_PKU(DT_, V_, kPackUnroll); } while (0)
#define DMXQ_PKV(DT_) do { if (pair && ldse) DMXQ_PK(DT_, 3); else if (pair) DMXQ_PK(DT_, 1); else if (ldse) DMXQ_PK(DT_, 2); else DMXQ_PK(DT_, 0); } while (0)
    if (dtype_in == DMXQ_F32) {
      // (16-byte code stores need four slots: the two-slot geometry of small tensors keeps 4-byte stores)
      const bool quad = !small && n_vec % 4 == 0 && aligned16(mant);
      if (quad && ldse) DMXQ_PKU(DMXQ_F32, 3, kPackUnroll);
      else if (quad) DMXQ_PKU(DMXQ_F32, 1, kPackUnroll);
      else if (ldse) DMXQ_PK(DMXQ_F32, 2);
      else DMXQ_PK(DMXQ_F32, 0);
    }
    else if (dtype_in == DMXQ_F16) DMXQ_PKV(DMXQ_F16);
    else DMXQ_PKV(DMXQ_BF16);
#undef DMXQ_PKL
#undef DMXQ_PKU
#undef DMXQ_PKV
#undef DMXQ_PK
  } else {
    const int grid = grid_for(rows * ((L + B - 1) / B));
    if (dtype_in == DMXQ_F32) DMXQ_LAUNCH(bfp_pack_generic_kernel<DMXQ_F32>, dim3(grid), dim3(kThreads), 0, s, in, mant, exps, rows, L, B, precision, asym);
    else if (dtype_in == DMXQ_F16) DMXQ_LAUNCH(bfp_pack_generic_kernel<DMXQ_F16>, dim3(grid), dim3(kThreads), 0, s, in, mant, exps, rows, L, B, precision, asym);
    else DMXQ_LAUNCH(bfp_pack_generic_kernel<DMXQ_BF16>, dim3(grid), dim3(kThreads), 0, s, in, mant, exps, rows, L, B, precision, asym);
  }
  return launch_status();
}

extern "C" int dmxq_bfp_unpack(const int8_t* mant, const uint8_t* exps, void* out, int dtype_out, int64_t rows, int64_t L,
                               int64_t block_size, int precision, void* stream) {
  if (!valid_dtype(dtype_out) || rows < 0 || L < 0 || block_size < 1) return DMXQ_ERR_BAD_ARG;
  if (precision < 2 || precision > 8) return DMXQ_ERR_UNSUPPORTED;
  if (rows * L == 0) return DMXQ_OK;
  if (!mant || !exps || !out) return DMXQ_ERR_BAD_ARG;
  const int64_t B = block_size;
  if (L % B == 0 && (B & (B - 1)) == 0 && B >= 8 && aligned16(out) && (reinterpret_cast<uintptr_t>(mant) & 7u) == 0) {
    const int64_t n_vec = rows * L / 8;
    int b_shift = 0;
    while (((int64_t)8 << b_shift) < B) b_shift++;
    // (tools/tune_pack.hip, bf16 out: up to 12 MiB of output 8-byte loads x 4 -- 3.5 us vs 3.8 on 1024 x 4096 --, up to 32 MiB 16-byte loads
    // x 8 -- 8.9 us vs 9.7 on 4096 x 4096 --, beyond that 16-byte loads x 2: 33.2 us vs 34.1 on 8192 x 8192)
    const bool pair = n_vec % 2 == 0 && aligned16(mant) && n_vec > (3 << 18);
    const int unroll = n_vec <= (3 << 18) ? 4 : (n_vec <= (1 << 21) || !pair ? 8 : 2);
    const int64_t tiles = (n_vec + (int64_t)kThreads * unroll - 1) / ((int64_t)kThreads * unroll);
    if (tiles > 0x7FFFFFFF) return DMXQ_ERR_UNSUPPORTED;
    const unsigned grid = (unsigned)tiles;
    hipStream_t s = (hipStream_t)stream;
#define DMXQ_UPK(DT_, P_, U_) DMXQ_LAUNCH((bfp_unpack_vec_kernel<DT_, P_, U_>), dim3(grid), dim3(kThreads), 0, s, mant, exps, out, n_vec, b_shift, precision)
#define DMXQ_UP(DT_) do { if (unroll == 4) DMXQ_UPK(DT_, false, 4); else if (!pair) DMXQ_UPK(DT_, false, 8); else if (unroll == 8) DMXQ_UPK(DT_, true, 8); \
                          else DMXQ_UPK(DT_, true, 2); } while (0)
    if (dtype_out == DMXQ_F32) {
      constexpr int UF = 2;
      const int64_t ft = (n_vec + (int64_t)kThreads * UF - 1) / ((int64_t)kThreads * UF);
      if (ft > 0x7FFFFFFF) return DMXQ_ERR_UNSUPPORTED;
      DMXQ_LAUNCH((bfp_unpack_f32_kernel<UF>), dim3((unsigned)ft), dim3(kThreads), 0, s, mant, exps, (float*)out, n_vec * 8, b_shift + 3, precision);
    }
    else if (dtype_out == DMXQ_F16) DMXQ_UP(DMXQ_F16);
    else DMXQ_UP(DMXQ_BF16);
#undef DMXQ_UPK
#undef DMXQ_UP
    return launch_status();
  }
  DMXQ_LAUNCH(bfp_unpack_kernel, dim3(grid_for(rows * L)), dim3(kThreads), 0, (hipStream_t)stream, mant, exps, out,
                     dtype_out, rows, L, block_size, precision);
  return launch_status();
}
