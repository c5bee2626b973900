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

template <int DTI>
__global__ __launch_bounds__(kThreads) void bfp_pack_rows_kernel(const void* __restrict__ in, int8_t* __restrict__ mant,
                                                                uint8_t* __restrict__ exps, int64_t n_vec, int lpb_arg,
                                                                int wl, int asym) {
  constexpr int EPL = 16 / Elem<DTI>::bytes;
  const int lpb = __builtin_amdgcn_readfirstlane(lpb_arg);
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t v = (int64_t)blockIdx.x * kThreads + threadIdx.x; v < n_vec; v += stride) {
    const u32x4 raw = load_raw16<true>(in, v * 16);
    const uint32_t mb = group_max_u32(absmax_bits<DTI>(raw), lpb);
    const uint32_t Eb = (mb & 0x7F800000u) >> 23;
    float x[EPL];
    widen<DTI, EPL>(raw, x);
    int8_t code[EPL];
    if (Eb == 0u || Eb == 255u) {
#pragma unroll
      for (int k = 0; k < EPL; k++) code[k] = 0;
    } else {
      const float inv_quantum = u2f((uint32_t)(127 - ((int)Eb - 127 - (wl - 2)) ) << 23);  // 2^-(e-(p-2))
#pragma unroll
      for (int k = 0; k < EPL; k++) {
        float q;
        if (asym) { const BfpBlockParams p = bfp_block_params<true, false>(mb, wl); q = bfp_q1<DMXQ_ROUND_NEAREST, true>(x[k], p, wl, DMXQ_ROUND_NEAREST, 0u); }
        else { const BfpBlockParams p = bfp_block_params<false, false>(mb, wl); q = bfp_q1<DMXQ_ROUND_NEAREST, false>(x[k], p, wl, DMXQ_ROUND_NEAREST, 0u); }
        code[k] = (int8_t)(int)(q * inv_quantum);  // exact: q is a multiple of the quantum, |code| <= 2^(p-1)
      }
    }
    if (EPL == 8) {
      uint32_t lo = 0, hi = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) { lo |= (uint32_t)(uint8_t)code[k] << (8 * k); hi |= (uint32_t)(uint8_t)code[4 + k] << (8 * k); }
      *(u32x2*)(mant + v * 8) = u32x2{lo, hi};
    } else {
      uint32_t lo = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) lo |= (uint32_t)(uint8_t)code[k] << (8 * k);
      *(uint32_t*)(mant + v * 4) = lo;
    }
    if ((v % lpb) == 0) exps[v / lpb] = (uint8_t)Eb;
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
    const int grid = grid_for(n_vec);
    if (dtype_in == DMXQ_F32) hipLaunchKernelGGL(bfp_pack_rows_kernel<DMXQ_F32>, dim3(grid), dim3(kThreads), 0, s, in, mant, exps, n_vec, (int)(B / epl), precision, asym);
    else if (dtype_in == DMXQ_F16) hipLaunchKernelGGL(bfp_pack_rows_kernel<DMXQ_F16>, dim3(grid), dim3(kThreads), 0, s, in, mant, exps, n_vec, (int)(B / epl), precision, asym);
    else hipLaunchKernelGGL(bfp_pack_rows_kernel<DMXQ_BF16>, dim3(grid), dim3(kThreads), 0, s, in, mant, exps, n_vec, (int)(B / epl), precision, asym);
  } else {
    const int grid = grid_for(rows * ((L + B - 1) / B));
    if (dtype_in == DMXQ_F32) hipLaunchKernelGGL(bfp_pack_generic_kernel<DMXQ_F32>, dim3(grid), dim3(kThreads), 0, s, in, mant, exps, rows, L, B, precision, asym);
    else if (dtype_in == DMXQ_F16) hipLaunchKernelGGL(bfp_pack_generic_kernel<DMXQ_F16>, dim3(grid), dim3(kThreads), 0, s, in, mant, exps, rows, L, B, precision, asym);
    else hipLaunchKernelGGL(bfp_pack_generic_kernel<DMXQ_BF16>, dim3(grid), dim3(kThreads), 0, s, in, mant, exps, rows, L, B, precision, asym);
  }
  return launch_status();
}

extern "C" int dmxq_bfp_unpack(const int8_t* mant, const uint8_t* exps, void* out, int dtype_out, int64_t rows, int64_t L,
                               int64_t block_size, int precision, void* stream) {
  if (!valid_dtype(dtype_out) || rows < 0 || L < 0 || block_size < 1) return DMXQ_ERR_BAD_ARG;
  if (precision < 2 || precision > 8) return DMXQ_ERR_UNSUPPORTED;
  if (rows * L == 0) return DMXQ_OK;
  if (!mant || !exps || !out) return DMXQ_ERR_BAD_ARG;
  hipLaunchKernelGGL(bfp_unpack_kernel, dim3(grid_for(rows * L)), dim3(kThreads), 0, (hipStream_t)stream, mant, exps, out,
                     dtype_out, rows, L, block_size, precision);
  return launch_status();
}
