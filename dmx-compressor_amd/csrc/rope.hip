// csrc/rope.hip — APPLY_LLAMA_ROPE, the last function id of the approximator slot (src/dmx/compressor/__init__.py:136-138).
//
// The reference's exact function is transformers' apply_rotary_pos_emb as restated in modeling/nn/custom_modules.py:142-172:
//     x_embed = (x * cos) + (rotate_half(x) * sin),   rotate_half(x) = cat(-x[..., D/2:], x[..., :D/2])
// with cos / sin of shape [B, S, D] unsqueezed so that they broadcast over the heads.  torch evaluates it IN THE TENSOR
// DTYPE: two products and a sum, each rounded to the dtype -- reproduced here (fp32 arithmetic, round to the dtype after
// every operation, -ffp-contract=off), so the result equals torch's CPU result BIT FOR BIT, not within a tolerance.
// x is viewed as [B, n1, n2, D]; cos / sin rows are indexed by (b, n2) when the embedding broadcasts over dim 1
// (unsqueeze_dim = 1: x = [B, heads, S, D]) or by (b, n1) (unsqueeze_dim = 2: x = [B, S, heads, D]).
// One lane owns 8 (16-bit) or 4 (fp32) consecutive d of one row and also loads the partner vector at d +- D/2 (a second
// read of x, served by L2: the two halves of a row are 128-256 bytes apart): algorithmic traffic = x in + x out (+ the
// cos / sin tables once per head).
#include "floatq.hpp"

namespace dmxq {

template <int DT>
__device__ __forceinline__ float rnd_dt(float v) {
  if (DT == DMXQ_BF16) return (float)(__bf16)v;
  if (DT == DMXQ_F16) return (float)(_Float16)opaque(v);
  return v;
}

// GEN: the module's casts in their general (rounding) form, per element in fp32 (floatq.hpp castg_vec), for float32 tensors and formats
// that are not range-only; the Range16 arguments are then identities
struct RopeCasts { CastG x, c, s, o; };
template <int DT, bool GEN = false>
__global__ __launch_bounds__(kThreads) void rope_kernel(const void* __restrict__ x, const void* __restrict__ cs,
                                                       const void* __restrict__ sn, void* __restrict__ out, int64_t n_vec,
                                                       int vpr /*vectors per row*/, FastDiv31 f_vpr, FastDiv31 f_n2, FastDiv31 f_n1,
                                                       int n1, int n2, int over_dim1, Range16 rgx, Range16 rgc, Range16 rgs,
                                                       Range16 rgo, RopeCasts gc) {
  constexpr int EPL = 16 / Elem<DT>::bytes;
  const int half = vpr / 2;
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t v = (int64_t)blockIdx.x * kThreads + threadIdx.x; v < n_vec; v += stride) {
    const uint32_t row = f_vpr.div((uint32_t)v), vc = (uint32_t)v - row * (uint32_t)vpr;       // (row, vector in row)
    const uint32_t r12 = f_n2.div(row), i2 = row - r12 * (uint32_t)n2;                          // row = (b * n1 + i1) * n2 + i2
    const uint32_t b = f_n1.div(r12), i1 = r12 - b * (uint32_t)n1;
    const uint32_t crow = over_dim1 ? b * (uint32_t)n2 + i2 : b * (uint32_t)n1 + i1;
    const bool lo = (int)vc < half;
    const int64_t pv = lo ? v + half : v - half;                                                // partner vector
    const u32x4 rx = load_raw16<true>(x, v * 16), rp = *(const u32x4*)((const char*)x + pv * 16);
    const u32x4 rc = *(const u32x4*)((const char*)cs + ((int64_t)crow * vpr + vc) * 16);
    const u32x4 rs = *(const u32x4*)((const char*)sn + ((int64_t)crow * vpr + vc) * 16);
    u32x4 cx = rx, cp = rp, cc = rc, cs2 = rs;
    if constexpr (DT != DMXQ_F32) {  // the module's input casts (dmxq_rope_cast: range-only formats; identity ranges for dmxq_rope)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        cx[j] = range16_word(rx[j], rgx); cp[j] = range16_word(rp[j], rgx);
        cc[j] = range16_word(rc[j], rgc); cs2[j] = range16_word(rs[j], rgs);
      }
    }
    float xv[EPL], pvv[EPL], c[EPL], s[EPL], y[EPL];
    widen<DT, EPL>(cx, xv); widen<DT, EPL>(cp, pvv); widen<DT, EPL>(cc, c); widen<DT, EPL>(cs2, s);
    if constexpr (GEN) { castg_vec<DT, EPL>(xv, gc.x); castg_vec<DT, EPL>(pvv, gc.x); castg_vec<DT, EPL>(c, gc.c); castg_vec<DT, EPL>(s, gc.s); }
#pragma unroll
    for (int k = 0; k < EPL; k++) {
      const float rot = lo ? -pvv[k] : pvv[k];
      const float t1 = rnd_dt<DT>(xv[k] * c[k]);
      const float t2 = rnd_dt<DT>(rot * s[k]);
      y[k] = GEN ? castg_dt<DT>(t1 + t2) : t1 + t2;   // rounded to DT by the store (GEN: before the output cast)
    }
    if constexpr (GEN) castg_vec<DT, EPL>(y, gc.o);
    OutVec<DT, EPL> o = pack_vec<DT, EPL>(y);
    if constexpr (DT != DMXQ_F32) {  // ... and its output cast
#pragma unroll
      for (int j = 0; j < 4; j++) o.w[j] = range16_word(o.w[j], rgo);
    }
    store_out<DT, EPL, true>((char*)out + v * 16, o);
  }
}

}  // namespace dmxq

using namespace dmxq;

static int rope_launch(const void* x, const void* cos_tab, const void* sin_tab, void* out, int dtype, int64_t B, int64_t n1, int64_t n2,
                       int64_t D, int broadcast_over_dim1, const Range16& rgx, const Range16& rgc, const Range16& rgs, const Range16& rgo,
                       void* stream, const RopeCasts* gen = nullptr) {
  if (!valid_dtype(dtype) || B < 0 || n1 < 0 || n2 < 0 || D < 0) return DMXQ_ERR_BAD_ARG;
  const int64_t n = B * n1 * n2 * D;
  if (n == 0) return DMXQ_OK;
  if (!x || !cos_tab || !sin_tab || !out) return DMXQ_ERR_BAD_ARG;
  const int epl = dtype == DMXQ_F32 ? 4 : 8;
  // whole aligned vectors in each half of a row, 31-bit indices; anything else is left to the caller (torch's own ops)
  if (D % (2 * epl) != 0 || n >= ((int64_t)1 << 31) || !aligned16(x) || !aligned16(out) || !aligned16(cos_tab) || !aligned16(sin_tab) ||
      x == out)
    return DMXQ_ERR_UNSUPPORTED;
  const int vpr = (int)(D / epl);
  const int64_t n_vec = n / epl;
  hipStream_t s = (hipStream_t)stream;
  const int grid = grid_for((n_vec + 1) / 2);
  const RopeCasts none{};
#define DMXQ_ROPE(D_, G_) DMXQ_LAUNCH((rope_kernel<D_, G_>), dim3(grid), dim3(kThreads), 0, s, x, cos_tab, sin_tab, out, n_vec, vpr, make_fastdiv31(vpr), \
                                  make_fastdiv31(n2), make_fastdiv31(n1), (int)n1, (int)n2, broadcast_over_dim1 ? 1 : 0, rgx, rgc, rgs, rgo, gen ? *gen : none)
  if (gen) { if (dtype == DMXQ_F32) DMXQ_ROPE(DMXQ_F32, true); else if (dtype == DMXQ_F16) DMXQ_ROPE(DMXQ_F16, true); else DMXQ_ROPE(DMXQ_BF16, true); }
  else { if (dtype == DMXQ_F32) DMXQ_ROPE(DMXQ_F32, false); else if (dtype == DMXQ_F16) DMXQ_ROPE(DMXQ_F16, false); else DMXQ_ROPE(DMXQ_BF16, false); }
#undef DMXQ_ROPE
  return launch_status();
}

extern "C" int dmxq_rope(const void* x, const void* cos_tab, const void* sin_tab, void* out, int dtype, int64_t B, int64_t n1,
                         int64_t n2, int64_t D, int broadcast_over_dim1, void* stream) {
  const Range16 id{0xFFFFFFFFu, 0u};
  return rope_launch(x, cos_tab, sin_tab, out, dtype, B, n1, n2, D, broadcast_over_dim1, id, id, id, id, stream);
}

// One operand (q or k) of an ApplyRotaryPosEmb DmxModule with its casts: out = cast_out(rope(cast_x(x), cast_cos(cos), cast_sin(sin))).
// bf16 tensors and range-only formats only (DMXQ_ERR_UNSUPPORTED otherwise), see dmxq_binary_cast.
extern "C" int dmxq_rope_cast(const void* x, const void* cos_tab, const void* sin_tab, void* out, int dtype, int64_t B, int64_t n1,
                              int64_t n2, int64_t D, int broadcast_over_dim1, const dmxq_float_fmt* cast_x, const dmxq_float_fmt* cast_cos,
                              const dmxq_float_fmt* cast_sin, const dmxq_float_fmt* cast_out, void* stream) {
  Range16 rx, rc, rs, ro;
  if (!valid_dtype(dtype)) return DMXQ_ERR_BAD_ARG;
  if (!range16_of(cast_x, dtype, &rx) || !range16_of(cast_cos, dtype, &rc) || !range16_of(cast_sin, dtype, &rs) || !range16_of(cast_out, dtype, &ro)) {
    RopeCasts g;  // not range-only (casts that round, float32 tensors): the general form
    if (!castg_of(cast_x, &g.x) || !castg_of(cast_cos, &g.c) || !castg_of(cast_sin, &g.s) || !castg_of(cast_out, &g.o)) return DMXQ_ERR_UNSUPPORTED;
    const Range16 id{0xFFFFFFFFu, 0u};
    return rope_launch(x, cos_tab, sin_tab, out, dtype, B, n1, n2, D, broadcast_over_dim1, id, id, id, id, stream, &g);
  }
  return rope_launch(x, cos_tab, sin_tab, out, dtype, B, n1, n2, D, broadcast_over_dim1, rx, rc, rs, ro, stream);
}
