// csrc/fixed_multi.hip — affine integer Q->DQ (INT8 / INT4 group quantisation) of MANY weights in one launch.
//
// Replaces, for a whole model at once, the per-module  x / scale + zp -> FixedPoint cast -> (x - zp) * scale  of
// numerical/cast.py:278-296 that `hf.pipeline(..., dmx_config="BASIC")` + INT8 group-quantised Linear weights runs for each of
// opt-125m's 73 Linear layers (BASELINE.json configs[2]): each of those tensors is launch-bound on its own (768 x 768 bf16:
// 0.4 us of streaming behind ~4 us of launch), exactly like the BFP case (bfp.hip dmxq_bfp_qdq_multi).
// Scope of the batched path: the integer formats of the alias tables (fraction 0, clamped, nearest: the SIMPLE case of
// elementwise.hip), a [C, inner] weight whose scale / zero point belong to slabs of `group_size` rows (ch_axis = 0) or to
// the whole tensor, inner a multiple of the lane-vector, fewer than 2^31 elements.  Every other tensor of the call gets its
// own dmxq_fixed_qdq launch, so the result is ALWAYS what one call per tensor would give.
// Arithmetic: identical to FixedOp<kUniform, SIMPLE> (same rounding form, same division: common.hpp div_for_clamped_int
// when the group's scale is inside its proven range, the IEEE division otherwise).
#include "common.hpp"

namespace dmxq {

constexpr int kFixedMultiMax = 40, kFmThreads = 256, kFmUnroll = 4;
struct FixedMultiDesc {
  const void* in; void* out; const float* scale; const int64_t* zp;
  int64_t n_vec, tile0;
  FastDiv31 f_grp;  // elements per (scale, zero point) group
  uint32_t pad;
};
struct FixedMultiArgs { FixedMultiDesc d[kFixedMultiMax]; int n; float t_min, t_max; };

__device__ __forceinline__ float fixed_simple_q(float x, float sc, float z, float rs, bool fast, float t_min, float t_max) {
  x = (fast ? div_for_clamped_int(x, Recip{sc, rs}) : x / sc) + z;
  float v = rintf((x + 0.5f) - 0.5f);  // sim_helper.cpp:14-21 in its fp32-only form (elementwise.hip rne_minus_half, SIMPLE case)
  v = v > t_max ? t_max : (v < t_min ? t_min : v);
  return (v - z) * sc;
}

template <int DTI, int DTO>
__global__ __launch_bounds__(kFmThreads) void fixed_multi_kernel(const FixedMultiArgs a) {
  constexpr int EPL = 16 / Elem<DTI>::bytes, OVB = EPL * Elem<DTO>::bytes;
  constexpr int64_t TILE = (int64_t)kFmThreads * kFmUnroll;
  const int64_t gt = blockIdx.x;
  int k = 0;
  for (int i = 1; i < a.n; i++) k = (a.d[i].tile0 <= gt) ? i : k;
  const FixedMultiDesc& d = a.d[k];
  const int64_t tile = gt - d.tile0;
  const char* src = (const char*)d.in + tile * (TILE * 16);
  char* dst = (char*)d.out + tile * (TILE * OVB);
  const int64_t v0 = tile * TILE + threadIdx.x;
  u32x4 raw[kFmUnroll];
  float sc[kFmUnroll], z[kFmUnroll];
  const int64_t lastv = d.n_vec - 1;
#pragma unroll
  for (int u = 0; u < kFmUnroll; u++) {
    const int64_t v = v0 + (int64_t)u * kFmThreads;
    const int64_t vc = v < d.n_vec ? v : lastv;  // clamped: unconditional loads
    raw[u] = load_raw16<true>(d.in, vc * 16);
    const uint32_t g = d.f_grp.div((uint32_t)(vc * EPL));
    sc[u] = d.scale[g];
    z[u] = (float)d.zp[g];
  }
  (void)src;
  __builtin_amdgcn_sched_barrier(0);
  OutVec<DTO, EPL> o[kFmUnroll];
#pragma unroll
  for (int u = 0; u < kFmUnroll; u++) {
    float x[EPL], y[EPL];
    widen<DTI, EPL>(raw[u], x);
    const float rs = 1.0f / sc[u];
    if (__builtin_amdgcn_ballot_w64(!recip_ok(sc[u])) == 0ull) {
      // pairs through the packed fp32 pipe (common.hpp affine_int_pairs); lanes holding an Inf / NaN quotient redo theirs
      const bool special = affine_int_pairs<EPL>(x, y, sc[u], rs, z[u], a.t_min, a.t_max);
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(special) != 0ull, 0)) {
        if (special) {
#pragma unroll
          for (int j = 0; j < EPL; j++) y[j] = fixed_simple_q(x[j], sc[u], z[u], rs, true, a.t_min, a.t_max);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < EPL; j++) y[j] = fixed_simple_q(x[j], sc[u], z[u], rs, false, a.t_min, a.t_max);
    }
    o[u] = pack_vec<DTO, EPL>(y);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int u = 0; u < kFmUnroll; u++) {
    const int64_t v = v0 + (int64_t)u * kFmThreads;
    if (v < d.n_vec) store_out<DTO, EPL, true>(dst + (int64_t)u * (kFmThreads * OVB) + threadIdx.x * (int64_t)OVB, o[u]);
  }
}

template <int DTI, int DTO>
static int launch_fixed_multi(const FixedMultiArgs& a, int64_t tiles, hipStream_t s) {
  DMXQ_LAUNCH((fixed_multi_kernel<DTI, DTO>), dim3((unsigned)tiles), dim3(kFmThreads), 0, s, a);
  return launch_status();
}

}  // namespace dmxq

using namespace dmxq;

extern "C" int dmxq_fixed_qdq(const void* in, void* out, int dtype_in, int dtype_out, int64_t outer, int64_t C, int64_t inner,
                              int precision, int fraction, int clamp, int symmetric, int rounding, const float* scale,
                              const int64_t* zero_point, int64_t group_size, uint64_t seed, void* stream);

extern "C" int dmxq_fixed_qdq_multi(const dmxq_affine_desc* tensors, int64_t n_tensors, int dtype_in, int dtype_out, int precision,
                                    int fraction, int clamp, int symmetric, int rounding, int64_t group_size, uint64_t seed,
                                    void* stream) {
  if (n_tensors < 0 || (n_tensors > 0 && !tensors)) return DMXQ_ERR_BAD_ARG;
  if (!valid_dtype(dtype_in) || !valid_dtype(dtype_out) || !valid_rounding(rounding) || precision < 1 || group_size < 1) return DMXQ_ERR_BAD_ARG;
  for (int64_t i = 0; i < n_tensors; i++) {
    const dmxq_affine_desc& t = tensors[i];
    if (t.outer < 0 || t.C < 0 || t.inner < 0 || (t.scale == nullptr) != (t.zero_point == nullptr)) return DMXQ_ERR_BAD_ARG;
    if (t.outer * t.C * t.inner > 0 && (!t.in || !t.out)) return DMXQ_ERR_BAD_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  const int epl = dtype_in == DMXQ_F32 ? 4 : 8;
  const bool simple = fraction == 0 && clamp && rounding == DMXQ_ROUND_NEAREST && precision <= 22;
  constexpr int64_t TILE = (int64_t)kFmThreads * kFmUnroll;
  FixedMultiArgs a;
  a.n = 0;
  a.t_min = (float)(-ldexp(1.0, precision - 1));   // sim_helper.cpp:5-12 fixed_min_max at fraction 0
  a.t_max = (float)(-(double)a.t_min - 1.0);
  if (symmetric) a.t_min = (float)((double)a.t_min + 1.0);
  int64_t tiles = 0;
  int rc = DMXQ_OK;
  auto flush = [&]() {
    if (a.n == 0) return;
    int r = DMXQ_ERR_BAD_ARG;
#define DMXQ_DT(I_, O_) if (dtype_in == I_ && dtype_out == O_) r = launch_fixed_multi<I_, O_>(a, tiles, s);
    DMXQ_DT(DMXQ_BF16, DMXQ_BF16) DMXQ_DT(DMXQ_F16, DMXQ_F16) DMXQ_DT(DMXQ_F32, DMXQ_F32) DMXQ_DT(DMXQ_BF16, DMXQ_F32)
    DMXQ_DT(DMXQ_F16, DMXQ_F32) DMXQ_DT(DMXQ_F32, DMXQ_BF16) DMXQ_DT(DMXQ_F32, DMXQ_F16)
#undef DMXQ_DT
    if (r != DMXQ_OK) rc = r;
    a.n = 0; tiles = 0;
  };
  for (int64_t i = 0; i < n_tensors && rc == DMXQ_OK; i++) {
    const dmxq_affine_desc& t = tensors[i];
    const int64_t n = t.outer * t.C * t.inner;
    if (n == 0) continue;
    // batchable: SIMPLE format, affine, one contiguous run of elements per group (outer == 1 with row slabs, or one group),
    // whole vectors, 16-byte aligned, 31-bit indices
    const bool one_group = t.C <= 1 || group_size >= t.C;
    const int64_t grp_elems = one_group ? n : group_size * t.inner;
    const bool batch = simple && t.scale && (t.outer == 1 || one_group) && (one_group ? n % epl == 0 : t.inner % epl == 0) &&
                       n < ((int64_t)1 << 31) && aligned16(t.in) && aligned16(t.out) && !(one_group && t.outer != 1 && t.C > 1);
    if (!batch) {
      const int r = dmxq_fixed_qdq(t.in, t.out, dtype_in, dtype_out, t.outer, t.C, t.inner, precision, fraction, clamp, symmetric,
                                   rounding, t.scale, t.zero_point, group_size, seed + (uint64_t)i, stream);
      if (r != DMXQ_OK) rc = r;
      continue;
    }
    const int64_t nt = (n / epl + TILE - 1) / TILE;
    if (a.n == kFixedMultiMax || tiles + nt >= ((int64_t)1 << 31)) flush();
    a.d[a.n] = FixedMultiDesc{t.in, t.out, t.scale, t.zero_point, n / epl, tiles, make_fastdiv31(grp_elems), 0u};
    a.n++;
    tiles += nt;
  }
  flush();
  return rc;
}
