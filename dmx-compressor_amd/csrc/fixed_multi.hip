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
// Round 5: the batched launch is stream_multi_kernel (stream.hpp) over FixedOp<kUniform, SIMPLE> itself -- the op, tile plans and
// straight-line tile forms of the single-tensor call (elementwise.hip), one op instance per tensor -- instead of a hand-written
// 256 x 4 kernel with a table read and a division per vector (an opt-125m layer's six float32 weights, 56 MB: 13.8 us = 51 %).
// The same file holds the multi-tensor float cast (dmxq_float_qdq_multi: the bias casts of a layer's Linear modules in one launch).
#define DMXQ_EW_PART 9   // (elementwise.hip's ops without its entry points)
#include "elementwise.hip"

extern "C" int dmxq_fixed_qdq(const void* in, void* out, int dtype_in, int dtype_out, int64_t outer, int64_t C, int64_t inner,
                              int precision, int fraction, int clamp, int symmetric, int rounding, const float* scale,
                              const int64_t* zero_point, int64_t group_size, uint64_t seed, void* stream);
extern "C" int dmxq_float_qdq(const void* in, void* out, int dtype_in, int dtype_out, int64_t n, int man_bits, int exp_bits,
                              int exp_bias, int flush_subnormal, int unsigned_abs, int rounding, uint64_t seed, void* stream);

namespace {
template <class OP>
int flush_multi(StreamMultiArgs<OP>& a, int dti, int dto, hipStream_t s) {
  int r = DMXQ_ERR_BAD_ARG;
#define DMXQ_DT(I_, O_) if (dti == I_ && dto == O_) r = launch_stream_multi<I_, O_, OP>(a, s);
  DMXQ_DT(DMXQ_BF16, DMXQ_BF16) DMXQ_DT(DMXQ_F16, DMXQ_F16) DMXQ_DT(DMXQ_F32, DMXQ_F32) DMXQ_DT(DMXQ_BF16, DMXQ_F32)
  DMXQ_DT(DMXQ_F16, DMXQ_F32) DMXQ_DT(DMXQ_F32, DMXQ_BF16) DMXQ_DT(DMXQ_F32, DMXQ_F16)
#undef DMXQ_DT
  a.n = 0;
  return r;
}
}  // namespace

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
  float t_min = (float)(-ldexp(1.0, precision - 1));   // sim_helper.cpp:5-12 fixed_min_max at fraction 0
  const float t_max = (float)(-(double)t_min - 1.0);
  if (symmetric) t_min = (float)((double)t_min + 1.0);
  const FixedFmt f{0, 1, DMXQ_ROUND_NEAREST, t_min, t_max, 0ull};
  using OP = FixedOp<kUniform, true>;
  StreamMultiArgs<OP> a;
  a.n = 0;
  int rc = DMXQ_OK;
  for (int64_t i = 0; i < n_tensors && rc == DMXQ_OK; i++) {
    const dmxq_affine_desc& t = tensors[i];
    const int64_t n = t.outer * t.C * t.inner;
    if (n == 0) continue;
    // batchable: SIMPLE format, affine, every 16-byte vector inside one channel (the kUniform lookup of the single-tensor call) or
    // ONE group for the whole tensor, whole vectors, 16-byte aligned, 31-bit indices
    const bool one_group = t.C <= 1 || group_size >= t.C;
    const bool batch = simple && t.scale && n % epl == 0 && (one_group || t.inner % epl == 0) && n < ((int64_t)1 << 31) &&
                       aligned16(t.in) && aligned16(t.out);
    if (!batch) {
      const int r = dmxq_fixed_qdq(t.in, t.out, dtype_in, dtype_out, t.outer, t.C, t.inner, precision, fraction, clamp, symmetric,
                                   rounding, t.scale, t.zero_point, group_size, seed + (uint64_t)i, stream);
      if (r != DMXQ_OK) rc = r;
      continue;
    }
    if (a.n == StreamMultiArgs<OP>::kMax) { const int r = flush_multi(a, dtype_in, dtype_out, s); if (r != DMXQ_OK) rc = r; }
    const ChannelMap cm = one_group ? make_channel_map(1, n, 1, n) : make_channel_map(t.C, t.inner, group_size, n);
    a.d[a.n] = StreamMultiDesc<OP>{t.in, t.out, n / epl, 0, OP{f, cm, t.scale, t.zero_point}};
    a.n++;
  }
  if (a.n > 0) { const int r = flush_multi(a, dtype_in, dtype_out, s); if (r != DMXQ_OK && rc == DMXQ_OK) rc = r; }
  return rc;
}

extern "C" int dmxq_float_qdq_multi(const dmxq_tensor_desc* tensors, int64_t n_tensors, int dtype_in, int dtype_out, int man_bits,
                                    int exp_bits, int exp_bias, int flush_subnormal, int unsigned_abs, int rounding, uint64_t seed,
                                    void* stream) {
  if (n_tensors < 0 || (n_tensors > 0 && !tensors)) return DMXQ_ERR_BAD_ARG;
  if (!valid_dtype(dtype_in) || !valid_dtype(dtype_out) || !valid_rounding(rounding)) return DMXQ_ERR_BAD_ARG;
  if (exp_bits < 1 || exp_bits > 8 || man_bits < 0) return DMXQ_ERR_BAD_ARG;
  if (man_bits > 22) return DMXQ_ERR_UNSUPPORTED;
  for (int64_t i = 0; i < n_tensors; i++) {
    const dmxq_tensor_desc& t = tensors[i];
    if (t.outer < 0 || t.L < 0 || t.inner < 0) return DMXQ_ERR_BAD_ARG;
    if (t.outer * t.L * t.inner > 0 && (!t.in || !t.out)) return DMXQ_ERR_BAD_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  const int epl = dtype_in == DMXQ_F32 ? 4 : 8;
  const FloatFmt f{man_bits, exp_bits, exp_bias, flush_subnormal ? 1 : 0, unsigned_abs ? 1 : 0, rounding, seed};
  using OP = FloatOp<DMXQ_ROUND_NEAREST>;
  const OP op{f, make_float_fast(f.man, f.exp_bits, f.bias), make_flush_fast(f.man, f.exp_bits, f.bias, f.flush && !f.unsigned_abs)};
  StreamMultiArgs<OP> a;
  a.n = 0;
  int rc = DMXQ_OK;
  for (int64_t i = 0; i < n_tensors && rc == DMXQ_OK; i++) {
    const dmxq_tensor_desc& t = tensors[i];
    const int64_t n = t.outer * t.L * t.inner;
    if (n == 0) continue;
    // batchable: nearest rounding (the other modes draw / branch per call), whole vectors, 16-byte aligned, 31-bit indices
    const bool batch = rounding == DMXQ_ROUND_NEAREST && n % epl == 0 && n < ((int64_t)1 << 31) && aligned16(t.in) && aligned16(t.out);
    if (!batch) {
      const int r = dmxq_float_qdq(t.in, t.out, dtype_in, dtype_out, n, man_bits, exp_bits, exp_bias, flush_subnormal, unsigned_abs,
                                   rounding, seed + (uint64_t)i, stream);
      if (r != DMXQ_OK) rc = r;
      continue;
    }
    if (a.n == StreamMultiArgs<OP>::kMax) { const int r = flush_multi(a, dtype_in, dtype_out, s); if (r != DMXQ_OK) rc = r; }
    a.d[a.n] = StreamMultiDesc<OP>{t.in, t.out, n / epl, 0, op};
    a.n++;
  }
  if (a.n > 0) { const int r = flush_multi(a, dtype_in, dtype_out, s); if (r != DMXQ_OK && rc == DMXQ_OK) rc = r; }
  return rc;
}

// One launch for BOTH parameter casts of a layer (round 5): the affine integer casts of its weights and the float casts of its biases.
extern "C" int dmxq_fixed_float_qdq_multi(const dmxq_affine_desc* fixed, int64_t n_fixed, int precision, int fraction, int clamp, int symmetric,
                                          int rounding_fixed, int64_t group_size, const dmxq_tensor_desc* flt, int64_t n_float, int man_bits,
                                          int exp_bits, int exp_bias, int flush_subnormal, int unsigned_abs, int rounding_float, int dtype,
                                          uint64_t seed, void* stream) {
  if (n_fixed < 0 || n_float < 0 || (n_fixed > 0 && !fixed) || (n_float > 0 && !flt) || !valid_dtype(dtype)) return DMXQ_ERR_BAD_ARG;
  // every argument error of EITHER list before the first launch (the two fallback calls validate their own list only)
  if (!valid_rounding(rounding_fixed) || !valid_rounding(rounding_float) || precision < 1 || group_size < 1 || exp_bits < 1 || exp_bits > 8 ||
      man_bits < 0)
    return DMXQ_ERR_BAD_ARG;
  if (man_bits > 22) return DMXQ_ERR_UNSUPPORTED;
  for (int64_t i = 0; i < n_fixed; i++) {
    const dmxq_affine_desc& t = fixed[i];
    if (t.outer < 0 || t.C < 0 || t.inner < 0 || (t.scale == nullptr) != (t.zero_point == nullptr)) return DMXQ_ERR_BAD_ARG;
    if (t.outer * t.C * t.inner > 0 && (!t.in || !t.out)) return DMXQ_ERR_BAD_ARG;
  }
  for (int64_t i = 0; i < n_float; i++) {
    const dmxq_tensor_desc& t = flt[i];
    if (t.outer < 0 || t.L < 0 || t.inner < 0) return DMXQ_ERR_BAD_ARG;
    if (t.outer * t.L * t.inner > 0 && (!t.in || !t.out)) return DMXQ_ERR_BAD_ARG;
  }
  using OPA = FixedOp<kUniform, true>;
  using OPB = FloatOp<DMXQ_ROUND_NEAREST>;
  using Args = StreamMulti2Args<OPA, OPB>;
  const int epl = dtype == DMXQ_F32 ? 4 : 8;
  bool combined = n_fixed >= 1 && n_float >= 1 && n_fixed <= Args::kMaxA && n_float <= Args::kMaxB && n_fixed + n_float <= Args::kMaxTensors &&
                  fraction == 0 && clamp && rounding_fixed == DMXQ_ROUND_NEAREST && precision >= 1 && precision <= 22 && group_size >= 1 &&
                  rounding_float == DMXQ_ROUND_NEAREST && exp_bits >= 1 && exp_bits <= 8 && man_bits >= 0 && man_bits <= 22;
  for (int64_t i = 0; combined && i < n_fixed; i++) {
    const dmxq_affine_desc& t = fixed[i];
    if (t.outer < 0 || t.C < 0 || t.inner < 0) return DMXQ_ERR_BAD_ARG;
    const int64_t n = t.outer * t.C * t.inner;
    const bool one_group = t.C <= 1 || group_size >= t.C;
    combined = n > 0 && t.in && t.out && t.scale && t.zero_point && n % epl == 0 && (one_group || t.inner % epl == 0) && n < ((int64_t)1 << 31) &&
               aligned16(t.in) && aligned16(t.out);
  }
  for (int64_t i = 0; combined && i < n_float; i++) {
    const dmxq_tensor_desc& t = flt[i];
    if (t.outer < 0 || t.L < 0 || t.inner < 0) return DMXQ_ERR_BAD_ARG;
    const int64_t n = t.outer * t.L * t.inner;
    combined = n > 0 && t.in && t.out && n % epl == 0 && n < ((int64_t)1 << 31) && aligned16(t.in) && aligned16(t.out);
  }
  if (combined) {
    float t_min = (float)(-ldexp(1.0, precision - 1));
    const float t_max = (float)(-(double)t_min - 1.0);
    if (symmetric) t_min = (float)((double)t_min + 1.0);
    const FixedFmt f{0, 1, DMXQ_ROUND_NEAREST, t_min, t_max, 0ull};
    const FloatFmt ff{man_bits, exp_bits, exp_bias, flush_subnormal ? 1 : 0, unsigned_abs ? 1 : 0, DMXQ_ROUND_NEAREST, seed};
    const OPB opb{ff, make_float_fast(ff.man, ff.exp_bits, ff.bias), make_flush_fast(ff.man, ff.exp_bits, ff.bias, ff.flush && !ff.unsigned_abs)};
    Args a;
    a.nA = (int)n_fixed;
    a.nB = (int)n_float;
    for (int64_t i = 0; i < n_fixed; i++) {
      const dmxq_affine_desc& t = fixed[i];
      const int64_t n = t.outer * t.C * t.inner;
      const bool one_group = t.C <= 1 || group_size >= t.C;
      const ChannelMap cm = one_group ? make_channel_map(1, n, 1, n) : make_channel_map(t.C, t.inner, group_size, n);
      a.a[i] = StreamMultiDesc<OPA>{t.in, t.out, n / epl, 0, OPA{f, cm, t.scale, t.zero_point}};
    }
    for (int64_t i = 0; i < n_float; i++) a.b[i] = StreamMultiDesc<OPB>{flt[i].in, flt[i].out, flt[i].outer * flt[i].L * flt[i].inner / epl, 0, opb};
    int r = DMXQ_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == DMXQ_F32) r = launch_stream_multi2<DMXQ_F32, DMXQ_F32, OPA, OPB>(a, s);
    else if (dtype == DMXQ_BF16) r = launch_stream_multi2<DMXQ_BF16, DMXQ_BF16, OPA, OPB>(a, s);
    else r = launch_stream_multi2<DMXQ_F16, DMXQ_F16, OPA, OPB>(a, s);
    if (r != DMXQ_ERR_UNSUPPORTED) return r;
  }
  // everything else: the two multi-tensor calls, which give every tensor they cannot batch a launch of its own
  const int r1 = dmxq_fixed_qdq_multi(fixed, n_fixed, dtype, dtype, precision, fraction, clamp, symmetric, rounding_fixed, group_size, seed, stream);
  if (r1 != DMXQ_OK) return r1;
  return dmxq_float_qdq_multi(flt, n_float, dtype, dtype, man_bits, exp_bits, exp_bias, flush_subnormal, unsigned_abs, rounding_float, seed, stream);
}
