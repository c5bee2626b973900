/* include/dmxq.h — C ABI of libdmxq.so, the MI355X (gfx950) fake-quantisation / sparsity operator library.
 *
 * This is the drop-in boundary for the hot path of d-matrix-ai/dmx-compressor (reference paths below are
 * relative to /root/reference/src/dmx/compressor/).  The reference's native seam is a pair of pybind
 * modules, `quant_cpu` / `quant_cuda`, picked per call by quant/quant_function.py:38-43; each entry point
 * here replaces one family of those pybind functions *and* the Python loop that drives it, so that one call
 * (= one kernel launch) handles a whole tensor.
 *
 * Conventions (all entry points):
 *   - plain C: raw DEVICE pointers + explicit 64-bit sizes, no torch types, never throws, no state for the caller to manage
 *     (the reductions -- dmxq_group_minmax, dmxq_channel_maxabs, dmxq_histc -- keep 64 KiB of flag words per device, allocated on
 *     first use and never freed: ONE wave of the kernel initialises the outputs and publishes the launch's epoch, the workgroups wait
 *     for it before their atomics, instead of a fill launch in front of every call.  Which wave is decided by a claim on an election
 *     word -- an extra workgroup without data volunteers at once, any waiting workgroup takes the job over after a bounded number of
 *     polls -- so the protocol does not depend on dispatch order or on what else occupies the GPU; csrc/reduce.hip "init gate".
 *     While `stream` is being captured into a graph, and with DMXQ_NO_INIT_GATE set in the environment, they use the fill launch,
 *     so a captured call is replayable.  HARDWARE ASSUMPTION of the default build: the publishing wave orders its plain `sc1` stores
 *     (the identities) before its flag store with `s_waitcnt vmcnt(0)`, and the waiting waves read the flag with agent-scope atomics --
 *     i.e. it relies on gfx950 acknowledging an `sc1` store only once it is visible at the agent coherence point (all eight XCD L2s)
 *     and on atomics never being served from a stale cached line.  That is behaviour of this hardware, not a guarantee of the LLVM
 *     AMDGPU memory model; `-DDMXQ_GATE_FENCES=1` builds the formal release / acquire form (measured slower than the fill launch it
 *     replaces, profiles/r05_gate_ab.txt; built as lib/libdmxq_gate_fences.so and put through the same stress tests, tests/test_gpu_round6.py), and DMXQ_NO_INIT_GATE=1 removes the protocol at run time);
 *   - caller-allocated outputs (the reference allocates with zeros_like and returns a new tensor,
 *     quant_cuda.cpp:116-139 — the host mirror keeps that ownership contract above this ABI);
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); launches are asynchronous;
 *   - as for any HIP launch, the device that owns the pointers and the stream must be CURRENT on the calling thread
 *     (the host mirror and the torch extension switch to the tensor's device around every call and restore it);
 *   - return value: DMXQ_OK, or an error code (see dmxq_status_string); nothing is launched on error;
 *   - tensors are described as a contiguous [outer, L, inner] (or [outer, C, inner]) view: `L`/`C` is the
 *     extent of the blocked / channel dimension, `inner` the product of the dimensions after it
 *     (block_dim=-1  <=>  inner = 1).  in and out may alias exactly (in-place) but must not partially overlap.
 */
#ifndef DMXQ_H
#define DMXQ_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum { DMXQ_F32 = 0, DMXQ_F16 = 1, DMXQ_BF16 = 2 } dmxq_dtype;

/* Rounding modes, numbered like the reference's `enum Mode` (quant/quant_cpu/quant_cpu.cpp:9-15). */
typedef enum { DMXQ_ROUND_UP = 0, DMXQ_ROUND_DOWN = 1, DMXQ_ROUND_NEAREST = 2, DMXQ_ROUND_STOCHASTIC = 3 } dmxq_rounding;

typedef enum {
  DMXQ_OK = 0,
  DMXQ_ERR_BAD_ARG = 1,     /* null pointer, negative size, misaligned/invalid enum ... */
  DMXQ_ERR_UNSUPPORTED = 2, /* parameter outside what the reference defines (e.g. mantissa bits = 23: UB there) */
  DMXQ_ERR_LAUNCH = 3,      /* hipGetLastError() != hipSuccess after the launch */
  DMXQ_ERR_PENDING = 4      /* a HIP error of ANOTHER user of this host thread was already pending before the call (HIP keeps one
                               "last error" per thread): the launches were issued but cannot be verified, so the outputs must not be
                               trusted; the foreign error is left in place for its owner to collect (hipGetLastError) */
} dmxq_status;

const char* dmxq_status_string(int status);
/* ABI version: bumped on any signature change or addition.  4 = round 5: + dmxq_float_qdq_multi, dmxq_fixed_float_qdq_multi; 3 = round 4: + dmxq_weight_hypernet_multi,
 * dmxq_unary_cast_table, dmxq_lut16_apply.  Nothing was ever removed or changed: a caller built against version n runs on any library >= n. */
int dmxq_abi_version(void);

/* Block floating point Q->DQ ("BFP[p|8]{B}", MXINT).
 * Replaces: numerical/format.py:304-343 BlockFloatingPoint.cast (split / per-chunk / cat loop)
 *           -> quant/quant_function.py:87-117 block_quantize -> quant_cpu.cpp:239-311 / quant_cuda/quant.cu:14-112
 *           + format.py:349-372 make_mantissa_asymmetric (symmetric = 0)
 *           + the `.float()` / `.to(physical_dtype)` round trip of numerical/cast.py:262,306 (dtype_in/dtype_out).
 * Blocks are `block_size` consecutive indices along L (stride `inner`); a ragged last block is allowed
 * (torch.split semantics).  block_size == 1 takes the reference's float_quantize detour (format.py:312-320).
 * precision = total mantissa bits incl. sign ("8" in BFP[8|8]); 2 <= precision <= 22 when block_size > 1.
 * symmetric: 1 = symmetric codes; 0 = the asymmetric format "(_N)" = symmetric pass + make_mantissa_asymmetric, which is
 * all the Python layer ever asks for (format.py:332 forces the native flag to true); DMXQ_BFP_ASYM_NATIVE (2) = the
 * native `symmetric = false` of the pybind seam block_quantize_*(a, wl, dim, symmetric) (quant_cpu.cpp:247-253: an
 * element equal to -max whose maximum has its top 7 mantissa bits set takes the next exponent), float32 only. */
#define DMXQ_BFP_ASYM_NATIVE 2
int dmxq_bfp_qdq(const void* in, void* out, int dtype_in, int dtype_out, int64_t outer, int64_t L, int64_t inner,
                 int64_t block_size, int precision, int rounding, int symmetric, uint64_t seed, void* stream);

/* Multi-tensor BFP Q->DQ: exactly the result of one dmxq_bfp_qdq call per tensor (same dtype pair and format for all),
 * in as few launches as possible.  Replaces the per-module loop of modeling/model.py fold_weights_and_biases /
 * DmxModule.weight_hypernet over MANY small weights (opt-125m: 73 Linear weights of 768x768 .. 3072x768, each of which
 * is launch-bound on its own: ~4 us per launch for 0.4-1.6 us of streaming).  `tensors` is a HOST array; tensors that
 * are flat row-blocked (inner == 1, L % block_size == 0, 16-byte aligned, nearest rounding) are packed 48 to a launch
 * over a concatenated tile space, the others get their own launch (stochastic rounding: seed + index). */
typedef struct { const void* in; void* out; int64_t outer, L, inner; } dmxq_tensor_desc;
int dmxq_bfp_qdq_multi(const dmxq_tensor_desc* tensors, int64_t n_tensors, int dtype_in, int dtype_out,
                       int64_t block_size, int precision, int rounding, int symmetric, uint64_t seed, void* stream);

/* Introspection (no launch): writes into buf (NUL-terminated, at most buf_len bytes) the kernel and tile geometry that
 * dmxq_bfp_qdq would launch for these arguments; `aligned` = both pointers are 16-byte aligned.  bench.py reports it
 * next to the roofline numbers, so that the named kernel is the dispatcher's own choice, not a constant. */
int dmxq_bfp_qdq_describe(int dtype_in, int dtype_out, int64_t outer, int64_t L, int64_t inner, int64_t block_size,
                          int precision, int rounding, int symmetric, int aligned, char* buf, int64_t buf_len);

/* Scaled block floating point Q->DQ ("SBFP<XP[p,0](CSN)><FP[0|e|m,bias](FN)>{B}", e.g. SBFP12_16 weight storage).
 * Replaces: numerical/format.py:453-479 ScaledBlockFloatingPoint.cast.  Per block: s = max|x| / (2^(p-1)-1);
 * y = fixed(x / s; p, 0, clamp, symmetric, nearest) * |float(s; man, exp, bias, flush)| where s > 0, else x.
 * Same [outer, L, inner] / ragged-tail conventions as dmxq_bfp_qdq. */
int dmxq_sbfp_qdq(const void* in, void* out, int dtype_in, int dtype_out, int64_t outer, int64_t L, int64_t inner,
                  int64_t block_size, int precision, int clamp, int symmetric, int scaler_man_bits, int scaler_exp_bits,
                  int scaler_exp_bias, int scaler_flush_subnormal, void* stream);

/* MX floating point Q->DQ ("MXFP8[E4M3]{32}", ...): low-bit float elements with a power-of-two (E8M0) block scale.
 * Replaces: numerical/format.py:545-564 MXFP.cast.  Per block: scale = 2^floor(log2 max|x|) / 2^(2^(e-1));
 * y = float(x / scale; man, exp, bias = 2^(e-1)-1, no flush, nearest) * scale.  Blocks run along L as in
 * dmxq_bfp_qdq (the reference's `cat(dim=block_dim)` slip is not reproduced); an all-zero block stays zero (the
 * reference produces NaN through log2(0)). */
int dmxq_mxfp_qdq(const void* in, void* out, int dtype_in, int dtype_out, int64_t outer, int64_t L, int64_t inner,
                  int64_t block_size, int man_bits, int exp_bits, void* stream);

/* Packed on-wire BFP: int8 mantissa codes + one uint8 shared exponent per block (the com.microsoft::QuantizeBFP /
 * DequantizeBFP pair the reference's export names, numerical/cast.py:34-55; type ids numerical/onnx.py).
 * x is [rows, L] contiguous, blocks of block_size along L (ragged tail allowed); mant: int8[rows*L];
 * exps: uint8[rows * ceil(L / block_size)] = biased fp32 exponent of the block maximum.  precision <= 8.
 * dmxq_bfp_unpack(dmxq_bfp_pack(x)) == dmxq_bfp_qdq(x) bit for bit for blocks with a normal finite maximum
 * (denormal / zero maxima pack to zeros, Inf/NaN maxima to exps = 255 -> NaN). */
int dmxq_bfp_pack(const void* in, int dtype_in, int8_t* mant, uint8_t* exps, int64_t rows, int64_t L,
                  int64_t block_size, int precision, int symmetric, void* stream);
int dmxq_bfp_unpack(const int8_t* mant, const uint8_t* exps, void* out, int dtype_out, int64_t rows, int64_t L,
                    int64_t block_size, int precision, void* stream);

/* Low-bit floating point Q->DQ ("FP[s|e|m,bias](F|_ N|S)").
 * Replaces: numerical/format.py:208-233 FloatingPoint.cast -> quant_function.py:120-152 float_quantize
 *           -> quant_cpu.cpp:359-402 / quant_cuda/float_kernel.cu.  0 <= man_bits <= 22 (23 is UB in the
 *           reference; the host mirror treats FP32 as the identity), 1 <= exp_bits <= 8.
 * unsigned_abs != 0 applies the final `.abs()` of sign-less formats (format.py:233). */
int dmxq_float_qdq(const void* in, void* out, int dtype_in, int dtype_out, int64_t n, int man_bits, int exp_bits,
                   int exp_bias, int flush_subnormal, int unsigned_abs, int rounding, uint64_t seed, void* stream);

/* Multi-tensor twin of dmxq_float_qdq (round 5): exactly the result of one dmxq_float_qdq call per tensor (same dtype pair and format for
 * all; tensor i holds outer * L * inner elements), in as few launches as possible.  Replaces the per-module bias casts of a layer
 * (modeling/nn/core.py:191-203 `_bias = self.bias_cast(self.bias)`, one launch per Linear per forward while biases are not folded:
 * an opt-125m decoder layer has six of 768 .. 3072 elements).  `tensors` is a HOST array.  Batched: nearest rounding, whole 16-byte
 * vectors, 16-byte aligned, < 2^31 elements; every other tensor gets its own launch (stochastic rounding: seed + index). */
int dmxq_float_qdq_multi(const dmxq_tensor_desc* tensors, int64_t n_tensors, int dtype_in, int dtype_out, int man_bits, int exp_bits,
                         int exp_bias, int flush_subnormal, int unsigned_abs, int rounding, uint64_t seed, void* stream);

/* Fixed point Q->DQ ("XP[p,f](C|_ S|_ R)") with the optional affine wrapper fused in.
 * Replaces: numerical/format.py:134-142 FixedPoint.cast -> quant_function.py:47-84 fixed_point_quantize
 *           -> quant_cpu.cpp:127-209 (CPU rounding: half-to-even through sim_helper.cpp:14-21), and
 *           numerical/cast.py:278-296:  x/sc + zp -> cast -> (x - zp)*sc.
 * The tensor is [outer, C, inner]; channel c uses scale[c / group_size], zero_point[c / group_size]
 * (cast.py:281-292 repeat_interleave).  scale == NULL: no affine.  per-tensor: C = 1, group_size = 1. */
int dmxq_fixed_qdq(const void* in, void* out, int dtype_in, int dtype_out, int64_t outer, int64_t C, int64_t inner,
                   int precision, int fraction, int clamp, int symmetric, int rounding, const float* scale,
                   const int64_t* zero_point, int64_t group_size, uint64_t seed, void* stream);

/* Multi-tensor twin of dmxq_fixed_qdq: exactly the result of one dmxq_fixed_qdq call per tensor (same dtype pair, format
 * and group_size for all; each tensor its own scale / zero-point arrays), in as few launches as possible.  Replaces the
 * per-module loop over the INT8 group-quantised Linear weights of a whole model (numerical/cast.py:278-296 once per module;
 * opt-125m: 73 weights).  `tensors` is a HOST array.  Batched (20 tensors per launch, each with its own op instance in the argument block): integer formats (fraction 0, clamped,
 * nearest), outer == 1 with row slabs of group_size channels or a single group, inner a multiple of the 16-byte vector,
 * < 2^31 elements; every other tensor gets its own launch (stochastic rounding: seed + index). */
typedef struct { const void* in; void* out; const float* scale; const int64_t* zero_point; int64_t outer, C, inner; } dmxq_affine_desc;
int dmxq_fixed_qdq_multi(const dmxq_affine_desc* tensors, int64_t n_tensors, int dtype_in, int dtype_out, int precision,
                         int fraction, int clamp, int symmetric, int rounding, int64_t group_size, uint64_t seed, void* stream);

/* dmxq_fixed_qdq_multi AND dmxq_float_qdq_multi in ONE launch (round 5): a layer's affine integer weight casts and its float bias casts
 * (modeling/nn/core.py:178-203: `_weight` and `_bias` of every module, every forward while nothing is folded), one dtype in and out for all.
 * Exactly the results of the two calls.  Combined: at most 12 + 12 tensors (21 in all), 32 MiB in all, every tensor batchable by its own
 * multi call's rules, nearest rounding on both sides; any other set is run as those two calls. */
int dmxq_fixed_float_qdq_multi(const dmxq_affine_desc* fixed, int64_t n_fixed, int precision, int fraction, int clamp, int symmetric,
                               int rounding_fixed, int64_t group_size, const dmxq_tensor_desc* flt, int64_t n_float, int man_bits,
                               int exp_bits, int exp_bias, int flush_subnormal, int unsigned_abs, int rounding_float, int dtype,
                               uint64_t seed, void* stream);

/* N:M structured-sparsity mask ("BTOPK{K:M,dim}") and its application.
 * Replaces: sparse.py:163-180 BlockTopK.forward (argsort + scatter) and sparse.py:300 `x * mask`.
 * Groups are M consecutive indices along L (stride inner); L % M == 0 required (sparse.py:166-168).
 * Rule: within a group, rank_i = #{j : s_j < s_i or (s_j == s_i and j < i)}, NaN ranks highest;
 * mask_i = 1 iff rank_i >= M-K (stable ascending argsort, first M-K zeroed).
 * mask_out (dtype_mask, float mask of 0/1) and/or y_out = x * mask (dtype_y) may be NULL.  M <= 64. */
int dmxq_nm_mask(const void* score, int dtype_score, const void* x, int dtype_x, void* mask_out, int dtype_mask,
                 void* y_out, int dtype_y, int64_t outer, int64_t L, int64_t inner, int K, int M, void* stream);

/* Unstructured top-k mask ("TOPK{density}") and its application: the n_zero lowest scores of the flattened tensor are
 * zeroed.  Replaces: sparse.py:109-123 TopK.forward (argsort + scatter; n_zero = int(n * (1 - density))) and
 * sparse.py:300 `x * mask`.  Order: ascending, -0 == +0, NaN largest; scores equal to the threshold value are zeroed
 * lowest index first (the reference's unstable argsort leaves that choice undefined).  mask_out (float 0/1 in
 * dtype_mask) and/or y_out = x * mask may be NULL.  workspace: dmxq_topk_workspace_bytes(n) bytes of device memory,
 * 8-byte aligned, contents irrelevant before and after.  n < 2^32. */
int64_t dmxq_topk_workspace_bytes(int64_t n);
int dmxq_topk_mask(const void* score, int dtype_score, const void* x, int dtype_x, void* mask_out, int dtype_mask,
                   void* y_out, int dtype_y, int64_t n, int64_t n_zero, void* workspace, void* stream);

/* Bernoulli supermask ("BERN"): mask[i] = 1 with probability score[i] (scores in [0, 1]), else 0.
 * Replaces: sparse.py:201-221 Bernoulli.forward (torch.bernoulli on the global generator; here a counter-based
 * stream keyed by `seed` and the element index, reproducible, statistically equivalent). */
int dmxq_bernoulli_mask(const void* score, void* mask_out, int dtype_score, int dtype_mask, int64_t n, uint64_t seed,
                        void* stream);

/* Per-group min/max over slabs of `group_size` channels (MinMaxObserver on torch.split slabs).
 * Replaces: numerical/cast.py:179-226 _observer_step + numerical/observer.py:173-193.
 * mn/mx: float[ceil(C/group_size)] device buffers (fully overwritten). */
int dmxq_group_minmax(const void* in, int dtype_in, int64_t outer, int64_t C, int64_t inner, int64_t group_size,
                      float* mn, float* mx, void* stream);

/* The same reduction folded INTO running values: mn[g] = min(mn[g], min over group g), mx[g] = max(mx[g], max over group g), in ONE
 * launch (no initialising launch: mn / mx must hold valid values, +inf / -inf before the first observation).  Replaces
 * numerical/observer.py:173-193 MinMaxObserver.forward as a whole -- the two reductions AND `min_val = torch.min(x_min, min_val)` /
 * `max_val = torch.max(...)` (4 launches with dmxq_group_minmax, ~10 per group in the reference).  Exact and order-independent (integer
 * atomics on the float bit patterns). */
int dmxq_group_minmax_accumulate(const void* in, int dtype_in, int64_t outer, int64_t C, int64_t inner, int64_t group_size,
                                 float* mn, float* mx, void* stream);

/* (min,max) -> (scale, zero_point).  Replaces numerical/observer.py:59-115 _calculate_qparams. */
int dmxq_qparams(const float* mn, const float* mx, int64_t n_groups, int qmin, int qmax, int symmetric_qscheme,
                 float* scale, int64_t* zero_point, void* stream);

/* torch.histc of a flat tensor: hist[b] = #{x : bin(x) == b}, bin(x) = (int64)((x - lo) * bins / (hi - lo)) in
 * fp32, x == hi counted in the last bin, x outside [lo, hi] and NaN dropped (ATen's CPU histc, which the
 * reference's HistogramObserver calls).  Replaces: numerical/observer.py:470-472 and :489-491.
 * Requires finite lo < hi (the "lo == hi means the data's own range" convention of torch.histc is resolved by
 * the caller) and bins <= 8192.  hist: float[bins] device buffer, fully overwritten. */
int dmxq_histc(const void* in, int dtype_in, int64_t n, int64_t bins, float lo, float hi, float* hist, void* stream);

/* Per-channel max|x| (SmoothQuant).  Replaces numerical/smoothquant.py:285-299 _maxabs. out: float[C]. */
int dmxq_channel_maxabs(const void* in, int dtype_in, int64_t outer, int64_t C, int64_t inner, float* out, void* stream);

/* SmoothQuant scale = clamp(a^alpha / clamp(b, min)^(1-alpha), min).  Replaces smoothquant.py:301-321. */
int dmxq_smoothquant_scale(const float* a_maxabs, const float* b_maxabs, int64_t C, float alpha, float scale_min,
                           float* scale, void* stream);

/* y = x * s[c] (divide = 0) or x / s[c] (divide = 1) along the channel dim of [outer, C, inner], computed in
 * fp32 and rounded once to dtype_out.  Replaces smoothquant.py:255-283 scale_a / scale_b. */
int dmxq_scale_channels(const void* in, void* out, int dtype_in, int dtype_out, int64_t outer, int64_t C,
                        int64_t inner, const float* scale, int divide, void* stream);

/* Fused weight hypernet: N:M mask -> SmoothQuant weight scale -> BFP Q->DQ in one pass over a [rows, L] weight.
 * Replaces the chain of modeling/nn/core.py:178-198 (weight_sparsifier -> smoothquant.scale_weight ->
 * weight_cast) that DmxModule re-runs on every forward, for the Linear layout (mask groups, scale channels and
 * BFP blocks all along the contiguous last dim).  M = 0: no mask (score ignored); sq_scale = NULL: no scaling
 * (else float[L], one per input channel).  Results are bit-identical to the unfused chain, whose intermediate
 * dtypes are reproduced (see csrc/hypernet.hip).  Returns DMXQ_ERR_UNSUPPORTED for geometries it does not fuse
 * (L % block_size != 0, block_size not a power of two in [8,512], M not in {0,2,4,8}, unaligned pointers,
 * dtype combination outside {w,score,out}: the caller then runs the unfused ops). */
int dmxq_weight_hypernet(const void* w, int dtype_w, const void* score, int dtype_score, int K, int M,
                         const float* sq_scale, void* out, int dtype_out, int64_t rows, int64_t L, int64_t block_size,
                         int precision, int symmetric, void* stream);

/* Multi-tensor form of dmxq_weight_hypernet: exactly the result of one dmxq_weight_hypernet call per tensor (one dtype triple, one
 * N:M pattern, one BFP format for all), in as few launches as possible -- up to 32 weights per launch over a concatenated tile
 * space.  Replaces the per-module loop of modeling/nn/core.py:178-198 (DmxModule.weight_hypernet, once per module and forward) and
 * of modeling/model.py fold_weights_and_biases over the Linear weights of a layer: under row sharding over 8 GPUs a
 * rank's seven Llama-3-8B shards are 1-15 MB each, launch-bound one by one.  `tensors` is a HOST array; sq_scale is float[L] for
 * every tensor or NULL for every tensor; score is ignored when M == 0.  All-or-nothing: DMXQ_ERR_UNSUPPORTED (nothing launched)
 * when any tensor has a geometry dmxq_weight_hypernet does not fuse -- the caller then goes tensor by tensor. */
typedef struct { const void* w; const void* score; const float* sq_scale; void* out; int64_t rows, L; } dmxq_hypernet_desc;
int dmxq_weight_hypernet_multi(const dmxq_hypernet_desc* tensors, int64_t n_tensors, int dtype_w, int dtype_score, int K, int M,
                               int dtype_out, int64_t block_size, int precision, int symmetric, void* stream);

/* The same chain for layouts whose blocked dimension is not the contiguous one: w viewed as [outer, L, inner] with the N:M groups,
 * the SmoothQuant channels and the BFP blocks all along L (stride inner) -- Conv1d / Conv2d weights [out, in, k...], whose weight
 * cast, sparsifier and SmoothQuant axis are dim 1 (modeling/nn/torch_modules.py:582-585, 674-677; the chain is
 * modeling/nn/core.py:178-198 as above).  Any block size >= 2 and a ragged last block (torch.split); M in {0, 2, 4, 8} dividing
 * block_size; every dtype combination.  Bit-identical to the unfused chain.  These weights are small: the call is launch-bound and
 * the point is one launch instead of three. */
int dmxq_weight_hypernet_strided(const void* w, int dtype_w, const void* score, int dtype_score, int K, int M,
                                 const float* sq_scale, void* out, int dtype_out, int64_t outer, int64_t L, int64_t inner,
                                 int64_t block_size, int precision, int symmetric, void* stream);

/* The activation twin of the fused weight path: SmoothQuant input scaling -> BFP input cast in one pass over x[rows, L]
 * (channels and blocks along the contiguous last dim).  out = BFP_QDQ(x / sq_scale[c]); the quotient is an IEEE fp32 division
 * and stays fp32 -- torch's promotion of (input dtype, fp32 scale), i.e. what `a / scale` returns in
 * numerical/smoothquant.py:255-268 -- which is then the dtype CastTo sees and returns (numerical/cast.py:262,306), so
 * dtype_out must be DMXQ_F32.  Replaces the first two steps of DmxModule.forward (modeling/nn/core.py:228-232:
 * smoothquant.scale_input -> input_casts) for Linear-layout modules: 6 B/element (bf16 in) instead of 14 (2+4, then 4+4).
 * Bit-identical to dmxq_scale_channels(divide) followed by dmxq_bfp_qdq.  DMXQ_ERR_UNSUPPORTED as dmxq_weight_hypernet. */
int dmxq_input_hypernet(const void* x, int dtype_x, const float* sq_scale, void* out, int dtype_out, int64_t rows, int64_t L,
                        int64_t block_size, int precision, int symmetric, void* stream);

/* A binary DmxModule (ResAdd, Mul) in one pass: out = cast_out(cast_a(a) op cast_b(b)), all three tensors of one dtype, the op
 * evaluated as torch does (fp32 arithmetic on the widened operands, one RNE rounding to the tensor dtype).  Replaces the four
 * launches of modeling/nn/core.py:228-264 for such a module (CastToDict.forward on both inputs, numerical/cast.py:59-86; the op;
 * the output cast): 6 B/element instead of 18.  A cast is described by its FloatingPoint format (numerical/format.py:174-233,
 * nearest rounding, signed, man_bits <= 22); NULL or exp_bits == 0 = SAME.  Two forms: when every cast is RANGE-ONLY for a 16-bit
 * tensor dtype (bf16 with man_bits >= 7, fp16 with man_bits >= 10; subnormals flushed: the FLOAT16-style formats of the BASIC rules)
 * everything happens on the packed 16-bit words; otherwise (casts that round, float32 tensors) per element in fp32 with the
 * arithmetic of dmxq_float_qdq.  DMXQ_ERR_UNSUPPORTED: n not a whole number of 16-byte vectors, unaligned pointers (the caller runs
 * the unfused ops).  Bit-identical to that chain; a / b may alias out.  (The host mirror maps the reference's pass-through of a
 * dtype's own format, numerical/format.py:209-212, to SAME before calling.) */
typedef struct { int man_bits, exp_bits, exp_bias, flush_subnormal; } dmxq_float_fmt;
enum { DMXQ_BINARY_ADD = 0, DMXQ_BINARY_MUL = 1 };
int dmxq_binary_cast(const void* a, const void* b, void* out, int dtype, int64_t n, int op, const dmxq_float_fmt* cast_a,
                     const dmxq_float_fmt* cast_b, const dmxq_float_fmt* cast_out, void* stream);

/* A ReLU DmxModule (modeling/nn/torch_modules.py ReLU through core.py:228-264: input cast, F.relu, output cast) in one pass on
 * the two forms of dmxq_binary_cast: out = cast_out(clamp_min(cast_in(x), 0)).  A third of the traffic of the three launches. */
int dmxq_relu_cast(const void* in, void* out, int dtype, int64_t n, const dmxq_float_fmt* cast_in, const dmxq_float_fmt* cast_out,
                   void* stream);

/* dmxq_binary_cast / dmxq_relu_cast followed by the BFP input cast of the ONE module that consumes the result (a Mul feeding the down
 * projection, a ReLU feeding fc2: each module casts its own input, modeling/nn/core.py:228-264 `input_casts`, so the producer's output
 * is read back once more just to be cast) in the producer's launch: out = BFP_QDQ(module(...)) over blocks of `block_size` along
 * contiguous rows of `row_len` elements (symmetric, nearest, `precision` mantissa bits), bit-identical to the two launches; the two
 * forms of dmxq_binary_cast.  DMXQ_ERR_UNSUPPORTED (the caller runs the two launches): row_len not a whole number of
 * blocks, block_size / (16 bytes of elements) not a power of two <= 64, and what dmxq_binary_cast does not take. */
int dmxq_binary_cast_bfp(const void* a, const void* b, void* out, int dtype, int64_t n, int op, const dmxq_float_fmt* cast_a,
                         const dmxq_float_fmt* cast_b, const dmxq_float_fmt* cast_out, int64_t row_len, int64_t block_size, int precision,
                         void* stream);
int dmxq_relu_cast_bfp(const void* in, void* out, int dtype, int64_t n, const dmxq_float_fmt* cast_in, const dmxq_float_fmt* cast_out,
                       int64_t row_len, int64_t block_size, int precision, void* stream);

/* One operand (q or k) of an ApplyRotaryPosEmb DmxModule (modeling/nn/custom_modules.py:142-194) with the module's casts:
 * out = cast_out(rope(cast_x(x), cast_cos(cos), cast_sin(sin))), rope as dmxq_rope below (torch's op-by-op arithmetic in the
 * tensor dtype).  Replaces, per operand, three input casts, the ~6 torch kernels of the exact function and the output cast.
 * The two forms of dmxq_binary_cast (range-only casts of 16-bit tensors on packed words; anything else per element in fp32). */
int dmxq_rope_cast(const void* x, const void* cos_tab, const void* sin_tab, void* out, int dtype, int64_t B, int64_t n1, int64_t n2,
                   int64_t D, int broadcast_over_dim1, const dmxq_float_fmt* cast_x, const dmxq_float_fmt* cast_cos,
                   const dmxq_float_fmt* cast_sin, const dmxq_float_fmt* cast_out, void* stream);

/* Approximator-slot ops.  The reference evaluates the exact torch.nn.functional op and then overwrites it with a
 * vsimd approximation that lives in a private package (functional/approximate.py:9-14, 300-327); with vsimd
 * absent (the public reference) the exact function is the result, and that is what these compute, in fp32.
 * Replaces: ApproximationMixin.approx_forward for GELU (torch_modules.py GELU), Softmax (torch_modules.py:
 * 989-998, `input_clamp` wrapper argument = input_clamp_min, -INFINITY disables) and LayerNorm (:1062-1082).
 * softmax / layernorm act on the contiguous last dim of a [rows, cols] view. weight/bias may be NULL. */
int dmxq_gelu(const void* in, void* out, int dtype_in, int dtype_out, int64_t n, int tanh_form, void* stream);

/* The other per-element function ids of the approximator slot (src/dmx/compressor/__init__.py:108-139).
 * GELU / GELU_TANH / SILU / EXP: the exact torch function in fp32, rounded once to dtype_out (what the reference computes
 * with vsimd absent, functional/approximate.py:300-304; modules torch_modules.py:1559-1576 SiLU, :236-242 Exp).
 * QUICK_GELU: transformers' QuickGELUActivation `x * sigmoid(1.702 * x)` (modeling/nn/custom_modules.py:112-117),
 * evaluated in the INPUT dtype like torch does (three roundings for 16-bit tensors).
 * SILU_EXPERIMENTAL: the reference's one in-repo approximation, functional/functions.py:7-21
 * `relu(x.to(float16)) * scale` (param = scale; dtype_out must be DMXQ_F16), reproduced bit for bit. */
typedef enum {
  DMXQ_UNARY_GELU = 0, DMXQ_UNARY_GELU_TANH = 1, DMXQ_UNARY_SILU = 2, DMXQ_UNARY_QUICK_GELU = 3, DMXQ_UNARY_EXP = 4,
  DMXQ_UNARY_SILU_EXPERIMENTAL = 5
} dmxq_unary_kind;
int dmxq_unary(const void* in, void* out, int dtype_in, int dtype_out, int64_t n, int kind, float param, void* stream);

/* RMSNorm over the contiguous last dim of a [rows, cols] view: y = x * rsqrt(mean(x^2) + eps) * weight, in fp32,
 * rounded once to dtype_out (torch.nn.functional.rms_norm's CPU result).  Replaces: modeling/nn/torch_modules.py:
 * 1144-1170 RMSNorm._forward -> approx_forward(F.rms_norm).  weight may be NULL. */
int dmxq_rmsnorm(const void* in, void* out, int dtype_in, int dtype_out, int64_t rows, int64_t cols, const void* weight,
                 int dtype_w, float eps, void* stream);
int dmxq_softmax(const void* in, void* out, int dtype_in, int dtype_out, int64_t rows, int64_t cols,
                 float input_clamp_min, void* stream);
int dmxq_layernorm(const void* in, void* out, int dtype_in, int dtype_out, int64_t rows, int64_t cols,
                   const void* weight, const void* bias, int dtype_wb, float eps, void* stream);

/* An activation-function or normalisation DmxModule in ONE pass: out = cast_out(f(cast_in(x))), x and out of one dtype.
 * Replaces the three launches of modeling/nn/core.py:228-264 for the modules whose `_forward` is one of these functions
 * (torch_modules.py GELU / SiLU :1559-1576 / Exp :236-242, custom_modules.py:112-117 QuickGELU, Softmax :989-998, LayerNorm
 * :1062-1082, RMSNorm :1144-1170; both casts are FLOAT16 in BASIC mode, src/dmx/compressor/__init__.py:360-455): input CastTo,
 * the exact torch function (functional/approximate.py:300-304 with vsimd absent), output CastTo -- 4 B/element for a 16-bit
 * tensor instead of 12.  The casts are the bit-exact casts of dmxq_float_qdq with CastTo's `.to(dtype)` after each (formats as
 * in dmxq_binary_cast: NULL or exp_bits == 0 = SAME); f is evaluated in fp32 on the cast input and rounded once to the tensor
 * dtype, as torch evaluates it (QUICK_GELU: in the tensor dtype).  The function is floating point: out == cast_out(v) for a v
 * within 1 ulp of the tensor dtype (16-bit tensors) of the correctly rounded f(cast_in(x)); float32 tensors: within the ulps of
 * the unfused functions before the output cast.
 * DMXQ_ERR_UNSUPPORTED (the caller runs the three launches): 16-bit tensors with a cast that is not range-only for the dtype
 * (bf16: man_bits >= 7, fp16: man_bits >= 10, subnormals flushed); unaligned pointers; n not a whole number of 16-byte
 * vectors; rows that do not take the register-resident row kernels (longer than 1024 lane-vectors, or -- norms -- not a
 * multiple of 4 elements); weight / bias are in the row dtype.  kind: DMXQ_UNARY_GELU .. DMXQ_UNARY_EXP. */
int dmxq_unary_cast(const void* in, void* out, int dtype, int64_t n, int kind, float param, const dmxq_float_fmt* cast_in,
                    const dmxq_float_fmt* cast_out, void* stream);

/* The same module on a 16-bit tensor as a TABLE (csrc/lut16.hip): cast_out(f(cast_in(x))) is a function of the 16 input bits.
 * dmxq_unary_cast_table fills table[p] (65,536 entries of 16 bits, device memory owned by the caller) with the module's result for
 * input PATTERN p: the two casts are the bit-exact casts of dmxq_float_qdq (ANY FloatingPoint format with man_bits <= 22, nearest
 * rounding -- rounding casts too, which dmxq_unary_cast refuses on 16-bit tensors) with CastTo's `.to(dtype)`, and f is evaluated in
 * float64 and rounded ONCE to the tensor dtype: the correctly rounded f(cast_in(x)) (QUICK_GELU: its three roundings in the tensor
 * dtype; DMXQ_UNARY_SILU_EXPERIMENTAL: functional/functions.py:7-21 with scale = param).  dmxq_lut16_apply then computes
 * out[i] = table[in[i]] with the table in the LDS: no arithmetic per element.  dtype: DMXQ_BF16 or DMXQ_F16.  The table is the
 * caller's to keep (one per (function, casts, dtype)); the library holds no state.  dmxq_lut16_apply: DMXQ_ERR_UNSUPPORTED when n
 * is not a whole number of 16-byte vectors or a pointer is not 16-byte aligned. */
int dmxq_unary_cast_table(int dtype, int kind, float param, const dmxq_float_fmt* cast_in, const dmxq_float_fmt* cast_out, void* table,
                          void* stream);
int dmxq_lut16_apply(const void* in, void* out, int64_t n, const void* table, void* stream);
int dmxq_softmax_cast(const void* in, void* out, int dtype, int64_t rows, int64_t cols, float input_clamp_min,
                      const dmxq_float_fmt* cast_in, const dmxq_float_fmt* cast_out, void* stream);
/* ... followed by the NEXT module's BFP input cast of the result (the `input_casts` entry of the ActActMatMul / Linear that consumes the
 * probabilities: modeling/nn/core.py:228-264, numerical/format.py:304-343, blocks of `block_size` along the rows, symmetric, nearest):
 * out = BFP_QDQ(cast_out(softmax(cast_in(x)))) in ONE pass, bit-identical to dmxq_softmax_cast followed by dmxq_bfp_qdq(.., block_size,
 * precision, nearest, symmetric).  The consumer then skips that cast.  DMXQ_ERR_UNSUPPORTED: as dmxq_softmax_cast, or rows that are not
 * whole lane-vectors, or a block size that is not 2^k lane-vectors (16-byte vectors; 8-byte ones for 16-bit rows of 4 k elements). */
int dmxq_softmax_cast_bfp(const void* in, void* out, int dtype, int64_t rows, int64_t cols, float input_clamp_min,
                          const dmxq_float_fmt* cast_in, const dmxq_float_fmt* cast_out, int64_t block_size, int precision, void* stream);
int dmxq_layernorm_cast(const void* in, void* out, int dtype, int64_t rows, int64_t cols, const void* weight, const void* bias,
                        float eps, const dmxq_float_fmt* cast_in, const dmxq_float_fmt* cast_out, void* stream);
int dmxq_rmsnorm_cast(const void* in, void* out, int dtype, int64_t rows, int64_t cols, const void* weight, float eps,
                      const dmxq_float_fmt* cast_in, const dmxq_float_fmt* cast_out, void* stream);
/* The norm modules followed by the (identical) BFP input casts of the modules that consume the result -- the q / k / v Linears after a
 * pre-attention norm, gate / up (fc1) after a pre-MLP norm: `input_casts` of modeling/nn/core.py:228-264 -- in ONE launch, as
 * dmxq_softmax_cast_bfp: out = BFP_QDQ(cast_out(norm(cast_in(x)))), bit-identical to the module's launch followed by dmxq_bfp_qdq. */
int dmxq_layernorm_cast_bfp(const void* in, void* out, int dtype, int64_t rows, int64_t cols, const void* weight, const void* bias,
                            float eps, const dmxq_float_fmt* cast_in, const dmxq_float_fmt* cast_out, int64_t block_size, int precision,
                            void* stream);
int dmxq_rmsnorm_cast_bfp(const void* in, void* out, int dtype, int64_t rows, int64_t cols, const void* weight, float eps,
                          const dmxq_float_fmt* cast_in, const dmxq_float_fmt* cast_out, int64_t block_size, int precision, void* stream);

/* APPLY_LLAMA_ROPE: x_embed = (x * cos) + (rotate_half(x) * sin), evaluated in the tensor dtype like torch does (every
 * product and the sum rounded to the dtype: bit-identical to torch's CPU result).  Replaces: modeling/nn/custom_modules.py:
 * 142-172 ApplyRotaryPosEmbBase.forward for ONE of q / k (call it twice).  x, out: [B, n1, n2, D] contiguous, out != x;
 * cos_tab / sin_tab: [B, n2, D] when broadcast_over_dim1 != 0 (unsqueeze_dim = 1: x = [B, heads, S, D]), else [B, n1, D]
 * (unsqueeze_dim = 2: x = [B, S, heads, D]); all of dtype `dtype`.  DMXQ_ERR_UNSUPPORTED: D not a multiple of two 16-byte
 * vectors, unaligned pointers, >= 2^31 elements (the caller keeps torch's own ops). */
int dmxq_rope(const void* x, const void* cos_tab, const void* sin_tab, void* out, int dtype, int64_t B, int64_t n1, int64_t n2,
              int64_t D, int broadcast_over_dim1, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DMXQ_H */
