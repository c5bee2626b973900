// csrc/hypernet_multi.hip -- the fused weight hypernet (hypernet.hip: N:M mask -> SmoothQuant scale -> BFP, one pass) over MANY
// weights in ONE launch.
//
// DmxModule.weight_hypernet (modeling/nn/core.py:178-198) runs once per module and forward; a decoder layer has seven Linear weights
// (Llama-3-8B: q, k, v, o, gate, up, down), and under row sharding over 8 GPUs a rank's shards are [512, 4096],
// [128, 4096] x 2, [512, 4096], [1792, 4096] x 2, [512, 14336]: 1-15 MB each, i.e. 2-5 us of streaming behind a ~1.6 us launch
// floor per weight -- seven launches cost ~36 us where the bytes need ~29.  Here the tile spaces of up to kHnMultiMax weights are
// concatenated (the scheme of bfp_rows_multi_kernel, bfp_rows.hpp): a workgroup finds its tensor by a scalar search over
// descriptors held in kernel arguments and runs ONE tile of kThreads x 2 (4 for sets above 160 M elements) units of hypernet_rows_units on it.  The arithmetic is
// the single-tensor kernel's, unit for unit: results are bit-identical to one dmxq_weight_hypernet call per tensor.
#include "hypernet_rows.hpp"

namespace dmxq {

constexpr int kHnMultiMax = 32;
struct HnMultiDesc {
  const void* w; const void* score; const float* scale; void* out;
  int64_t n_units, L, tile0;   // tile0: first tile of this tensor in the concatenated tile space
  FastDiv31 f_L;
  int small;
};
struct HnMultiArgs { HnMultiDesc d[kHnMultiMax]; int n, K, lpb, wl, asym; };
static_assert(sizeof(HnMultiArgs) <= 3072, "kernel arguments stay well under the 4 KiB limit");

template <int DTW, int DTS, int DTO, int M, bool HAS_SCALE, int UN>
__global__ __launch_bounds__(kThreads) void hypernet_rows_multi_kernel(uint32_t e0, uint32_t e1, uint32_t e2, uint32_t e3, uint32_t e4, uint32_t e5,
                                                                      uint32_t e6, uint32_t e7, uint32_t e8, uint32_t e9, const HnMultiArgs ma) {
  // (round 5, as stream.hpp stream_multi_kernel: the first tiles of tensors 1 .. 10 as preloaded scalar arguments -- a layer's seven weights are
  //  resolved without touching the argument block; larger sets finish the scan in it)
  const uint32_t tile = blockIdx.x;
  int k = (e0 <= tile) + (e1 <= tile) + (e2 <= tile) + (e3 <= tile) + (e4 <= tile) + (e5 <= tile) + (e6 <= tile) + (e7 <= tile) + (e8 <= tile) +
          (e9 <= tile);
  if (k == 10) {
    for (int i = 11; i < ma.n; i++) k = ((uint32_t)ma.d[i].tile0 <= tile) ? i : k;  // tile0 ascending
  }
  const HnMultiDesc& d = ma.d[k];
  const HnArgs a{d.w, d.score, d.scale, d.out, d.n_units, d.L, ma.K, ma.lpb, ma.wl, ma.asym, d.small, d.f_L};
  const int64_t u0 = ((int64_t)tile - d.tile0) * ((int64_t)kThreads * UN) + threadIdx.x;
  if (u0 >= a.n_units) return;  // (whole lane groups of a block leave together: n_units is a multiple of the lanes of a block)
  const bool asym = __builtin_amdgcn_readfirstlane(ma.asym) != 0;
  const int lpb = __builtin_amdgcn_readfirstlane(ma.lpb);
  if (lpb == 8) {  // BFP16_64, the BASIC rule's weight format: compile-time lane count (branch-free DPP maximum)
    if (asym) hypernet_rows_units<DTW, DTS, DTO, M, HAS_SCALE, true, 8, true, false, UN>(a, 8, u0, kThreads);
    else hypernet_rows_units<DTW, DTS, DTO, M, HAS_SCALE, true, 8, false, false, UN>(a, 8, u0, kThreads);
  } else {
    if (asym) hypernet_rows_units<DTW, DTS, DTO, M, HAS_SCALE, true, 0, true, false, UN>(a, lpb, u0, kThreads);
    else hypernet_rows_units<DTW, DTS, DTO, M, HAS_SCALE, true, 0, false, false, UN>(a, lpb, u0, kThreads);
  }
}

template <int DTW, int DTS, int DTO>
static int launch_hn_multi(const HnMultiArgs& a, int64_t tiles, int M, bool has_scale, int units, hipStream_t s) {
  uint32_t e[10];
  for (int i = 0; i < 10; i++) e[i] = i + 1 < a.n ? (uint32_t)a.d[i + 1].tile0 : 0xFFFFFFFFu;
#define DMXQ_HM(M_, S_)                                                                                                                       \
  do {                                                                                                                                        \
    if (units == kHnUnitsSmall)                                                                                                               \
      DMXQ_LAUNCH((hypernet_rows_multi_kernel<DTW, DTS, DTO, M_, S_, kHnUnitsSmall>), dim3((unsigned)tiles), dim3(kThreads), 0, s, e[0], e[1], e[2], \
                  e[3], e[4], e[5], e[6], e[7], e[8], e[9], a);                                                                               \
    else                                                                                                                                      \
      DMXQ_LAUNCH((hypernet_rows_multi_kernel<DTW, DTS, DTO, M_, S_, kHnUnits>), dim3((unsigned)tiles), dim3(kThreads), 0, s, e[0], e[1], e[2], e[3], \
                  e[4], e[5], e[6], e[7], e[8], e[9], a);                                                                                     \
  } while (0)
  if (has_scale) { switch (M) { case 0: DMXQ_HM(0, true); break; case 2: DMXQ_HM(2, true); break; case 4: DMXQ_HM(4, true); break; default: DMXQ_HM(8, true); } }
  else { switch (M) { case 0: DMXQ_HM(0, false); break; case 2: DMXQ_HM(2, false); break; case 4: DMXQ_HM(4, false); break; default: DMXQ_HM(8, false); } }
#undef DMXQ_HM
  return launch_status();
}

}  // namespace dmxq

using namespace dmxq;

extern "C" int dmxq_weight_hypernet_multi(const dmxq_hypernet_desc* tensors, int64_t n_tensors, int dtype_w, int dtype_score, int K,
                                          int M, int dtype_out, int64_t block_size, int precision, int symmetric, void* stream) {
  if (n_tensors < 0 || (n_tensors > 0 && !tensors)) return DMXQ_ERR_BAD_ARG;
  if (!valid_dtype(dtype_w) || !valid_dtype(dtype_out) || block_size < 1) return DMXQ_ERR_BAD_ARG;
  if (M != 0 && (!valid_dtype(dtype_score) || K < 1 || K > M)) return DMXQ_ERR_BAD_ARG;
  const int64_t B = block_size;
  bool any = false, has_scale = false, first = true;
  int64_t total = 0;   // elements of the whole set (overflow-checked: rows * L and the running sum stay below 2^62)
  for (int64_t i = 0; i < n_tensors; i++) {
    const dmxq_hypernet_desc& t = tensors[i];
    if (t.rows < 0 || t.L < 0) return DMXQ_ERR_BAD_ARG;
    if (t.rows != 0 && t.L > (((int64_t)1 << 62) - total) / t.rows) return DMXQ_ERR_BAD_ARG;
    total += t.rows * t.L;
    if (t.rows * t.L == 0) continue;
    if (!t.w || !t.out || (M != 0 && !t.score)) return DMXQ_ERR_BAD_ARG;
    if (first) { has_scale = t.sq_scale != nullptr; first = false; }
    else if ((t.sq_scale != nullptr) != has_scale) return DMXQ_ERR_BAD_ARG;   // all with a SmoothQuant scale, or none
    any = true;
  }
  // the fusable geometry of dmxq_weight_hypernet, for every tensor; anything else is the caller's job (one call per tensor / unfused)
  if (!(M == 0 || M == 2 || M == 4 || M == 8) || (B & (B - 1)) != 0 || B < 8 || B > 512 || precision < 2 || precision > 20)
    return DMXQ_ERR_UNSUPPORTED;
  // units per lane by the size of the whole set (hypernet_rows.hpp): 2 up to 160 M elements, 4 beyond
  const int units = total <= ((int64_t)160 << 20) ? kHnUnitsSmall : kHnUnits;
  const int64_t TILE = (int64_t)kThreads * units;
  // ALL-OR-NOTHING (include/dmxq.h): every check that can return UNSUPPORTED runs here, before the first launch -- the per-tensor tile
  // count too (it sat in the launch loop below until round 5, behind a possible flush() of earlier batches: ADVICE r4)
  for (int64_t i = 0; i < n_tensors; i++) {
    const dmxq_hypernet_desc& t = tensors[i];
    if (t.rows * t.L == 0) continue;
    if (t.L % B != 0 || t.L % 8 != 0 || !aligned16(t.w) || !aligned16(t.out) || (M != 0 && !aligned16(t.score)) ||
        (t.sq_scale && !aligned16(t.sq_scale)))
      return DMXQ_ERR_UNSUPPORTED;
    if ((t.rows * t.L / 8 + TILE - 1) / TILE >= ((int64_t)1 << 31)) return DMXQ_ERR_UNSUPPORTED;
  }
  if (!any) return DMXQ_OK;
  hipStream_t s = (hipStream_t)stream;
  HnMultiArgs a;
  a.n = 0; a.K = K; a.lpb = (int)(B / 8); a.wl = precision; a.asym = symmetric ? 0 : 1;
  int64_t tiles = 0;
  int rc = DMXQ_OK;
  const int ds = M ? dtype_score : dtype_w;
  auto flush = [&]() {
    if (a.n == 0) return;
    int r = DMXQ_ERR_UNSUPPORTED;
#define DMXQ_DT(W_, S_, O_) \
  if (dtype_w == W_ && ds == S_ && dtype_out == O_) r = launch_hn_multi<W_, S_, O_>(a, tiles, M, has_scale, units, s);
    DMXQ_DT(DMXQ_BF16, DMXQ_F32, DMXQ_BF16)   // the dtype triples of dmxq_weight_hypernet
    DMXQ_DT(DMXQ_BF16, DMXQ_F32, DMXQ_F32)
    DMXQ_DT(DMXQ_BF16, DMXQ_BF16, DMXQ_BF16)
    DMXQ_DT(DMXQ_F16, DMXQ_F32, DMXQ_F16)
    DMXQ_DT(DMXQ_F16, DMXQ_F32, DMXQ_F32)
    DMXQ_DT(DMXQ_F16, DMXQ_F16, DMXQ_F16)
    DMXQ_DT(DMXQ_F32, DMXQ_F32, DMXQ_F32)
#undef DMXQ_DT
    if (r != DMXQ_OK && rc == DMXQ_OK) rc = r;
    a.n = 0; tiles = 0;
  };
  bool triple_ok = false;
  {
    static const int ok[7][3] = {{DMXQ_BF16, DMXQ_F32, DMXQ_BF16}, {DMXQ_BF16, DMXQ_F32, DMXQ_F32}, {DMXQ_BF16, DMXQ_BF16, DMXQ_BF16},
                                 {DMXQ_F16, DMXQ_F32, DMXQ_F16}, {DMXQ_F16, DMXQ_F32, DMXQ_F32}, {DMXQ_F16, DMXQ_F16, DMXQ_F16},
                                 {DMXQ_F32, DMXQ_F32, DMXQ_F32}};
    for (auto& t : ok) triple_ok |= (dtype_w == t[0] && ds == t[1] && dtype_out == t[2]);
  }
  if (!triple_ok) return DMXQ_ERR_UNSUPPORTED;   // before anything is launched: the caller falls back as a whole
  for (int64_t i = 0; i < n_tensors && rc == DMXQ_OK; i++) {
    const dmxq_hypernet_desc& t = tensors[i];
    const int64_t n = t.rows * t.L;
    if (n == 0) continue;
    const int64_t nt = (n / 8 + TILE - 1) / TILE;   // (< 2^31: checked above)
    if (a.n == kHnMultiMax || tiles + nt >= ((int64_t)1 << 31)) flush();
    a.d[a.n] = HnMultiDesc{t.w, M ? t.score : nullptr, t.sq_scale, t.out, n / 8, t.L, tiles, make_fastdiv31(t.L),
                           n < ((int64_t)1 << 31) ? 1 : 0};
    a.n++;
    tiles += nt;
  }
  flush();
  return rc;
}
