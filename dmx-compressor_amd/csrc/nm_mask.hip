// csrc/nm_mask.hip — N:M structured-sparsity mask (+ fused apply) for gfx950.
//
// Replaces sparse.py:163-180 BlockTopK.forward (transpose -> reshape(-1, M) -> argsort -> ones.scatter_ ->
// reshape -> transpose) and the `x * mask` of Sparsify.forward (sparse.py:300): the reference materialises
// int64 sort indices (8 B/elem) plus a ones tensor; here each lane owns one M-group, ranks its M scores in
// registers (M(M-1)/2 comparisons, no sort, no indices) and writes mask and/or x*mask directly.
//
// Rank rule (pinned by tests/golden/nm_mask_*.npz against the reference): ascending STABLE order, NaN last:
//   before(j, i) = s_j < s_i  or  (s_j == s_i or both NaN) and j < i   [NaN is "greater" than any number]
//   rank_i = #{j : before(j, i)} ;  mask_i = (rank_i >= M - K) ? 1 : 0
// y = x * mask is a real multiply (a masked negative gives -0.0, NaN*0 = NaN), as in the reference.
#include "common.hpp"

namespace dmxq {

// key that orders like the reference's sort: monotone map of the float to a signed-comparable integer,
// -0.0 == +0.0, every NaN maps to the same top key.
__device__ __forceinline__ int32_t sort_key(float s) {
  if (s != s) return 0x7FFFFFFF;
  if (s == 0.0f) return 0;
  const int32_t b = (int32_t)f2u(s);
  return b >= 0 ? b : (int32_t)(0x80000000u - (uint32_t)b);
}

struct NmArgs {
  const void* score; const void* x; void* mask; void* y;
  int dts, dtx, dtm, dty;
  int64_t outer, L, inner;
  int K;
};

template <int M>
__global__ __launch_bounds__(kThreads) void nm_mask_kernel(NmArgs a) {
  const int64_t ngrp = a.L / M;
  const int64_t total = a.outer * ngrp * a.inner;
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x; t < total; t += stride) {
    const int64_t j = t % a.inner;
    const int64_t g = (t / a.inner) % ngrp;
    const int64_t o = t / (a.inner * ngrp);
    const int64_t e0 = (o * a.L + g * M) * a.inner + j;
    int32_t key[M];
#pragma unroll
    for (int i = 0; i < M; i++) key[i] = sort_key(load_rt(a.score, a.dts, e0 + i * a.inner));
    int rank[M];
#pragma unroll
    for (int i = 0; i < M; i++) rank[i] = 0;
#pragma unroll
    for (int i = 0; i < M; i++)
#pragma unroll
      for (int jj = 0; jj < i; jj++) {
        // jj < i: on equal keys the lower index jj sorts first
        const bool jj_first = key[jj] <= key[i];
        rank[i] += jj_first ? 1 : 0;
        rank[jj] += jj_first ? 0 : 1;
      }
#pragma unroll
    for (int i = 0; i < M; i++) {
      const float mk = rank[i] >= M - a.K ? 1.0f : 0.0f;
      const int64_t e = e0 + i * a.inner;
      if (a.mask) store_rt(a.mask, a.dtm, e, mk);
      if (a.y) store_rt(a.y, a.dty, e, load_rt(a.x, a.dtx, e) * mk);
    }
  }
}

// any M <= 64: same rule with runtime loops (scores re-read instead of kept in registers)
__global__ __launch_bounds__(kThreads) void nm_mask_anyM_kernel(NmArgs a, int M) {
  const int64_t ngrp = a.L / M;
  const int64_t total = a.outer * ngrp * a.inner;
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x; t < total; t += stride) {
    const int64_t j = t % a.inner;
    const int64_t g = (t / a.inner) % ngrp;
    const int64_t o = t / (a.inner * ngrp);
    const int64_t e0 = (o * a.L + g * M) * a.inner + j;
    for (int i = 0; i < M; i++) {
      const int32_t ki = sort_key(load_rt(a.score, a.dts, e0 + i * a.inner));
      int rank = 0;
      for (int jj = 0; jj < M; jj++) {
        const int32_t kj = sort_key(load_rt(a.score, a.dts, e0 + jj * a.inner));
        rank += (kj < ki || (kj == ki && jj < i)) ? 1 : 0;
      }
      const float mk = rank >= M - a.K ? 1.0f : 0.0f;
      const int64_t e = e0 + i * a.inner;
      if (a.mask) store_rt(a.mask, a.dtm, e, mk);
      if (a.y) store_rt(a.y, a.dty, e, load_rt(a.x, a.dtx, e) * mk);
    }
  }
}

}  // namespace dmxq

using namespace dmxq;

extern "C" int dmxq_nm_mask(const void* score, int dtype_score, const void* x, int dtype_x, void* mask_out,
                            int dtype_mask, void* y_out, int dtype_y, int64_t outer, int64_t L, int64_t inner, int K,
                            int M, void* stream) {
  if (!valid_dtype(dtype_score) || outer < 0 || L < 0 || inner < 0) return DMXQ_ERR_BAD_ARG;
  if (M < 1 || M > 64 || K < 1 || K > M || L % M != 0) return DMXQ_ERR_BAD_ARG;  // sparse.py:158,166-168
  if (mask_out && !valid_dtype(dtype_mask)) return DMXQ_ERR_BAD_ARG;
  if (y_out && (!valid_dtype(dtype_y) || !valid_dtype(dtype_x) || !x)) return DMXQ_ERR_BAD_ARG;
  if (!mask_out && !y_out) return DMXQ_ERR_BAD_ARG;
  const int64_t n = outer * L * inner;
  if (n == 0) return DMXQ_OK;
  if (!score) return DMXQ_ERR_BAD_ARG;
  NmArgs a{score, x, mask_out, y_out, dtype_score, dtype_x, dtype_mask, dtype_y, outer, L, inner, K};
  hipStream_t s = (hipStream_t)stream;
  const int grid = grid_for(n / M);
  switch (M) {
    case 2: hipLaunchKernelGGL(nm_mask_kernel<2>, dim3(grid), dim3(kThreads), 0, s, a); break;
    case 4: hipLaunchKernelGGL(nm_mask_kernel<4>, dim3(grid), dim3(kThreads), 0, s, a); break;
    case 8: hipLaunchKernelGGL(nm_mask_kernel<8>, dim3(grid), dim3(kThreads), 0, s, a); break;
    case 16: hipLaunchKernelGGL(nm_mask_kernel<16>, dim3(grid), dim3(kThreads), 0, s, a); break;
    default: hipLaunchKernelGGL(nm_mask_anyM_kernel, dim3(grid), dim3(kThreads), 0, s, a, M); break;
  }
  return launch_status();
}
