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

// ---------------------------------------------------------------------------------------------------------
// Vectorised path: inner == 1 (groups along the contiguous dim), M in {2,4,8,16}, 16-byte aligned streams.
// A lane owns U = 8 (M <= 8) or 16 consecutive elements = U/M whole groups, moved with 16-byte accesses
// (one per 16-bit stream, two or four per fp32 stream); UNROLL units per lane are loaded before the ranking.
template <int U>
__device__ __forceinline__ void load_elems(const void* p, int dt, int64_t e0, float (&v)[U]) {
  if (dt == DMXQ_F32) {
#pragma unroll
    for (int k = 0; k < U; k += 4) {
      const f32x4 t = *(const f32x4*)((const float*)p + e0 + k);
      v[k] = t.x; v[k + 1] = t.y; v[k + 2] = t.z; v[k + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int k = 0; k < U; k += 8) {
      const u32x4 t = *(const u32x4*)((const uint16_t*)p + e0 + k);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (dt == DMXQ_BF16) {
          v[k + 2 * j] = u2f(t[j] << 16);
          v[k + 2 * j + 1] = u2f(t[j] & 0xFFFF0000u);
        } else {
          v[k + 2 * j] = half_lo(t[j]);
          v[k + 2 * j + 1] = half_hi(t[j]);
        }
      }
    }
  }
}

template <int U>
__device__ __forceinline__ void store_elems(void* p, int dt, int64_t e0, const float (&v)[U]) {
  if (dt == DMXQ_F32) {
#pragma unroll
    for (int k = 0; k < U; k += 4) *(f32x4*)((float*)p + e0 + k) = f32x4{v[k], v[k + 1], v[k + 2], v[k + 3]};
  } else {
#pragma unroll
    for (int k = 0; k < U; k += 8) {
      u32x4 t;
#pragma unroll
      for (int j = 0; j < 4; j++)
        t[j] = dt == DMXQ_BF16 ? pack2<DMXQ_BF16>(v[k + 2 * j], v[k + 2 * j + 1]) : pack2<DMXQ_F16>(v[k + 2 * j], v[k + 2 * j + 1]);
      *(u32x4*)((uint16_t*)p + e0 + k) = t;
    }
  }
}

template <int M, int UNROLL>
__global__ __launch_bounds__(kThreads) void nm_mask_vec_kernel(NmArgs a, int64_t n_units) {
  constexpr int U = M <= 8 ? 8 : 16;
  const int64_t stride = (int64_t)gridDim.x * kThreads * UNROLL;
  for (int64_t u0 = ((int64_t)blockIdx.x * UNROLL) * kThreads + threadIdx.x; u0 < n_units; u0 += stride) {
    float s[UNROLL][U], x[UNROLL][U];
#pragma unroll
    for (int r = 0; r < UNROLL; r++) {
      const int64_t u = u0 + (int64_t)r * kThreads;
      if (u < n_units) {
        load_elems<U>(a.score, a.dts, u * U, s[r]);
        if (a.y) load_elems<U>(a.x, a.dtx, u * U, x[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < UNROLL; r++) {
      const int64_t u = u0 + (int64_t)r * kThreads;
      if (u >= n_units) continue;
      float mk[U];
#pragma unroll
      for (int g = 0; g < U; g += M) {
        int32_t key[M];
        int rank[M];
#pragma unroll
        for (int i = 0; i < M; i++) { key[i] = sort_key(s[r][g + i]); rank[i] = 0; }
#pragma unroll
        for (int i = 0; i < M; i++)
#pragma unroll
          for (int jj = 0; jj < i; jj++) {
            const bool jj_first = key[jj] <= key[i];  // equal keys: the lower index sorts first
            rank[i] += jj_first ? 1 : 0;
            rank[jj] += jj_first ? 0 : 1;
          }
#pragma unroll
        for (int i = 0; i < M; i++) mk[g + i] = rank[i] >= M - a.K ? 1.0f : 0.0f;
      }
      if (a.mask) store_elems<U>(a.mask, a.dtm, u * U, mk);
      if (a.y) {
        float y[U];
#pragma unroll
        for (int i = 0; i < U; i++) y[i] = x[r][i] * mk[i];
        store_elems<U>(a.y, a.dty, u * U, y);
      }
    }
  }
}

template <int M>
static void launch_nm_vec(const NmArgs& a, int64_t n, hipStream_t s) {
  constexpr int U = M <= 8 ? 8 : 16, UNROLL = 2;
  const int64_t n_units = n / U;
  DMXQ_LAUNCH((nm_mask_vec_kernel<M, UNROLL>), dim3(grid_for((n_units + UNROLL - 1) / UNROLL)), dim3(kThreads), 0, s,
                     a, n_units);
}

// float32 score -> float32 mask (BlockTopK.forward on a float32 weight's score: the standalone mask op), M in {2, 4, 8}.  In the
// kernel above a lane owns 8 consecutive elements = 32 bytes of each float32 stream, i.e. two 16-byte accesses 32 bytes apart: every
// load / store instruction of a wave touches half of each line it covers (59 % of roofline).  Here a lane owns ONE 16-byte vector per
// slot -- whole groups for M <= 4; for M = 8 neighbouring lanes swap vectors of two slots (DPP quad_perm, as csrc/bfp_pack.hip), so the
// even lane ranks the pair's group of slot 2k and the odd lane that of slot 2k + 1, and the mask bits travel back the same way --,
// workgroup-contiguous one-pass tiles, all loads first.  n_vec % 2 == 0 for M = 8 (whole groups).
template <int M, int UN>
__global__ __launch_bounds__(kThreads) void nm_mask_f32_kernel(const float* __restrict__ score, float* __restrict__ mask, int64_t n_vec, int K) {
  static_assert(M == 2 || M == 4 || M == 8, "groups of 2, 4 or 8");
  static_assert(UN % 2 == 0, "slots are taken in pairs");
  const int64_t base = (int64_t)blockIdx.x * (kThreads * UN) + threadIdx.x;
  u32x4 raw[UN];
#pragma unroll
  for (int u = 0; u < UN; u++) {
    const int64_t v = base + (int64_t)u * kThreads;
    // clamped: unconditional loads (M = 8: the same position of the last pair, so that the exchange below moves defined data)
    const int64_t vc = v < n_vec ? v : (M == 8 ? n_vec - 2 + (threadIdx.x & 1) : n_vec - 1);
    raw[u] = load_raw16<true>(score, vc * 16);
  }
  __builtin_amdgcn_sched_barrier(0);
  const int thr = M - K;
  auto rank_bits = [&](const int32_t (&key)[M]) __attribute__((always_inline)) -> uint32_t {
    int rank[M];
#pragma unroll
    for (int i = 0; i < M; i++) rank[i] = 0;
#pragma unroll
    for (int i = 0; i < M; i++)
#pragma unroll
      for (int jj = 0; jj < i; jj++) {
        const bool jj_first = key[jj] <= key[i];  // equal keys: the lower index sorts first
        rank[i] += jj_first ? 1 : 0;
        rank[jj] += jj_first ? 0 : 1;
      }
    uint32_t bits = 0u;
#pragma unroll
    for (int i = 0; i < M; i++) bits |= rank[i] >= thr ? (1u << i) : 0u;
    return bits;
  };
  auto store_bits = [&](int64_t v, uint32_t b4) __attribute__((always_inline)) {
    if (v < n_vec) {
      const uint32_t one = 0x3F800000u;
      __builtin_nontemporal_store(u32x4{(b4 & 1u) ? one : 0u, (b4 & 2u) ? one : 0u, (b4 & 4u) ? one : 0u, (b4 & 8u) ? one : 0u}, (u32x4*)(mask + v * 4));
    }
  };
  if constexpr (M <= 4) {
#pragma unroll
    for (int u = 0; u < UN; u++) {
      uint32_t b4 = 0u;
#pragma unroll
      for (int g = 0; g < 4; g += M) {
        int32_t key[M];
#pragma unroll
        for (int i = 0; i < M; i++) key[i] = sort_key(u2f(raw[u][g + i]));
        b4 |= rank_bits(key) << g;
      }
      store_bits(base + (int64_t)u * kThreads, b4);
    }
  } else {
    const bool odd = (threadIdx.x & 1) != 0;
#pragma unroll
    for (int u = 0; u < UN; u += 2) {
      // the even lane gives away its slot u + 1 vector, the odd lane its slot u vector
      u32x4 r;
#pragma unroll
      for (int j = 0; j < 4; j++) r[j] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(odd ? raw[u][j] : raw[u + 1][j]), 0xB1, 0xF, 0xF, false);
      int32_t key[8];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        key[j] = sort_key(u2f(odd ? r[j] : raw[u][j]));              // first half of the group: the even lane's vector
        key[4 + j] = sort_key(u2f(odd ? raw[u + 1][j] : r[j]));      // second half: the odd lane's
      }
      const uint32_t b8 = rank_bits(key);
      // even lane: bits 0-3 are its own slot-u vector, bits 4-7 the odd lane's slot-u vector; odd lane: bits 0-3 the even lane's slot u + 1
      // vector, bits 4-7 its own
      const uint32_t back = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(odd ? (b8 & 0xFu) : (b8 >> 4)), 0xB1, 0xF, 0xF, false);
      store_bits(base + (int64_t)u * kThreads, odd ? back : (b8 & 0xFu));
      store_bits(base + (int64_t)(u + 1) * kThreads, odd ? (b8 >> 4) : back);
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

// hypernet.hip: compile-time typed mask-and-multiply (no mask output); DMXQ_ERR_UNSUPPORTED = not applicable
extern "C" int dmxq_internal_nm_sparsify_typed(const void* score, int dtype_score, const void* x, int dtype_x, void* y, int dtype_y,
                                               int64_t n, int K, int M, void* stream);

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
  const bool vec_ok = inner == 1 && (M == 2 || M == 4 || M == 8 || M == 16) && n % 16 == 0 && aligned16(score) &&
                      (!x || aligned16(x)) && (!mask_out || aligned16(mask_out)) && (!y_out || aligned16(y_out));
  if (vec_ok && y_out && !mask_out) {
    const int rc = dmxq_internal_nm_sparsify_typed(score, dtype_score, x, dtype_x, y_out, dtype_y, n, K, M, stream);
    if (rc != DMXQ_ERR_UNSUPPORTED) return rc;
  }
  if (vec_ok && mask_out && !y_out && dtype_score == DMXQ_F32 && dtype_mask == DMXQ_F32 && M <= 8) {
    constexpr int UN = 4;
    const int64_t n_vec = n / 4;
    const int64_t tiles = (n_vec + kThreads * UN - 1) / (kThreads * UN);
    if (tiles <= 0x7FFFFFFF) {
      switch (M) {
        case 2: DMXQ_LAUNCH((nm_mask_f32_kernel<2, UN>), dim3((unsigned)tiles), dim3(kThreads), 0, s, (const float*)score, (float*)mask_out, n_vec, K); break;
        case 4: DMXQ_LAUNCH((nm_mask_f32_kernel<4, UN>), dim3((unsigned)tiles), dim3(kThreads), 0, s, (const float*)score, (float*)mask_out, n_vec, K); break;
        default: DMXQ_LAUNCH((nm_mask_f32_kernel<8, UN>), dim3((unsigned)tiles), dim3(kThreads), 0, s, (const float*)score, (float*)mask_out, n_vec, K); break;
      }
      return launch_status();
    }
  }
  if (vec_ok) {
    switch (M) {
      case 2: launch_nm_vec<2>(a, n, s); break;
      case 4: launch_nm_vec<4>(a, n, s); break;
      case 8: launch_nm_vec<8>(a, n, s); break;
      default: launch_nm_vec<16>(a, n, s); break;
    }
    return launch_status();
  }
  const int grid = grid_for(n / M);
  switch (M) {
    case 2: DMXQ_LAUNCH(nm_mask_kernel<2>, dim3(grid), dim3(kThreads), 0, s, a); break;
    case 4: DMXQ_LAUNCH(nm_mask_kernel<4>, dim3(grid), dim3(kThreads), 0, s, a); break;
    case 8: DMXQ_LAUNCH(nm_mask_kernel<8>, dim3(grid), dim3(kThreads), 0, s, a); break;
    case 16: DMXQ_LAUNCH(nm_mask_kernel<16>, dim3(grid), dim3(kThreads), 0, s, a); break;
    default: DMXQ_LAUNCH(nm_mask_anyM_kernel, dim3(grid), dim3(kThreads), 0, s, a, M); break;
  }
  return launch_status();
}
