// csrc/topk.hip — unstructured top-k sparsity mask ("TOPK{density}", sparse.py:109-123 TopK.forward) and its
// application (`x * mask`, sparse.py:300) for gfx950.
//
// The reference argsorts the whole flattened score tensor (int64 indices, 8 B/elem) to zero its n_zero smallest
// entries.  Here: a 3-level most-significant-digit radix SELECT (digits of 11, 11, 10 bits) finds the key T of the n_zero-th smallest score
// (one 2048-bin histogram pass per level over an order-preserving 32-bit key; the bucket choice between levels is a
// one-wave kernel, so there is no host round trip), which also yields how many scores are below T and how many equal
// it; a final pass writes mask / x*mask.  Ties at T: exactly the first r_eq of them in index order are zeroed (stable
// ascending order, the rule of nm_mask.hip: -0 == +0, NaN largest) -- the reference's unstable sort leaves that
// choice undefined.  Only when r_eq is fewer than all ties does an index-ordered count run (per-chunk tie counts,
// a scan over chunks, per-thread prefixes inside a chunk); otherwise those two launches return immediately.
// Passes over the scores: 3 + 1 (+ 1 with boundary ties), one write.
#include "common.hpp"

namespace dmxq {

constexpr int kTopkChunk = kThreads * 8;  // elements per chunk: 8 consecutive per thread
constexpr int kTopkLevels = 3, kTopkBins = 2048;  // digits of 11, 11 and 10 bits, most significant first
__host__ __device__ constexpr int topk_shift(int level) { return level == 0 ? 21 : (level == 1 ? 10 : 0); }
__host__ __device__ constexpr int topk_bits(int level) { return level == 2 ? 10 : 11; }

struct TopkState {
  uint32_t prefix;      // key bits fixed so far (high bits)
  uint32_t pad;
  int64_t rank;         // 0-based rank of the wanted key among the keys matching `prefix`
  int64_t less;         // number of keys known to be below the wanted one
  int64_t count_eq;     // after the last level: number of keys equal to T
};
// workspace layout: TopkState | uint32 hist[2048] | int64 chunk_counts[ceil(n / kTopkChunk)]
struct TopkWs {
  TopkState* st;
  uint32_t* hist;
  int64_t* chunks;
};
__host__ __device__ inline TopkWs topk_ws(void* base) {
  char* p = (char*)base;
  return TopkWs{(TopkState*)p, (uint32_t*)(p + 64), (int64_t*)(p + 64 + 4 * kTopkBins)};
}

// unsigned order-preserving key: -0 == +0, every NaN the same, largest key
__device__ __forceinline__ uint32_t ukey(float s) {
  if (s != s) return 0xFFFFFFFFu;
  if (s == 0.0f) return 0x80000000u;
  const uint32_t b = f2u(s);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// 8 consecutive scores as keys; compile-time dtype, 16-byte loads when the base allows it (fp32: two), scalar otherwise.
// e + 8 <= n required.
template <int DT>
__device__ __forceinline__ void load8_keys(const void* p, int64_t e, bool vec, uint32_t (&key)[8]) {
  float v[8];
  if (vec) {
    if (DT == DMXQ_F32) {
      const f32x4 a = *(const f32x4*)((const float*)p + e), b = *(const f32x4*)((const float*)p + e + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
      const u32x4 t = *(const u32x4*)((const uint16_t*)p + e);
      widen<DT, 8>(t, v);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = load1<DT>(p, e + k);
  }
#pragma unroll
  for (int k = 0; k < 8; k++) key[k] = ukey(v[k]);
}

__device__ __forceinline__ bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
// runtime-dtype 8-wide access for x / mask / y (16-byte accesses when `vec`)
__device__ __forceinline__ void load8_any(const void* p, int dt, int64_t e, bool vec, float (&v)[8]) {
  if (vec && dt == DMXQ_F32) {
    const f32x4 a = *(const f32x4*)((const float*)p + e), b = *(const f32x4*)((const float*)p + e + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else if (vec && dt == DMXQ_BF16) {
    widen<DMXQ_BF16, 8>(*(const u32x4*)((const uint16_t*)p + e), v);
  } else if (vec) {
    widen<DMXQ_F16, 8>(*(const u32x4*)((const uint16_t*)p + e), v);
  } else {
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = load_rt(p, dt, e + k);
  }
}
__device__ __forceinline__ void store8_any(void* p, int dt, int64_t e, bool vec, const float (&v)[8]) {
  if (vec && dt == DMXQ_F32) store_vec<DMXQ_F32, 8>(p, e, v);
  else if (vec && dt == DMXQ_BF16) store_vec<DMXQ_BF16, 8>(p, e, v);
  else if (vec) store_vec<DMXQ_F16, 8>(p, e, v);
  else {
#pragma unroll
    for (int k = 0; k < 8; k++) store_rt(p, dt, e + k, v[k]);
  }
}

__global__ void topk_init_kernel(TopkWs ws, int64_t n_zero) {
  if (threadIdx.x == 0) *ws.st = TopkState{0u, 0u, n_zero - 1, 0, 0};
  for (int i = threadIdx.x; i < kTopkBins; i += blockDim.x) ws.hist[i] = 0u;
}

// level 0..2: histogram of this level's digit over the keys whose higher digits equal st->prefix.  The scores of real
// tensors crowd into a few exponent values (11-bit digits split them by the top mantissa bits), and the workgroup
// additionally keeps kHistCopies copies of the histogram in LDS, lane -> copy, to spread same-address atomics; they are
// summed when the workgroup flushes.  (A 13-bit first digit in 32 KiB of LDS measured slower: 148 vs 120 us.)
constexpr int kHistCopies = 4;
template <int DT>
__global__ __launch_bounds__(kThreads) void topk_hist_kernel(const void* __restrict__ score, int64_t n, int level, int vec,
                                                            TopkWs ws) {
  __shared__ uint32_t h[kHistCopies][kTopkBins];
  for (int i = threadIdx.x; i < kHistCopies * kTopkBins; i += kThreads) (&h[0][0])[i] = 0u;
  __syncthreads();
  const uint32_t prefix = ws.st->prefix;
  const int shift = topk_shift(level), above = shift + topk_bits(level);  // bits above this digit are fixed by `prefix`
  const uint32_t dmask = (1u << topk_bits(level)) - 1u;
  uint32_t* mine = h[threadIdx.x & (kHistCopies - 1)];
  const int64_t n8 = n / 8;
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t u = (int64_t)blockIdx.x * kThreads + threadIdx.x; u < n8; u += stride) {
    uint32_t key[8];
    load8_keys<DT>(score, u * 8, vec != 0, key);
#pragma unroll
    for (int k = 0; k < 8; k++)
      if (level == 0 || (key[k] >> above) == prefix) atomicAdd(&mine[(key[k] >> shift) & dmask], 1u);
  }
  if (blockIdx.x == 0 && threadIdx.x < (unsigned)(n - n8 * 8)) {  // tail
    const uint32_t k = ukey(load1<DT>(score, n8 * 8 + threadIdx.x));
    if (level == 0 || (k >> above) == prefix) atomicAdd(&mine[(k >> shift) & dmask], 1u);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < kTopkBins; b += kThreads) {
    uint32_t sum = 0u;
#pragma unroll
    for (int c = 0; c < kHistCopies; c++) sum += h[c][b];
    if (sum) atomicAdd(&ws.hist[b], sum);
  }
}

// one workgroup of 256 threads, 8 bins each: pick the bin that holds `rank`, descend
__global__ void topk_select_kernel(TopkWs ws, int level) {
  constexpr int PER = kTopkBins / 256;
  __shared__ int64_t tsum[256];
  const int t = threadIdx.x;
  uint32_t c[PER];
  int64_t mine = 0;
#pragma unroll
  for (int k = 0; k < PER; k++) { c[k] = ws.hist[t * PER + k]; mine += c[k]; }
  tsum[t] = mine;
  __syncthreads();
  if (t == 0) {
    int64_t run = 0;
    for (int b = 0; b < 256; b++) { const int64_t v = tsum[b]; tsum[b] = run; run += v; }  // exclusive prefix over threads
  }
  __syncthreads();
  const int64_t rank = ws.st->rank;
  int64_t below = tsum[t];
  __syncthreads();
  if (mine > 0 && rank >= below && rank < below + mine) {
#pragma unroll
    for (int k = 0; k < PER; k++) {
      if (rank >= below && rank < below + c[k]) {
        ws.st->prefix = (ws.st->prefix << topk_bits(level)) | (uint32_t)(t * PER + k);
        ws.st->rank = rank - below;
        ws.st->less += below;
        if (level == kTopkLevels - 1) ws.st->count_eq = c[k];
      }
      below += c[k];
    }
  }
#pragma unroll
  for (int k = 0; k < PER; k++) ws.hist[t * PER + k] = 0u;
}

// number of keys equal to T in every chunk (skipped when all ties are zeroed anyway)
template <int DT>
__global__ __launch_bounds__(kThreads) void topk_tie_count_kernel(const void* __restrict__ score, int64_t n, int vec,
                                                                 int64_t n_zero, TopkWs ws) {
  const TopkState st = *ws.st;
  if (n_zero - st.less >= st.count_eq) return;
  const uint32_t T = st.prefix;
  __shared__ int part[kThreads / kWave];
  const int64_t n_chunks = (n + kTopkChunk - 1) / kTopkChunk;
  for (int64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const int64_t e0 = c * kTopkChunk + (int64_t)threadIdx.x * 8;
    int cnt = 0;
    if (e0 + 8 <= n) {
      uint32_t key[8];
      load8_keys<DT>(score, e0, vec != 0, key);
#pragma unroll
      for (int k = 0; k < 8; k++) cnt += key[k] == T ? 1 : 0;
    } else {
#pragma unroll
      for (int k = 0; k < 8; k++)
        if (e0 + k < n) cnt += ukey(load1<DT>(score, e0 + k)) == T ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    __syncthreads();
    if ((threadIdx.x & (kWave - 1)) == 0) part[threadIdx.x / kWave] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
      int s = 0;
      for (int i = 0; i < kThreads / kWave; i++) s += part[i];
      ws.chunks[c] = s;
    }
  }
}

// exclusive scan of the chunk counts, in place (one workgroup; chunks <= n / 2048)
__global__ __launch_bounds__(kThreads) void topk_scan_kernel(int64_t n_chunks, int64_t n_zero, TopkWs ws) {
  const TopkState st = *ws.st;
  if (n_zero - st.less >= st.count_eq) return;
  __shared__ int64_t sh[kThreads];
  __shared__ int64_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < n_chunks; base += kThreads) {
    const int64_t i = base + threadIdx.x;
    const int64_t v = i < n_chunks ? ws.chunks[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < kThreads; o <<= 1) {  // Hillis-Steele inclusive scan
      const int64_t add = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
      __syncthreads();
      sh[threadIdx.x] += add;
      __syncthreads();
    }
    const int64_t incl = sh[threadIdx.x], c0 = carry;
    if (i < n_chunks) ws.chunks[i] = c0 + incl - v;
    __syncthreads();
    if (threadIdx.x == kThreads - 1) carry = c0 + incl;
    __syncthreads();
  }
}

template <int DT>
__global__ __launch_bounds__(kThreads) void topk_apply_kernel(const void* __restrict__ score, int vec,
                                                             const void* __restrict__ x, int dtx, void* __restrict__ mask,
                                                             int dtm, void* __restrict__ y, int dty, int64_t n,
                                                             int64_t n_zero, TopkWs ws) {
  const TopkState st = *ws.st;
  const uint32_t T = st.prefix;
  const int64_t r_eq = n_zero - st.less;          // ties to zero (>= 1 when n_zero >= 1)
  const bool scan = r_eq < st.count_eq;           // only some of the ties go: index order decides
  __shared__ int part[kThreads / kWave];
  const int64_t n_chunks = (n + kTopkChunk - 1) / kTopkChunk;
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
  for (int64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const int64_t e0 = c * kTopkChunk + (int64_t)threadIdx.x * 8;
    uint32_t key[8];
    int cnt = 0;
    if (e0 + 8 <= n) {
      load8_keys<DT>(score, e0, vec != 0, key);
    } else {
#pragma unroll
      for (int k = 0; k < 8; k++) key[k] = e0 + k < n ? ukey(load1<DT>(score, e0 + k)) : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) cnt += (e0 + k < n && key[k] == T) ? 1 : 0;
    int64_t before = 0;  // ties with a lower index than this thread's first element
    if (scan) {
      int incl = cnt;    // inclusive scan over the lanes of the wave, then over the waves
#pragma unroll
      for (int o = 1; o < kWave; o <<= 1) { const int up = __shfl_up(incl, o); incl += lane >= o ? up : 0; }
      __syncthreads();
      if (lane == kWave - 1) part[wv] = incl;
      __syncthreads();
      int wbase = 0;
      for (int i = 0; i < wv; i++) wbase += part[i];
      before = ws.chunks[c] + wbase + (incl - cnt);
    }
    float mk[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      bool keep = key[k] > T;
      if (key[k] == T) { keep = scan ? before >= r_eq : false; before += 1; }
      if (n_zero <= 0) keep = true;
      mk[k] = keep ? 1.0f : 0.0f;
    }
    if (e0 + 8 <= n) {  // whole 8-element unit: 16-byte accesses where the bases allow them
      if (mask) store8_any(mask, dtm, e0, al16(mask), mk);
      if (y) {
        float xv[8], yv[8];
        load8_any(x, dtx, e0, al16(x), xv);
#pragma unroll
        for (int k = 0; k < 8; k++) yv[k] = xv[k] * mk[k];
        store8_any(y, dty, e0, al16(y), yv);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int64_t e = e0 + k;
        if (e < n) {
          if (mask) store_rt(mask, dtm, e, mk[k]);
          if (y) store_rt(y, dty, e, load_rt(x, dtx, e) * mk[k]);
        }
      }
    }
  }
}

}  // namespace dmxq

using namespace dmxq;

extern "C" int64_t dmxq_topk_workspace_bytes(int64_t n) {
  const int64_t chunks = n > 0 ? (n + kTopkChunk - 1) / kTopkChunk : 0;
  return 64 + 4 * kTopkBins + 8 * chunks;
}

extern "C" int dmxq_topk_mask(const void* score, int dtype_score, const void* x, int dtype_x, void* mask_out,
                              int dtype_mask, void* y_out, int dtype_y, int64_t n, int64_t n_zero, void* workspace,
                              void* stream) {
  if (!valid_dtype(dtype_score) || n < 0 || n_zero < 0 || n_zero > n) return DMXQ_ERR_BAD_ARG;
  if (mask_out && !valid_dtype(dtype_mask)) return DMXQ_ERR_BAD_ARG;
  if (y_out && (!x || !valid_dtype(dtype_x) || !valid_dtype(dtype_y))) return DMXQ_ERR_BAD_ARG;
  if (n == 0) return DMXQ_OK;
  if (n >= ((int64_t)1 << 32)) return DMXQ_ERR_UNSUPPORTED;  // 32-bit histogram counters
  if (!score || !workspace || (reinterpret_cast<uintptr_t>(workspace) & 7u)) return DMXQ_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  const TopkWs ws = topk_ws(workspace);
  const int64_t n_chunks = (n + kTopkChunk - 1) / kTopkChunk;
  const int cgrid = (int)(n_chunks < kMaxBlocks ? n_chunks : kMaxBlocks);
  const int vec = aligned16(score) ? 1 : 0;
  const int hgrid = grid_for((n + 7) / 8);
  DMXQ_LAUNCH(topk_init_kernel, dim3(1), dim3(256), 0, s, ws, n_zero > 0 ? n_zero : 1);
#define DMXQ_TOPK(D_)                                                                                                  \
  do {                                                                                                                 \
    if (n_zero > 0) {                                                                                                  \
      for (int level = 0; level < kTopkLevels; level++) {                                                              \
        DMXQ_LAUNCH(topk_hist_kernel<D_>, dim3(hgrid), dim3(kThreads), 0, s, score, n, level, vec, ws);         \
        DMXQ_LAUNCH(topk_select_kernel, dim3(1), dim3(256), 0, s, ws, level);                                   \
      }                                                                                                                \
      DMXQ_LAUNCH(topk_tie_count_kernel<D_>, dim3(cgrid), dim3(kThreads), 0, s, score, n, vec, n_zero, ws);     \
      DMXQ_LAUNCH(topk_scan_kernel, dim3(1), dim3(kThreads), 0, s, n_chunks, n_zero, ws);                       \
    }                                                                                                                  \
    DMXQ_LAUNCH(topk_apply_kernel<D_>, dim3(cgrid), dim3(kThreads), 0, s, score, vec, x, dtype_x, mask_out,     \
                       dtype_mask, y_out, dtype_y, n, n_zero, ws);                                                     \
  } while (0)
  if (dtype_score == DMXQ_F32) DMXQ_TOPK(DMXQ_F32);
  else if (dtype_score == DMXQ_F16) DMXQ_TOPK(DMXQ_F16);
  else DMXQ_TOPK(DMXQ_BF16);
#undef DMXQ_TOPK
  return launch_status();
}
