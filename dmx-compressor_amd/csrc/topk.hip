// csrc/topk.hip — unstructured top-k sparsity mask ("TOPK{density}", sparse.py:109-123 TopK.forward) and its
// application (`x * mask`, sparse.py:300) for gfx950.
//
// The reference argsorts the whole flattened score tensor (int64 indices, 8 B/elem) to zero its n_zero smallest
// entries.  Here: a 4-level most-significant-digit radix SELECT finds the key T of the n_zero-th smallest score
// (one 256-bin histogram pass per level over an order-preserving 32-bit key; the bucket choice between levels is a
// one-wave kernel, so there is no host round trip), which also yields how many scores are below T and how many equal
// it; a final pass writes mask / x*mask.  Ties at T: exactly the first r_eq of them in index order are zeroed (stable
// ascending order, the rule of nm_mask.hip: -0 == +0, NaN largest) -- the reference's unstable sort leaves that
// choice undefined.  Only when r_eq is fewer than all ties does an index-ordered count run (per-chunk tie counts,
// a scan over chunks, per-thread prefixes inside a chunk); otherwise those two launches return immediately.
// Passes over the scores: 4 + 1 (+ 1 with boundary ties), one write.
#include "common.hpp"

namespace dmxq {

constexpr int kTopkChunk = kThreads * 8;  // elements per chunk: 8 consecutive per thread

struct TopkState {
  uint32_t prefix;      // key bits fixed so far (high bits)
  uint32_t pad;
  int64_t rank;         // 0-based rank of the wanted key among the keys matching `prefix`
  int64_t less;         // number of keys known to be below the wanted one
  int64_t count_eq;     // after the last level: number of keys equal to T
};
// workspace layout: TopkState | uint32 hist[256] | int64 chunk_counts[ceil(n / kTopkChunk)]
struct TopkWs {
  TopkState* st;
  uint32_t* hist;
  int64_t* chunks;
};
__host__ __device__ inline TopkWs topk_ws(void* base) {
  char* p = (char*)base;
  return TopkWs{(TopkState*)p, (uint32_t*)(p + 64), (int64_t*)(p + 64 + 1024)};
}

// unsigned order-preserving key: -0 == +0, every NaN the same, largest key
__device__ __forceinline__ uint32_t ukey(float s) {
  if (s != s) return 0xFFFFFFFFu;
  if (s == 0.0f) return 0x80000000u;
  const uint32_t b = f2u(s);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__global__ void topk_init_kernel(TopkWs ws, int64_t n_zero) {
  if (threadIdx.x == 0) *ws.st = TopkState{0u, 0u, n_zero - 1, 0, 0};
  ws.hist[threadIdx.x] = 0u;
}

// level 0..3: histogram of byte (3 - level) of the keys whose higher bytes equal st->prefix
__global__ __launch_bounds__(kThreads) void topk_hist_kernel(const void* __restrict__ score, int dt, int64_t n, int level,
                                                            TopkWs ws) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0u;
  __syncthreads();
  const uint32_t prefix = ws.st->prefix;
  const int shift = 24 - 8 * level;
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t e0 = (int64_t)blockIdx.x * kThreads + threadIdx.x; e0 < n; e0 += 4 * stride) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) v[u] = load_rt(score, dt, e0 + u * stride < n ? e0 + u * stride : e0);  // 4 loads in flight
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint32_t k = ukey(v[u]);
      if (e0 + u * stride < n && (level == 0 || (k >> (shift + 8)) == prefix)) atomicAdd(&h[(k >> shift) & 255u], 1u);
    }
  }
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&ws.hist[threadIdx.x], h[threadIdx.x]);
}

// one workgroup of 256: pick the bucket that holds rank, descend
__global__ void topk_select_kernel(TopkWs ws, int level) {
  __shared__ int64_t cum[256];
  const int t = threadIdx.x;
  cum[t] = ws.hist[t];
  __syncthreads();
  if (t == 0) {
    int64_t run = 0;
    for (int b = 0; b < 256; b++) { const int64_t c = cum[b]; cum[b] = run; run += c; }  // exclusive prefix
  }
  __syncthreads();
  const int64_t rank = ws.st->rank;
  const int64_t mine = cum[t], cnt = ws.hist[t];
  __syncthreads();
  if (cnt > 0 && rank >= mine && rank < mine + cnt) {
    ws.st->prefix = (ws.st->prefix << 8) | (uint32_t)t;
    ws.st->rank = rank - mine;
    ws.st->less += mine;
    if (level == 3) ws.st->count_eq = cnt;
  }
  ws.hist[t] = 0u;
}

// number of keys equal to T in every chunk (skipped when all ties are zeroed anyway)
__global__ __launch_bounds__(kThreads) void topk_tie_count_kernel(const void* __restrict__ score, int dt, int64_t n,
                                                                 int64_t n_zero, TopkWs ws) {
  const TopkState st = *ws.st;
  if (n_zero - st.less >= st.count_eq) return;
  const uint32_t T = st.prefix;
  __shared__ int part[kThreads / kWave];
  const int64_t n_chunks = (n + kTopkChunk - 1) / kTopkChunk;
  for (int64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const int64_t e0 = c * kTopkChunk + (int64_t)threadIdx.x * 8;
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < 8; k++)
      if (e0 + k < n) cnt += ukey(load_rt(score, dt, e0 + k)) == T ? 1 : 0;
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

__global__ __launch_bounds__(kThreads) void topk_apply_kernel(const void* __restrict__ score, int dts,
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
#pragma unroll
    for (int k = 0; k < 8; k++) {
      key[k] = e0 + k < n ? ukey(load_rt(score, dts, e0 + k)) : 0xFFFFFFFFu;
      cnt += (e0 + k < n && key[k] == T) ? 1 : 0;
    }
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
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int64_t e = e0 + k;
      if (e < n) {
        bool keep = key[k] > T;
        if (key[k] == T) { keep = scan ? before >= r_eq : false; before += 1; }
        if (n_zero <= 0) keep = true;
        const float mk = keep ? 1.0f : 0.0f;
        if (mask) store_rt(mask, dtm, e, mk);
        if (y) store_rt(y, dty, e, load_rt(x, dtx, e) * mk);
      }
    }
  }
}

}  // namespace dmxq

using namespace dmxq;

extern "C" int64_t dmxq_topk_workspace_bytes(int64_t n) {
  const int64_t chunks = n > 0 ? (n + kTopkChunk - 1) / kTopkChunk : 0;
  return 64 + 1024 + 8 * chunks;
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
  const int grid = grid_for(n), cgrid = (int)(n_chunks < kMaxBlocks ? n_chunks : kMaxBlocks);
  hipLaunchKernelGGL(topk_init_kernel, dim3(1), dim3(256), 0, s, ws, n_zero > 0 ? n_zero : 1);
  if (n_zero > 0) {
    for (int level = 0; level < 4; level++) {
      hipLaunchKernelGGL(topk_hist_kernel, dim3(grid), dim3(kThreads), 0, s, score, dtype_score, n, level, ws);
      hipLaunchKernelGGL(topk_select_kernel, dim3(1), dim3(256), 0, s, ws, level);
    }
    hipLaunchKernelGGL(topk_tie_count_kernel, dim3(cgrid), dim3(kThreads), 0, s, score, dtype_score, n, n_zero, ws);
    hipLaunchKernelGGL(topk_scan_kernel, dim3(1), dim3(kThreads), 0, s, n_chunks, n_zero, ws);
  }
  hipLaunchKernelGGL(topk_apply_kernel, dim3(cgrid), dim3(kThreads), 0, s, score, dtype_score, x, dtype_x, mask_out,
                     dtype_mask, y_out, dtype_y, n, n_zero, ws);
  return launch_status();
}
