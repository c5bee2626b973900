// tools/tune_reduce2.hip -- round 4 A/B harness for the calibration reductions (not part of the product library).
// Round 2 (tools/tune_reduce.hip) found a ticket (last workgroup reduces a scratch area, ONE launch) slower than fill + atomics; it
// used __threadfence() (agent-scope release + acquire: buffer_wbl2 / buffer_inv of the whole L2) around a returning atomic on ONE
// address.  Questions here: what do the pieces cost -- the fences, the same-address arrivals -- and does a ticket with
// agent-scope (L2-bypassing) partial stores / loads, no cache-wide fence and <= 32 arrivals per address beat fill + atomics?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/tune_reduce2.hip -o tools/tune_reduce2
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
#define AGENT __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ void atomic_min_f(float* a, float v) {
  if (v >= 0.0f) atomicMin((int*)a, (int)f2u(v)); else atomicMax((unsigned*)a, f2u(v));
}
__device__ __forceinline__ void atomic_max_f(float* a, float v) {
  if (v >= 0.0f) atomicMax((int*)a, (int)f2u(v)); else atomicMin((unsigned*)a, f2u(v));
}
__global__ void fill2(float* a, float va, float* b, float vb, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { a[i] = va; if (b) b[i] = vb; }
}

template <int T, int U>
__global__ __launch_bounds__(T) void read_only(const void* __restrict__ in, int64_t n_vec, uint32_t* sink) {
  const int64_t base = (int64_t)blockIdx.x * T * U + threadIdx.x;
  u32x4 raw[U];
#pragma unroll
  for (int u = 0; u < U; u++) raw[u] = __builtin_nontemporal_load((const u32x4*)in + (base + (int64_t)u * T < n_vec ? base + (int64_t)u * T : n_vec - 1));
  uint32_t acc = 0;
#pragma unroll
  for (int u = 0; u < U; u++) acc |= raw[u].x ^ raw[u].y ^ raw[u].z ^ raw[u].w;
  if (acc == 0x12345u) *sink = acc;
}

// bf16 min / max on the PACKED words: as u16, every negative pattern is above every positive one and grows with the magnitude; as
// i16, positives are above negatives and grow with the value.  So with umax = max_u16, imax = max_i16, umin = min_u16 over all words:
//   float min = umax if umax has the sign bit (some negative: the largest magnitude), else umin (all positive: the smallest)
//   float max = imax if imax >= 0 (some positive: the largest), else umin (all negative: the smallest magnitude)
// 3 packed operations per dword (2 elements) instead of 2 widenings + 2 min + 2 max.  (NaN / -0.0 handling is the product's job.)
struct Pk { uint32_t umax, imax, umin; };
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) { uint32_t r; asm volatile("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) { uint32_t r; asm volatile("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ uint32_t pk_max_i16(uint32_t a, uint32_t b) { uint32_t r; asm volatile("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

template <int U, int PK>
__device__ __forceinline__ void lane_minmax(const u32x4 (&raw)[U], float& lo, float& hi) {
  if (PK) {
    Pk p{0u, 0x80008000u, 0xFFFFFFFFu};
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int j = 0; j < 4; j++) { p.umax = pk_max_u16(p.umax, raw[u][j]); p.imax = pk_max_i16(p.imax, raw[u][j]); p.umin = pk_min_u16(p.umin, raw[u][j]); }
    const uint32_t umax = max(p.umax & 0xFFFFu, p.umax >> 16), umin = min(p.umin & 0xFFFFu, p.umin >> 16);
    const int imax = max((int)(int16_t)(p.imax & 0xFFFFu), (int)(int16_t)(p.imax >> 16));
    lo = u2f(((umax & 0x8000u) ? umax : umin) << 16);
    hi = u2f((imax >= 0 ? (uint32_t)imax : umin) << 16);
  } else {
    lo = INFINITY; hi = -INFINITY;
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const float a = u2f(raw[u][j] << 16), b = u2f(raw[u][j] & 0xFFFF0000u);
        lo = fminf(lo, fminf(a, b)); hi = fmaxf(hi, fmaxf(a, b));
      }
  }
}

// FIN 0: atomics (out pre-filled); 1: round-2 ticket (__threadfence x 2); 2: partial only; 4: LIGHT ticket (agent-scope partial stores,
// s_waitcnt, relaxed ticket atomic, agent-scope partial loads); 5: ticket with release / acquire on the atomic (the compiler's fences);
// 6: LIGHT ticket in two levels of <= 16 arrivals per address; 7: relaxed ticket with NO ordering at all (timing only: cost of the arrivals)
// FIN 8: ONE launch, no ticket: workgroup 0 writes the identities to mn / mx with agent-scope stores, waits for them, then publishes
// this launch's EPOCH in a flag word; every workgroup requests the flag word right BEHIND its data loads (vector memory returns in
// order: the value arrives with the last data, for free) and only spins -- re-reading it -- if it does not yet hold the epoch, before
// it issues its (non-returning) atomics.  `cnt[1024 + slot]` is the flag, `epoch` a host counter (unique per launch).
template <int T, int U, int FIN, int PK>
__global__ __launch_bounds__(T) void minmax_tile(const void* __restrict__ in, int64_t n_vec, int tiles_per_group, float* mn, float* mx,
                                                 float* part, unsigned* cnt, unsigned epoch = 0, int n_groups = 1) {
  const int tile = blockIdx.x, g = tile / tiles_per_group;
  const int64_t base = (int64_t)tile * T * U + threadIdx.x;
  unsigned* flag = cnt + 1024 + (epoch & 63u);
  if (FIN == 9 && tile == 0) {   // FIN 9: the identities go out FIRST, the data loads right behind them; the flag follows once only the loads are outstanding
    for (int i = threadIdx.x; i < n_groups; i += T) {
      __hip_atomic_store(&mn[i], INFINITY, __ATOMIC_RELAXED, AGENT);
      __hip_atomic_store(&mx[i], -INFINITY, __ATOMIC_RELAXED, AGENT);
    }
  }
  if (FIN == 8 && tile == 0) {
    for (int i = threadIdx.x; i < n_groups; i += T) {
      __hip_atomic_store(&mn[i], INFINITY, __ATOMIC_RELAXED, AGENT);
      __hip_atomic_store(&mx[i], -INFINITY, __ATOMIC_RELAXED, AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, AGENT);
  }
  u32x4 raw[U];
#pragma unroll
  for (int u = 0; u < U; u++) raw[u] = __builtin_nontemporal_load((const u32x4*)in + (base + (int64_t)u * T < n_vec ? base + (int64_t)u * T : n_vec - 1));
  if (FIN == 9 && tile == 0) {
    static_assert(U == 16 || FIN != 9, "the wait below counts the data loads");
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // the stores in front of the 16 data loads are acknowledged
    __builtin_amdgcn_s_barrier();
    if (threadIdx.x == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, AGENT);
  }
  unsigned seen = 0;
  if ((FIN == 8 || FIN == 9) && threadIdx.x == 0) seen = __hip_atomic_load(flag, __ATOMIC_RELAXED, AGENT);   // issued behind the data loads
  float lo, hi;
  lane_minmax<U, PK>(raw, lo, hi);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o)); }
  __shared__ float s_lo[T / 64], s_hi[T / 64];
  __shared__ int s_last;
  const int w = threadIdx.x / 64;
  if ((threadIdx.x & 63) == 0) { s_lo[w] = lo; s_hi[w] = hi; }
  __syncthreads();
  if (threadIdx.x >= 64) return;     // wave 0 finishes
  lo = threadIdx.x < T / 64 ? s_lo[threadIdx.x] : INFINITY;
  hi = threadIdx.x < T / 64 ? s_hi[threadIdx.x] : -INFINITY;
#pragma unroll
  for (int o = T / 128; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o)); }
  if (FIN == 0) { if (threadIdx.x == 0) { atomic_min_f(&mn[g], lo); atomic_max_f(&mx[g], hi); } return; }
  if (FIN == 8 || FIN == 9) {
    if (threadIdx.x == 0) {
      while (seen != epoch) seen = __hip_atomic_load(flag, __ATOMIC_RELAXED, AGENT);
      atomic_min_f(&mn[g], lo); atomic_max_f(&mx[g], hi);
    }
    return;
  }
  if (FIN == 2) { if (threadIdx.x == 0) { part[2 * tile] = lo; part[2 * tile + 1] = hi; } return; }
  int last = 0;
  if (FIN == 1) {
    if (threadIdx.x == 0) { part[2 * tile] = lo; part[2 * tile + 1] = hi; __threadfence(); last = atomicAdd(&cnt[g], 1u) == (unsigned)tiles_per_group - 1u; }
  } else if (FIN == 5) {
    if (threadIdx.x == 0) { part[2 * tile] = lo; part[2 * tile + 1] = hi; last = __hip_atomic_fetch_add(&cnt[g], 1u, __ATOMIC_ACQ_REL, AGENT) == (unsigned)tiles_per_group - 1u; }
  } else {
    if (threadIdx.x == 0) {
      __hip_atomic_store(&part[2 * tile], lo, __ATOMIC_RELAXED, AGENT);
      __hip_atomic_store(&part[2 * tile + 1], hi, __ATOMIC_RELAXED, AGENT);
      if (FIN != 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int lvl1 = FIN == 6 ? 16 : tiles_per_group;
      unsigned* c1 = FIN == 6 ? &cnt[64 + tile / 16] : &cnt[g];
      last = __hip_atomic_fetch_add(c1, 1u, __ATOMIC_RELAXED, AGENT) == (unsigned)lvl1 - 1u;
    }
  }
  last = __shfl(last, 0);
  if (!last) return;
  if (FIN == 1) __threadfence();
  if (FIN == 6) {   // level 1: 16 partials -> one, then the second ticket
    float l2 = INFINITY, h2 = -INFINITY;
    if (threadIdx.x < 16) {
      l2 = __hip_atomic_load(&part[2 * ((tile & ~15) + threadIdx.x)], __ATOMIC_RELAXED, AGENT);
      h2 = __hip_atomic_load(&part[2 * ((tile & ~15) + threadIdx.x) + 1], __ATOMIC_RELAXED, AGENT);
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) { l2 = fminf(l2, __shfl_xor(l2, o)); h2 = fmaxf(h2, __shfl_xor(h2, o)); }
    int last2 = 0;
    const int n2 = tiles_per_group / 16;
    if (threadIdx.x == 0) {
      cnt[64 + tile / 16] = 0u;
      __hip_atomic_store(&part[8192 + 2 * (tile / 16)], l2, __ATOMIC_RELAXED, AGENT);
      __hip_atomic_store(&part[8192 + 2 * (tile / 16) + 1], h2, __ATOMIC_RELAXED, AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      last2 = __hip_atomic_fetch_add(&cnt[g], 1u, __ATOMIC_RELAXED, AGENT) == (unsigned)n2 - 1u;
    }
    last2 = __shfl(last2, 0);
    if (!last2) return;
    l2 = INFINITY; h2 = -INFINITY;
    for (int i = threadIdx.x; i < n2; i += 64) {
      l2 = fminf(l2, __hip_atomic_load(&part[8192 + 2 * (g * n2 + i)], __ATOMIC_RELAXED, AGENT));
      h2 = fmaxf(h2, __hip_atomic_load(&part[8192 + 2 * (g * n2 + i) + 1], __ATOMIC_RELAXED, AGENT));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { l2 = fminf(l2, __shfl_xor(l2, o)); h2 = fmaxf(h2, __shfl_xor(h2, o)); }
    if (threadIdx.x == 0) { mn[g] = l2; mx[g] = h2; cnt[g] = 0u; }
    return;
  }
  float l2 = INFINITY, h2 = -INFINITY;
  for (int i = threadIdx.x; i < tiles_per_group; i += 64) {
    if (FIN == 1 || FIN == 5) { l2 = fminf(l2, part[2 * (g * tiles_per_group + i)]); h2 = fmaxf(h2, part[2 * (g * tiles_per_group + i) + 1]); }
    else {
      l2 = fminf(l2, __hip_atomic_load(&part[2 * (g * tiles_per_group + i)], __ATOMIC_RELAXED, AGENT));
      h2 = fmaxf(h2, __hip_atomic_load(&part[2 * (g * tiles_per_group + i) + 1], __ATOMIC_RELAXED, AGENT));
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { l2 = fminf(l2, __shfl_xor(l2, o)); h2 = fmaxf(h2, __shfl_xor(h2, o)); }
  if (threadIdx.x == 0) { mn[g] = l2; mx[g] = h2; cnt[g] = 0u; }
}

// per-column max|x| of [rows, cols] bf16: workgroup = strip of 512 columns x (W * U) rows; wave w takes rows w, w+W, ...
// FIN 0: atomics (out pre-zeroed), 2: partial only, 4: light ticket per strip (arrivals = row splits)
// PK: |x| maxima on the packed 16-bit words (v_and + v_pk_max_u16 per dword)
template <int W, int U, int FIN, int PK>
__global__ __launch_bounds__(W * 64) void maxabs_cols(const void* __restrict__ in, int64_t rows, int64_t cols, float* out, float* part, unsigned* cnt,
                                                       unsigned epoch = 0) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t col0 = ((int64_t)blockIdx.x * 64 + lane) * 8;
  const int64_t r0 = (int64_t)blockIdx.y * W * U + w;
  unsigned* flag = cnt + 1024 + (epoch & 63u);
  if (FIN == 8 && blockIdx.x == 0 && blockIdx.y == 0) {   // self-initialising: see minmax_tile FIN 8
    for (int64_t i = threadIdx.x; i < cols; i += W * 64) __hip_atomic_store((uint32_t*)&out[i], 0u, __ATOMIC_RELAXED, AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, AGENT);
  }
  u32x4 raw[U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    const int64_t r = r0 + (int64_t)u * W < rows ? r0 + (int64_t)u * W : rows - 1;
    raw[u] = __builtin_nontemporal_load((const u32x4*)((const uint16_t*)in + r * cols + col0));
  }
  unsigned seen = 0;
  if (FIN == 8 && threadIdx.x == 0) seen = __hip_atomic_load(flag, __ATOMIC_RELAXED, AGENT);
  __shared__ uint32_t sm[W][PK ? 4 : 8][64];
  __shared__ int s_last;
  if (PK) {
    uint32_t m[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int j = 0; j < 4; j++) m[j] = pk_max_u16(m[j], raw[u][j] & 0x7FFF7FFFu);
#pragma unroll
    for (int k = 0; k < 4; k++) sm[w][k][lane] = m[k];
  } else {
    uint32_t m[8];
#pragma unroll
    for (int k = 0; k < 8; k++) m[k] = 0u;
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        m[2 * j] = max(m[2 * j], (raw[u][j] << 16) & 0x7FFFFFFFu);
        m[2 * j + 1] = max(m[2 * j + 1], raw[u][j] & 0x7FFF0000u);
      }
#pragma unroll
    for (int k = 0; k < 8; k++) sm[w][k][lane] = m[k];
  }
  if (FIN == 8) {
    __shared__ unsigned s_seen;
    if (threadIdx.x == 0) {
      while (seen != epoch) seen = __hip_atomic_load(flag, __ATOMIC_RELAXED, AGENT);
      s_seen = seen;
    }
  }
  __syncthreads();
  // 512 columns of the strip: column c = lane l * 8 + k
  for (int c = threadIdx.x; c < 512; c += W * 64) {
    uint32_t r;
    if (PK) {   // thread c: dword c / 2 of the strip's packed row = (lane (c / 8), k (c % 8) / 2), half c & 1
      const int l = c >> 3, k = (c & 7) >> 1;
      uint32_t p = sm[0][k][l];
#pragma unroll
      for (int i = 1; i < W; i++) p = pk_max_u16(p, sm[i][k][l]);
      r = (c & 1) ? (p & 0xFFFF0000u) : (p << 16);
    } else {
      const int l = c >> 3, k = c & 7;
      r = sm[0][k][l];
#pragma unroll
      for (int i = 1; i < W; i++) r = max(r, sm[i][k][l]);
    }
    const int64_t col = (int64_t)blockIdx.x * 512 + c;
    if (FIN == 0 || FIN == 8) atomicMax((unsigned*)&out[col], r);
    else if (FIN == 2) part[(int64_t)blockIdx.y * cols + col] = u2f(r);
    else __hip_atomic_store((uint32_t*)&part[(int64_t)blockIdx.y * cols + col], r, __ATOMIC_RELAXED, AGENT);
  }
  if (FIN == 4) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(&cnt[blockIdx.x], 1u, __ATOMIC_RELAXED, AGENT) == gridDim.y - 1u;
    __syncthreads();
    if (s_last) {
      for (int c = threadIdx.x; c < 512; c += W * 64) {
        const int64_t col = (int64_t)blockIdx.x * 512 + c;
        uint32_t r = 0u;
        for (unsigned y = 0; y < gridDim.y; y++) r = max(r, __hip_atomic_load((uint32_t*)&part[(int64_t)y * cols + col], __ATOMIC_RELAXED, AGENT));
        out[col] = u2f(r);
      }
      if (threadIdx.x == 0) cnt[blockIdx.x] = 0u;
    }
  }
}

static unsigned g_epoch = 0;
struct Variant { std::string name; std::function<void(const void*, hipStream_t)> run; std::vector<float> us; };

int main(int argc, char** argv) {
  const int ROUNDS = argc > 1 ? atoi(argv[1]) : 7;
  const int64_t rows = argc > 2 ? atoll(argv[2]) : 4096, cols = argc > 3 ? atoll(argv[3]) : 4096, n = rows * cols, n_vec = n / 8;
  const int NBUF = (int)std::max<int64_t>(2, std::min<int64_t>(48, (int64_t)1280 * 1024 * 1024 / (n * 2))), LAUNCHES = 50;
  std::vector<void*> in(NBUF);
  std::vector<uint16_t> h(n);
  uint64_t s = 88172645463325252ull;
  for (int64_t i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint16_t)(((s >> 20) & 0x8FFF) | 0x3000) ^ (uint16_t)((s >> 40) & 0x0F00); }
  for (int b = 0; b < NBUF; b++) { CK(hipMalloc(&in[b], n * 2)); CK(hipMemcpy(in[b], h.data(), n * 2, hipMemcpyHostToDevice)); }
  float *mn, *mx, *part; unsigned* cnt; uint32_t* sink;
  CK(hipMalloc(&mn, 65536 * 4)); CK(hipMalloc(&mx, 65536 * 4)); CK(hipMalloc(&part, 64 << 20)); CK(hipMalloc(&cnt, 65536 * 4)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(cnt, 0, 65536 * 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  std::vector<Variant> vs;
  static const char* fin_name[] = {"fill+atomics", "ticket __threadfence (round 2)", "partial-only", "", "LIGHT ticket", "ticket acq_rel atomic", "LIGHT ticket 2-level (16 x 16)", "ticket, no ordering (timing only)", "SELF-INIT: wg 0 fills + epoch flag read behind the loads", "SELF-INIT, wg 0: stores, loads, vmcnt(16), flag"};
#define ADD_READ(T, U) vs.push_back({"read        T" #T " U" #U, [=](const void* i, hipStream_t q) { \
    hipLaunchKernelGGL((read_only<T, U>), dim3((unsigned)((n_vec + T * U - 1) / (T * U))), dim3(T), 0, q, i, n_vec, sink); }, {}})
#define ADD_MM(T, U, FIN, G, PK) vs.push_back({std::string("minmax G" #G " T" #T " U" #U " ") + (PK ? "pk16 " : "f32  ") + fin_name[FIN], [=](const void* i, hipStream_t q) { \
    const int tiles = (int)((n_vec + T * U - 1) / (T * U)); \
    if (FIN == 0) hipLaunchKernelGGL(fill2, dim3((G + 255) / 256), dim3(256), 0, q, mn, INFINITY, mx, -INFINITY, G); \
    hipLaunchKernelGGL((minmax_tile<T, U, FIN, PK>), dim3(tiles), dim3(T), 0, q, i, n_vec, std::max(1, tiles / G), mn, mx, part, cnt, ++g_epoch, (int)G); }, {}})
#define ADD_MA(W, U, FIN, PK) vs.push_back({std::string("maxabs W" #W " U" #U " ") + (PK ? "pk16 " : "u32  ") + fin_name[FIN], [=](const void* i, hipStream_t q) { \
    if (FIN == 0) hipLaunchKernelGGL(fill2, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, q, mn, 0.0f, (float*)nullptr, 0.0f, (int)cols); \
    hipLaunchKernelGGL((maxabs_cols<W, U, FIN, PK>), dim3((unsigned)(cols / 512), (unsigned)((rows + W * U - 1) / (W * U))), dim3(W * 64), 0, q, i, rows, cols, mn, part, cnt, ++g_epoch); }, {}})
  ADD_READ(512, 16); ADD_READ(256, 8);
  ADD_MM(512, 16, 0, 1, 0); ADD_MM(512, 16, 2, 1, 0); ADD_MM(512, 16, 2, 1, 1); ADD_MM(512, 16, 1, 1, 0); ADD_MM(512, 16, 5, 1, 0); ADD_MM(512, 16, 4, 1, 0);
  ADD_MM(512, 16, 7, 1, 0); ADD_MM(512, 16, 6, 1, 0); ADD_MM(512, 16, 6, 1, 1); ADD_MM(512, 16, 4, 1, 1);
  ADD_MM(512, 16, 0, 32, 0); ADD_MM(512, 16, 2, 32, 0); ADD_MM(512, 16, 4, 32, 0); ADD_MM(512, 16, 4, 32, 1); ADD_MM(512, 16, 5, 32, 0);
  ADD_MM(512, 16, 8, 1, 1); ADD_MM(512, 16, 8, 32, 1); ADD_MM(512, 16, 8, 1, 0); ADD_MM(512, 16, 9, 1, 1); ADD_MM(512, 16, 9, 32, 1); ADD_MM(512, 16, 2, 32, 1);
  ADD_MM(256, 16, 2, 1, 1); ADD_MM(256, 16, 6, 1, 1); ADD_MM(256, 16, 4, 32, 1); ADD_MM(256, 8, 2, 1, 1); ADD_MM(1024, 8, 4, 32, 1); ADD_MM(1024, 8, 6, 1, 1);
  ADD_MA(16, 8, 0, 0); ADD_MA(16, 8, 2, 0); ADD_MA(16, 8, 2, 1); ADD_MA(16, 8, 4, 0); ADD_MA(16, 8, 4, 1);
  ADD_MA(16, 8, 8, 1); ADD_MA(16, 8, 8, 0);
  ADD_MA(8, 16, 2, 1); ADD_MA(8, 16, 4, 1); ADD_MA(8, 8, 4, 1); ADD_MA(4, 16, 4, 1); ADD_MA(16, 4, 4, 1); ADD_MA(4, 8, 4, 1);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // every finishing form must reproduce fill + atomics
  {
    std::vector<float> ref_mm(64), ref_ma(cols), got(std::max<int64_t>(cols, 64));
    for (auto& v : vs) {
      const bool mm = v.name.rfind("minmax", 0) == 0, ma = v.name.rfind("maxabs", 0) == 0;
      if (!(mm || ma) || v.name.find("partial-only") != std::string::npos || v.name.find("timing only") != std::string::npos) continue;
      CK(hipMemset(mn, 0x55, 65536 * 4)); CK(hipMemset(mx, 0x55, 65536 * 4));
      if (v.name.find("fill+atomics") != std::string::npos) {
        v.run(in[0], st); CK(hipStreamSynchronize(st));
        if (mm) { const int G = v.name.find("G32") != std::string::npos ? 32 : 1; CK(hipMemcpy(ref_mm.data() + (G == 1 ? 0 : 2), mn, 4 * G, hipMemcpyDeviceToHost)); if (G == 1) CK(hipMemcpy(ref_mm.data() + 1, mx, 4, hipMemcpyDeviceToHost)); }
        else CK(hipMemcpy(ref_ma.data(), mn, 4 * cols, hipMemcpyDeviceToHost));
        continue;
      }
      for (int rep = 0; rep < 3; rep++) { v.run(in[0], st); }
      CK(hipStreamSynchronize(st));
      bool ok = true;
      if (mm) {
        const int G = v.name.find("G32") != std::string::npos ? 32 : 1;
        CK(hipMemcpy(got.data(), mn, 4 * G, hipMemcpyDeviceToHost));
        if (G == 1) { float hx; CK(hipMemcpy(&hx, mx, 4, hipMemcpyDeviceToHost)); ok = got[0] == ref_mm[0] && hx == ref_mm[1]; }
        else for (int g = 0; g < 32; g++) ok &= got[g] == ref_mm[2 + g];
      } else {
        CK(hipMemcpy(got.data(), mn, 4 * cols, hipMemcpyDeviceToHost));
        for (int64_t c = 0; c < cols; c++) ok &= got[c] == ref_ma[c];
      }
      if (!ok) printf("# MISMATCH: %s\n", v.name.c_str());
    }
    printf("# finishing forms checked against fill + atomics\n");
  }
  for (auto& v : vs) for (int i = 0; i < 10; i++) v.run(in[i % NBUF], st);
  CK(hipStreamSynchronize(st));
  for (int r = 0; r < ROUNDS; r++)
    for (auto& v : vs) {
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < LAUNCHES; i++) v.run(in[i % NBUF], st);
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      v.us.push_back(ms * 1e3f / LAUNCHES);
    }
  printf("# rows %lld cols %lld bf16, %d rotating buffers; read-only roofline = 2 B/element over 8 TB/s\n", (long long)rows, (long long)cols, NBUF);
  printf("%-64s %9s %9s %9s %8s\n", "variant", "min_us", "med_us", "TB/s(med)", "%8TB/s");
  for (auto& v : vs) {
    std::sort(v.us.begin(), v.us.end());
    float med = v.us[v.us.size() / 2], mnu = v.us[0];
    double tbs = 2.0 * n / (med * 1e-6) / 1e12;
    printf("%-64s %9.2f %9.2f %9.3f %7.1f%%\n", v.name.c_str(), mnu, med, tbs, 100.0 * tbs / 8.0);
  }
  return 0;
}
