// tools/tune_reduce.hip — on-GPU A/B harness for the calibration reductions (not part of the product library).
// Question it answers: how far above the plain 16-byte read of the same tensor do min/max (per tensor, per group of rows) and
// per-column max|x| sit, as a function of (a) loads in flight per lane and workgroup shape, (b) how the partial results are
// combined: atomics into a pre-filled output (fill launch + kernel) or a ticket (last workgroup reduces a scratch area: one launch).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/tune_reduce.hip -o /tmp/tune_reduce
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

// FIN 0: atomics (out pre-filled); 1: ticket; 2: partial only
template <int T, int U, int FIN>
__global__ __launch_bounds__(T) void minmax_tile(const void* __restrict__ in, int64_t n_vec, int tiles_per_group, float* mn, float* mx,
                                                 float2* part, unsigned* cnt) {
  const int tile = blockIdx.x, g = tile / tiles_per_group;
  const int64_t base = (int64_t)tile * T * U + threadIdx.x;
  u32x4 raw[U];
#pragma unroll
  for (int u = 0; u < U; u++) raw[u] = __builtin_nontemporal_load((const u32x4*)in + (base + (int64_t)u * T < n_vec ? base + (int64_t)u * T : n_vec - 1));
  float lo = INFINITY, hi = -INFINITY;
#pragma unroll
  for (int u = 0; u < U; u++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const float a = u2f(raw[u][j] << 16), b = u2f(raw[u][j] & 0xFFFF0000u);
      lo = fminf(lo, fminf(a, b)); hi = fmaxf(hi, fmaxf(a, b));
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o)); }
  __shared__ float s_lo[T / 64], s_hi[T / 64];
  __shared__ int s_last;
  const int w = threadIdx.x / 64;
  if ((threadIdx.x & 63) == 0) { s_lo[w] = lo; s_hi[w] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 1; i < T / 64; i++) { lo = fminf(lo, s_lo[i]); hi = fmaxf(hi, s_hi[i]); }
    if (FIN == 0) { atomic_min_f(&mn[g], lo); atomic_max_f(&mx[g], hi); }
    else if (FIN == 3) {  // filter: the output only moves one way, so a (possibly stale) read that already beats ours makes the atomic a no-op
      if (lo < __hip_atomic_load(&mn[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomic_min_f(&mn[g], lo);
      if (hi > __hip_atomic_load(&mx[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomic_max_f(&mx[g], hi);
    } else {
      part[tile] = make_float2(lo, hi);
      if (FIN == 1) {
        __threadfence();
        const unsigned t = atomicAdd(&cnt[g], 1u);
        s_last = (t == (unsigned)tiles_per_group - 1u);
      }
    }
  }
  if (FIN == 1) {
    __syncthreads();
    if (s_last) {
      __threadfence();
      float l2 = INFINITY, h2 = -INFINITY;
      for (int i = threadIdx.x; i < tiles_per_group; i += T) {
        const float* pp = (const float*)&part[g * tiles_per_group + i];
        l2 = fminf(l2, __builtin_nontemporal_load(pp)); h2 = fmaxf(h2, __builtin_nontemporal_load(pp + 1));
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { l2 = fminf(l2, __shfl_xor(l2, o)); h2 = fmaxf(h2, __shfl_xor(h2, o)); }
      __syncthreads();
      if ((threadIdx.x & 63) == 0) { s_lo[w] = l2; s_hi[w] = h2; }
      __syncthreads();
      if (threadIdx.x == 0) {
        for (int i = 1; i < T / 64; i++) { l2 = fminf(l2, s_lo[i]); h2 = fmaxf(h2, s_hi[i]); }
        mn[g] = l2; mx[g] = h2; cnt[g] = 0u;
      }
    }
  }
}

// per-column max|x| of [rows, cols] bf16: workgroup = strip of 512 columns x (W * U) rows; wave w takes rows w, w+W, ...
// FIN 0: atomics (out pre-zeroed), 1: ticket per strip, 2: partial only
template <int W, int U, int FIN>
__global__ __launch_bounds__(W * 64) void maxabs_cols(const void* __restrict__ in, int64_t rows, int64_t cols, float* out, float* part, unsigned* cnt) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t col0 = ((int64_t)blockIdx.x * 64 + lane) * 8;
  const int64_t r0 = (int64_t)blockIdx.y * W * U + w;
  u32x4 raw[U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    const int64_t r = r0 + (int64_t)u * W < rows ? r0 + (int64_t)u * W : rows - 1;
    raw[u] = __builtin_nontemporal_load((const u32x4*)((const uint16_t*)in + r * cols + col0));
  }
  uint32_t m[8];  // |x| as bit patterns of the bf16 widened: integer max == float max for non-negative values
#pragma unroll
  for (int k = 0; k < 8; k++) m[k] = 0u;
#pragma unroll
  for (int u = 0; u < U; u++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      m[2 * j] = max(m[2 * j], (raw[u][j] << 16) & 0x7FFFFFFFu);
      m[2 * j + 1] = max(m[2 * j + 1], raw[u][j] & 0x7FFF0000u);
    }
  __shared__ uint32_t sm[W][8][64];
  __shared__ int s_last;
#pragma unroll
  for (int k = 0; k < 8; k++) sm[w][k][lane] = m[k];
  __syncthreads();
  // 512 columns, W*64 threads: thread t reduces column t (and t + W*64 ...) over the W waves
  for (int c = threadIdx.x; c < 512; c += W * 64) {
    const int l = c >> 3, k = c & 7;
    uint32_t r = sm[0][k][l];
#pragma unroll
    for (int i = 1; i < W; i++) r = max(r, sm[i][k][l]);
    const int64_t col = (int64_t)blockIdx.x * 512 + c;
    if (FIN == 0) atomicMax((unsigned*)&out[col], r);
    else if (FIN == 3) { if (r > __hip_atomic_load((unsigned*)&out[col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax((unsigned*)&out[col], r); }
    else part[(int64_t)blockIdx.y * cols + col] = u2f(r);
  }
  if (FIN == 1) {
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) s_last = (atomicAdd(&cnt[blockIdx.x], 1u) == gridDim.y - 1u);
    __syncthreads();
    if (s_last) {
      __threadfence();
      for (int c = threadIdx.x; c < 512; c += W * 64) {
        const int64_t col = (int64_t)blockIdx.x * 512 + c;
        uint32_t r = 0u;
        for (unsigned y = 0; y < gridDim.y; y++) r = max(r, f2u(__builtin_nontemporal_load(&part[(int64_t)y * cols + col])));
        out[col] = u2f(r);
      }
      if (threadIdx.x == 0) cnt[blockIdx.x] = 0u;
    }
  }
}

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
#define ADD_READ(T, U) vs.push_back({"read        T" #T " U" #U, [=](const void* i, hipStream_t q) { \
    hipLaunchKernelGGL((read_only<T, U>), dim3((unsigned)((n_vec + T * U - 1) / (T * U))), dim3(T), 0, q, i, n_vec, sink); }, {}})
#define ADD_MM(T, U, FIN, G) vs.push_back({std::string("minmax G" #G " T" #T " U" #U) + (FIN == 0 ? " fill+atomics" : FIN == 1 ? " ticket" : FIN == 3 ? " fill+filtered atomics" : " partial-only"), [=](const void* i, hipStream_t q) { \
    const int tiles = (int)((n_vec + T * U - 1) / (T * U)); \
    if (FIN == 0 || FIN == 3) hipLaunchKernelGGL(fill2, dim3((G + 255) / 256), dim3(256), 0, q, mn, INFINITY, mx, -INFINITY, G); \
    hipLaunchKernelGGL((minmax_tile<T, U, FIN>), dim3(tiles), dim3(T), 0, q, i, n_vec, std::max(1, tiles / G), mn, mx, (float2*)part, cnt); }, {}})
#define ADD_MA(W, U, FIN) vs.push_back({std::string("maxabs W" #W " U" #U) + (FIN == 0 ? " fill+atomics" : FIN == 1 ? " ticket" : FIN == 3 ? " fill+filtered atomics" : " partial-only"), [=](const void* i, hipStream_t q) { \
    if (FIN == 0 || FIN == 3) hipLaunchKernelGGL(fill2, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, q, mn, 0.0f, (float*)nullptr, 0.0f, (int)cols); \
    hipLaunchKernelGGL((maxabs_cols<W, U, FIN>), dim3((unsigned)(cols / 512), (unsigned)((rows + W * U - 1) / (W * U))), dim3(W * 64), 0, q, i, rows, cols, mn, part, cnt); }, {}})
  ADD_READ(512, 16); ADD_READ(256, 8); ADD_READ(1024, 16);
  ADD_MM(512, 16, 0, 1); ADD_MM(512, 16, 1, 1); ADD_MM(512, 16, 2, 1); ADD_MM(512, 16, 3, 1);
  ADD_MM(1024, 16, 0, 1); ADD_MM(1024, 16, 2, 1); ADD_MM(1024, 16, 3, 1); ADD_MM(1024, 8, 3, 1); ADD_MM(256, 16, 3, 1); ADD_MM(256, 8, 3, 1); ADD_MM(512, 8, 3, 1);
  ADD_MM(512, 16, 0, 32); ADD_MM(512, 16, 3, 32); ADD_MM(512, 16, 2, 32); ADD_MM(256, 16, 3, 32); ADD_MM(512, 8, 3, 32);
  ADD_MA(16, 8, 0); ADD_MA(16, 8, 3); ADD_MA(16, 8, 2); ADD_MA(8, 16, 3); ADD_MA(8, 8, 3); ADD_MA(4, 16, 3); ADD_MA(16, 4, 3); ADD_MA(16, 16, 3);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (auto& v : vs) for (int i = 0; i < 10; i++) v.run(in[i % NBUF], st);
  CK(hipStreamSynchronize(st));
  // check the ticket results against the atomics results once
  {
    std::vector<float> a(2), b(2);
    vs[3].run(in[0], st); CK(hipStreamSynchronize(st)); CK(hipMemcpy(&a[0], mn, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&a[1], mx, 4, hipMemcpyDeviceToHost));
    CK(hipMemset(mn, 0, 4)); CK(hipMemset(mx, 0, 4));
    vs[4].run(in[0], st); CK(hipStreamSynchronize(st)); CK(hipMemcpy(&b[0], mn, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&b[1], mx, 4, hipMemcpyDeviceToHost));
    printf("# minmax atomics (%g, %g) ticket (%g, %g) %s\n", a[0], a[1], b[0], b[1], (a == b) ? "same" : "DIFFERENT");
  }
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
  printf("%-44s %9s %9s %9s %8s\n", "variant", "min_us", "med_us", "TB/s(med)", "%8TB/s");
  for (auto& v : vs) {
    std::sort(v.us.begin(), v.us.end());
    float med = v.us[v.us.size() / 2], mnu = v.us[0];
    double tbs = 2.0 * n / (med * 1e-6) / 1e12;
    printf("%-44s %9.2f %9.2f %9.3f %7.1f%%\n", v.name.c_str(), mnu, med, tbs, 100.0 * tbs / 8.0);
  }
  return 0;
}
