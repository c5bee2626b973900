#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int T, int U>
__global__ __launch_bounds__(T) void rd(const char* __restrict__ in, uint32_t* sink) {
  const char* src = in + (size_t)blockIdx.x * T * U * 16 + threadIdx.x * 16;
  u32x4 r[U];
#pragma unroll
  for (int u = 0; u < U; u++) r[u] = __builtin_nontemporal_load((const u32x4*)(src + u * T * 16));
  uint32_t x = 0;
#pragma unroll
  for (int u = 0; u < U; u++) x ^= r[u].x ^ r[u].y ^ r[u].z ^ r[u].w;
  if (x == 0x12345678u) sink[0] = x;
}
template <int T, int U>
__global__ __launch_bounds__(T) void wr(char* __restrict__ out, uint32_t v) {
  char* dst = out + (size_t)blockIdx.x * T * U * 16 + threadIdx.x * 16;
#pragma unroll
  for (int u = 0; u < U; u++) __builtin_nontemporal_store(u32x4{v, v + u, v, v}, (u32x4*)(dst + u * T * 16));
}
template <int T, int U>
__global__ __launch_bounds__(T) void cp(const char* __restrict__ in, char* __restrict__ out) {
  const char* src = in + (size_t)blockIdx.x * T * U * 16 + threadIdx.x * 16;
  char* dst = out + (size_t)blockIdx.x * T * U * 16 + threadIdx.x * 16;
  u32x4 r[U];
#pragma unroll
  for (int u = 0; u < U; u++) r[u] = __builtin_nontemporal_load((const u32x4*)(src + u * T * 16));
#pragma unroll
  for (int u = 0; u < U; u++) __builtin_nontemporal_store(r[u], (u32x4*)(dst + u * T * 16));
}
__global__ void empty() {}
int main() {
  const size_t n = 32u << 20;  // bytes per tensor
  const int NB = 20;
  char *a, *b; uint32_t* sink;
  hipMalloc(&a, n * NB); hipMalloc(&b, n * NB); hipMalloc(&sink, 4);
  hipMemset(a, 1, n * NB); hipMemset(b, 2, n * NB);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](const char* name, auto launch, double bytes) {
    for (int i = 0; i < 50; i++) launch(i % NB);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 400; i++) launch(i % NB);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %7.2f us  %7.1f GB/s\n", name, ms * 1000 / 400, bytes / (ms / 400 * 1e-3) / 1e9);
  };
  constexpr int T = 512, U = 16; const int grid = n / (T * U * 16);
  time("empty kernel", [&](int i) { empty<<<256, 512>>>(); }, 0);
  time("read 32 MiB  (512x16)", [&](int i) { rd<T, U><<<grid, T>>>(a + i * n, sink); }, n);
  time("write 32 MiB (512x16)", [&](int i) { wr<T, U><<<grid, T>>>(b + i * n, i); }, n);
  time("copy 32+32   (512x16)", [&](int i) { cp<T, U><<<grid, T>>>(a + i * n, b + i * n); }, 2.0 * n);
  time("read 32 MiB  (256x4)", [&](int i) { rd<256, 4><<<n / (256 * 4 * 16), 256>>>(a + i * n, sink); }, n);
  time("write 32 MiB (256x4)", [&](int i) { wr<256, 4><<<n / (256 * 4 * 16), 256>>>(b + i * n, i); }, n);
  time("copy 32+32   (256x4)", [&](int i) { cp<256, 4><<<n / (256 * 4 * 16), 256>>>(a + i * n, b + i * n); }, 2.0 * n);
  return 0;
}
