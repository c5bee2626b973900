#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_u __attribute__((aligned(4)));
__global__ void copy_unal(const char* __restrict__ in, char* __restrict__ out, int64_t nvec, int off) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < nvec; i += stride) {
    u32x4 v = *(const u32x4_u*)(in + off + i * 16);
    *(u32x4_u*)(out + off + i * 16) = v;
  }
}
int main() {
  const int64_t n = 64 << 20;  // bytes
  char *a, *b;
  hipMalloc(&a, 8 * n + 64); hipMalloc(&b, 8 * n + 64);
  hipMemset(a, 1, 8 * n + 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int off : {0, 8, 4, 2, 1}) {
    for (int w = 0; w < 3; w++) copy_unal<<<4096, 256>>>(a, b, n / 16, off);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 20; r++) copy_unal<<<4096, 256>>>(a + (r % 8) * n, b + (r % 8) * n, n / 16, off);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipError_t err = hipGetLastError();
    printf("offset %d: %.2f us per 64 MiB copy, %.1f GB/s  (%s)\n", off, ms * 1000 / 20, 2.0 * n / (ms / 20 * 1e-3) / 1e9, hipGetErrorString(err));
  }
  // correctness
  unsigned char* h = (unsigned char*)malloc(4096);
  for (int i = 0; i < 4096; i++) h[i] = (unsigned char)(i * 7 + 3);
  hipMemcpy(a, h, 4096, hipMemcpyHostToDevice); hipMemset(b, 0, 4096);
  copy_unal<<<1, 64>>>(a, b, 64, 2);
  unsigned char* g = (unsigned char*)malloc(4096); hipMemcpy(g, b, 4096, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 2; i < 2 + 1024; i++) bad += g[i] != h[i];
  printf("unaligned copy mismatches: %d\n", bad);
  return 0;
}
