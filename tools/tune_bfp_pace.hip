// tools/tune_bfp_pace.hip — on-GPU A/B harness (not part of the product library): the hot kernel (bfp_rows_kernel, bf16, B = 16, nearest) at the
// geometries of rows_plan's size classes over paced loads (profiles/r05_tune_pace.txt sections 3 and 10).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math -ffp-contract=off -fno-gpu-flush-denormals-to-zero -mllvm -amdgpu-kernarg-preload-count=16 \
//         -Iinclude tools/tune_bfp_pace.hip -o tools/tune_bfp_pace
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>
#include "../dmx-compressor_amd/csrc/bfp_rows.hpp"
using namespace dmxq;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
struct Variant { std::string name; std::function<void(const void*, void*, hipStream_t)> run; std::vector<float> us; };
int main(int argc, char** argv) {
  const int ROUNDS = argc > 1 ? atoi(argv[1]) : 7;
  const int64_t max_n = (int64_t)6144 * 4096;
  const int NBUF = 12, LAUNCHES = 50;
  std::vector<void*> in(NBUF), out(NBUF);
  std::vector<uint16_t> h(max_n);
  uint64_t s = 88172645463325252ull;
  for (int64_t i = 0; i < max_n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint16_t)(((s >> 20) & 0x8FFF) | 0x3000) ^ (uint16_t)((s >> 40) & 0x0F00); }
  for (int b = 0; b < NBUF; b++) { CK(hipMalloc(&in[b], max_n * 2)); CK(hipMalloc(&out[b], max_n * 2)); CK(hipMemcpy(in[b], h.data(), max_n * 2, hipMemcpyHostToDevice)); }
  hipStream_t st; CK(hipStreamCreate(&st));
  std::vector<Variant> vs;
  constexpr int MODE = kRowsNtLoad | kRowsNtStore;
#define ADD(ROWS, T, U, P) vs.push_back({"rows " #ROWS " " #T "x" #U " pace " #P, [=](const void* i, void* o, hipStream_t q) { \
    const int64_t n_vec = (int64_t)ROWS * 512; int g = (int)((n_vec + (int64_t)T * U - 1) / ((int64_t)T * U)); \
    hipLaunchKernelGGL((bfp_rows_kernel<DMXQ_BF16, DMXQ_BF16, DMXQ_ROUND_NEAREST, false, U, MODE, T, 2, U, 16, 0, P>), dim3(g), dim3(T), 0, q, i, o, n_vec, 2, 8, 2, 0ull); }, {}})
#define SWEEP(ROWS, T, U) ADD(ROWS, T, U, 0); ADD(ROWS, T, U, 1); ADD(ROWS, T, U, 2); ADD(ROWS, T, U, 3); ADD(ROWS, T, U, 4)
  SWEEP(1024, 128, 2); SWEEP(1792, 128, 2); SWEEP(2048, 512, 4); SWEEP(2560, 128, 8); SWEEP(3072, 512, 12); SWEEP(3584, 512, 14); SWEEP(4608, 512, 18); SWEEP(6144, 512, 2);
  SWEEP(2048, 128, 8); SWEEP(2048, 256, 8); SWEEP(2560, 256, 8); SWEEP(6144, 512, 4);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 20; w++) for (auto& v : vs) for (int i = 0; i < 10; i++) v.run(in[i % NBUF], out[i % NBUF], st);
  CK(hipStreamSynchronize(st)); CK(hipGetLastError());
  for (int r = 0; r < ROUNDS; r++)
    for (auto& v : vs) {
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < LAUNCHES; i++) v.run(in[i % NBUF], out[i % NBUF], st);
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); v.us.push_back(ms * 1e3f / LAUNCHES);
    }
  for (auto& v : vs) { std::sort(v.us.begin(), v.us.end()); float med = v.us[v.us.size() / 2];
    printf("%-34s min %6.2f med %6.2f  %5.1f%%\n", v.name.c_str(), v.us[0], med, 0.0); }
  return 0;
}
