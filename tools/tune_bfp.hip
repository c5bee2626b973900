// tools/tune_bfp.hip — on-GPU A/B harness for the hot kernel (not part of the product library).
// Interleaved rounds in ONE process (cdna_hip_programming.md §5.4 rule 24), rotating over NBUF buffer pairs
// (> 256 MiB Infinity Cache) so every launch streams HBM; a plain 16-byte copy kernel of the same launch
// geometry is the measured ceiling on the same box (rule 10).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math -ffp-contract=off tools/tune_bfp.hip -o /tmp/tune_bfp
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

template <int UNROLL, int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void copy_kernel(const void* __restrict__ in, void* __restrict__ out, int64_t n_vec) {
  constexpr bool NTL = (MODE & 1) != 0, NTS = (MODE & 2) != 0, CONTIG = (MODE & 4) != 0, WCONTIG = (MODE & 8) != 0;
  const int64_t step = WCONTIG ? 64 : (CONTIG ? (int64_t)THREADS : (int64_t)gridDim.x * THREADS);
  const int64_t sweep = (int64_t)gridDim.x * THREADS * UNROLL;
  int64_t v = WCONTIG ? (int64_t)blockIdx.x * THREADS * UNROLL + (int64_t)(threadIdx.x / 64) * 64 * UNROLL + (threadIdx.x & 63)
              : (CONTIG ? (int64_t)blockIdx.x * THREADS * UNROLL + threadIdx.x : (int64_t)blockIdx.x * THREADS + threadIdx.x);
  for (; v < n_vec; v += sweep) {
    u32x4 raw[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) if (v + u * step < n_vec) raw[u] = load_raw16<NTL>(in, (v + u * step) * 16);
#pragma unroll
    for (int u = 0; u < UNROLL; u++) if (v + u * step < n_vec) {
      u32x4* dst = (u32x4*)((char*)out + (v + u * step) * 16);
      if (NTS) __builtin_nontemporal_store(raw[u], dst); else *dst = raw[u];
    }
  }
}

struct Variant { std::string name; std::function<void(const void*, void*, hipStream_t)> run; std::vector<float> us; };

int main(int argc, char** argv) {
  // usage: tune_bfp [rounds] [rows] [cols] [alloc_rows]   (alloc_rows >= rows: size of each hipMalloc in rows, to
  // separate the effect of the allocation size / alignment from the amount of work)
  const int64_t rows = argc > 2 ? atoll(argv[2]) : 4096, cols = argc > 3 ? atoll(argv[3]) : 4096, n = rows * cols, n_vec = n / 8;
  const int64_t alloc_rows = argc > 4 ? std::max<int64_t>(rows, atoll(argv[4])) : rows, n_alloc = alloc_rows * cols;
  const int NBUF = (int)std::max<int64_t>(2, std::min<int64_t>(48, (int64_t)1280 * 1024 * 1024 / (n * 4))), LAUNCHES = 50, ROUNDS = argc > 1 ? atoi(argv[1]) : 7;
  std::vector<void*> in(NBUF), out(NBUF);
  std::vector<uint16_t> h(n);
  uint64_t s = 88172645463325252ull;
  for (int64_t i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint16_t)(((s >> 20) & 0x8FFF) | 0x3000) ^ (uint16_t)((s >> 40) & 0x0F00); }
  // TUNE_CARVE=1: all inputs, then all outputs, carved back to back out of ONE allocation (what a caching allocator hands out from a
  // recycled block) instead of one hipMalloc each
  if (getenv("TUNE_CARVE") && atoi(getenv("TUNE_CARVE"))) {
    char* big; CK(hipMalloc((void**)&big, (size_t)n_alloc * 2 * 2 * NBUF));
    for (int b = 0; b < NBUF; b++) { in[b] = big + (size_t)b * n_alloc * 2; out[b] = big + (size_t)(NBUF + b) * n_alloc * 2; CK(hipMemcpy(in[b], h.data(), n * 2, hipMemcpyHostToDevice)); }
  } else
  for (int b = 0; b < NBUF; b++) { CK(hipMalloc(&in[b], n_alloc * 2)); CK(hipMalloc(&out[b], n_alloc * 2)); CK(hipMemcpy(in[b], h.data(), n * 2, hipMemcpyHostToDevice)); }
  printf("# rows %lld cols %lld alloc_rows %lld nbuf %d  in[0]=%p out[0]=%p in[1]=%p\n", (long long)rows, (long long)cols, (long long)alloc_rows, NBUF, in[0], out[0], in[1]);
  hipStream_t st; CK(hipStreamCreate(&st));
  std::vector<Variant> vs;
  // TUNE_LDS=<KiB>: reserve that much dynamic LDS per workgroup when the grid is one round of <= 256 workgroups, so that the dispatcher
  // cannot place two of them on one CU (160 KiB) and leave another CU idle
  static const size_t lds_reserve = getenv("TUNE_LDS") ? (size_t)atoi(getenv("TUNE_LDS")) * 1024 : 0;
#define ADD_BFPG(U, M, T, GRID, F, GR) vs.push_back({"bfp  U" #U " M" #M " T" #T " G" #GRID " F" #F " grp" #GR, [=](const void* i, void* o, hipStream_t q) { \
    int g = (GRID) > 0 ? (GRID) : (int)((n_vec + (int64_t)T * U - 1) / ((int64_t)T * U)); \
    auto kern = bfp_rows_kernel<DMXQ_BF16, DMXQ_BF16, DMXQ_ROUND_NEAREST, false, U, M, T, F, GR>; \
    const size_t lds = (g <= 256 && (U) >= 4) ? lds_reserve : 0; \
    static bool attr = false; \
    if (lds && !attr) { CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr = true; } \
    hipLaunchKernelGGL(kern, dim3(g), dim3(T), lds, q, i, o, n_vec, 2, 8, 2, 0ull); }, {}})
#define ADD_BFPL(U, M, T, GRID, F, GR, LPBC) vs.push_back({"bfp  U" #U " M" #M " T" #T " G" #GRID " F" #F " grp" #GR " lpbc" #LPBC, [=](const void* i, void* o, hipStream_t q) { \
    int g = (GRID) > 0 ? (GRID) : (int)((n_vec + (int64_t)T * U - 1) / ((int64_t)T * U)); \
    hipLaunchKernelGGL((bfp_rows_kernel<DMXQ_BF16, DMXQ_BF16, DMXQ_ROUND_NEAREST, false, U, M, T, F, GR, 16, LPBC>), dim3(g), dim3(T), 0, q, i, o, n_vec, 2, 8, 2, 0ull); }, {}})
#define ADD_BFPF(U, M, T, GRID, F) ADD_BFPG(U, M, T, GRID, F, U)
#define ADD_BFP(U, M, T, GRID) ADD_BFPF(U, M, T, GRID, 0)
#define ADD_COPY(U, M, T, GRID) vs.push_back({"copy U" #U " M" #M " T" #T " G" #GRID, [=](const void* i, void* o, hipStream_t q) { \
    int g = (GRID) > 0 ? (GRID) : (int)((n_vec + (int64_t)T * U - 1) / ((int64_t)T * U)); \
    hipLaunchKernelGGL((copy_kernel<U, M, T>), dim3(g), dim3(T), 0, q, i, o, n_vec); }, {}})
  // G0 = exact grid.  M bits: 1 nt-load, 2 nt-store, 4 sc1-store, 8 sc0-store (copy: 4 = workgroup-contiguous tiles; bfp is always tiled)
#define ADD_BFPB(ASYM_, LPB_, LPBC_) vs.push_back({"bfp  512x16 " #ASYM_ " lpb" #LPB_ " lpbc" #LPBC_, [=](const void* i, void* o, hipStream_t q) { \
    int g = (int)((n_vec + (int64_t)512 * 16 - 1) / ((int64_t)512 * 16)); \
    hipLaunchKernelGGL((bfp_rows_kernel<DMXQ_BF16, DMXQ_BF16, DMXQ_ROUND_NEAREST, ASYM_, 16, 3, 512, 2, 16, 16, LPBC_>), dim3(g), dim3(512), 0, q, i, o, n_vec, LPB_, 8, 2, 0ull); }, {}})
#ifdef TUNE_COMPACT   // -DTUNE_COMPACT: the compact (results in place) one-round kernel at 16 .. 24 vectors per lane against 512 x 2 / 512 x 16 / 17 / 18
#define ADD_CP(U, GR) vs.push_back({"compact U" #U " T512 grp" #GR, [=](const void* i, void* o, hipStream_t q) { \
    int g = (int)((n_vec + (int64_t)512 * U - 1) / ((int64_t)512 * U)); \
    hipLaunchKernelGGL((bfp_rows_compact_kernel<DMXQ_BF16, U, 512, GR>), dim3(g), dim3(512), 0, q, i, o, n_vec, 2, 8); }, {}})
  ADD_COPY(2, 7, 512, 0); ADD_BFPG(2, 3, 512, 0, 2, 2); ADD_BFPG(16, 3, 512, 0, 2, 16); ADD_BFPG(18, 3, 512, 0, 2, 18);
  ADD_CP(16, 16); ADD_CP(18, 18); ADD_CP(19, 19); ADD_CP(20, 20); ADD_CP(20, 10); ADD_CP(21, 21); ADD_CP(22, 22); ADD_CP(22, 11); ADD_CP(23, 23); ADD_CP(24, 24); ADD_CP(24, 12);
#elif defined(TUNE_RT)   // -DTUNE_RT: the run-time-rounding build (literal path, FAST = 4) at deeper tiles; argv[5] = rounding code (1 down, 3 stochastic)
  const int rt_round = argc > 5 ? atoi(argv[5]) : 1;
#define ADD_RT(U, T, GR) vs.push_back({"bfp-rt U" #U " T" #T " grp" #GR, [=](const void* i, void* o, hipStream_t q) { \
    int g = (int)((n_vec + (int64_t)T * U - 1) / ((int64_t)T * U)); \
    hipLaunchKernelGGL((bfp_rows_kernel<DMXQ_BF16, DMXQ_BF16, kRuntimeRounding, false, U, 3, T, 4, GR>), dim3(g), dim3(T), 0, q, i, o, n_vec, 2, 8, rt_round, 1234ull); }, {}})
  ADD_COPY(16, 7, 512, 0); ADD_COPY(2, 7, 512, 0);
  ADD_RT(2, 512, 2); ADD_RT(4, 512, 4); ADD_RT(8, 512, 4); ADD_RT(12, 512, 4); ADD_RT(16, 512, 2); ADD_RT(16, 512, 4); ADD_RT(16, 512, 8); ADD_RT(8, 256, 4); ADD_RT(16, 256, 4);
#elif defined(TUNE_SMALLFIT)   // -DTUNE_SMALLFIT: 9-16 MiB, the product's 128 x 2 / 512 x 4 against one round of exactly fitting depth
  ADD_COPY(4, 7, 512, 0); ADD_BFPG(2, 3, 128, 0, 2, 2); ADD_BFPG(4, 3, 512, 0, 2, 4); ADD_BFPG(3, 3, 512, 0, 2, 3);
  ADD_BFPG(5, 3, 512, 0, 2, 5); ADD_BFPG(6, 3, 512, 0, 2, 6); ADD_BFPG(7, 3, 512, 0, 2, 7); ADD_BFPG(8, 3, 512, 0, 2, 8);
  ADD_BFPG(10, 3, 256, 0, 2, 10); ADD_BFPG(12, 3, 256, 0, 2, 12); ADD_BFPG(14, 3, 256, 0, 2, 14); ADD_BFPG(16, 3, 256, 0, 2, 16);
#elif defined(TUNE_MIN)   // -DTUNE_MIN: five variants only (compiles in a minute)
  ADD_COPY(16, 7, 512, 0); ADD_BFPG(2, 3, 512, 0, 2, 2); ADD_BFPG(16, 3, 512, 0, 2, 16); ADD_BFPG(17, 3, 512, 0, 2, 17); ADD_BFPG(18, 3, 512, 0, 2, 18);
#else
  if (getenv("TUNE_SET") && std::string(getenv("TUNE_SET")) == "blocks") {
    // round 3: block size 64 / 128 (8 / 16 lanes per block), symmetric vs asymmetric, lane count compile-time vs run-time
    ADD_COPY(16, 7, 512, 0);
    ADD_BFPB(false, 2, 0); ADD_BFPB(false, 8, 0); ADD_BFPB(true, 8, 0); ADD_BFPB(false, 16, 0); ADD_BFPB(true, 16, 0); ADD_BFPB(false, 4, 0);
    ADD_BFPB(false, 8, 8); ADD_BFPB(true, 8, 8); ADD_BFPB(false, 2, 2); ADD_BFPB(true, 2, 0);
    if (getenv("TUNE_WIDE")) { ADD_BFPG(8, 3, 1024, 0, 2, 8); ADD_BFPG(16, 3, 1024, 0, 2, 16); ADD_BFPG(32, 3, 256, 0, 2, 32); ADD_BFPG(16, 3, 512, 0, 2, 8); ADD_BFPG(16, 3, 512, 0, 2, 4); }
  } else if (getenv("TUNE_SET") && std::string(getenv("TUNE_SET")) == "wg") {
    // round 3: workgroup size at 16 (and 8) vectors per lane -- the flat-stream ops preferred 64 .. 256 lanes at the headline size
    ADD_COPY(16, 7, 512, 0); ADD_COPY(16, 7, 256, 0); ADD_COPY(16, 7, 128, 0); ADD_COPY(16, 7, 64, 0);
    ADD_BFPG(16, 3, 512, 0, 2, 16); ADD_BFPG(16, 3, 256, 0, 2, 16); ADD_BFPG(16, 3, 128, 0, 2, 16); ADD_BFPG(16, 3, 64, 0, 2, 16);
    ADD_BFPG(8, 3, 512, 0, 2, 8); ADD_BFPG(8, 3, 256, 0, 2, 8); ADD_BFPG(8, 3, 128, 0, 2, 8); ADD_BFPG(8, 3, 64, 0, 2, 8);
    ADD_BFPG(32, 3, 128, 0, 2, 32); ADD_BFPG(32, 3, 64, 0, 2, 32); ADD_BFPG(4, 3, 256, 0, 2, 4); ADD_BFPG(4, 3, 512, 0, 2, 4); ADD_BFPG(2, 3, 512, 0, 2, 2);
  } else if (getenv("TUNE_SET") && std::string(getenv("TUNE_SET")) == "small") {
    ADD_COPY(2, 7, 512, 0); ADD_COPY(4, 7, 512, 0);
    ADD_BFPG(1, 3, 512, 0, 2, 1); ADD_BFPG(2, 3, 512, 0, 2, 2); ADD_BFPG(3, 3, 512, 0, 2, 3); ADD_BFPG(4, 3, 512, 0, 2, 4); ADD_BFPG(6, 3, 512, 0, 2, 6); ADD_BFPG(8, 3, 512, 0, 2, 8);
    ADD_BFPG(2, 3, 128, 0, 2, 2); ADD_BFPG(1, 3, 256, 0, 2, 1); ADD_BFPG(2, 3, 256, 0, 2, 2); ADD_BFPG(4, 3, 256, 0, 2, 4); ADD_BFPG(8, 3, 256, 0, 2, 8);
  } else if (getenv("TUNE_SET") && std::string(getenv("TUNE_SET")) == "deep") {
    // round 4: ONE round of <= 256 workgroups with a compile-time depth of 17 .. 24 vectors per lane for 32-48 MiB, against 512 x 2
    ADD_COPY(2, 7, 512, 0);
    ADD_BFPG(2, 3, 512, 0, 2, 2); ADD_BFPG(16, 3, 512, 0, 2, 16);
    ADD_BFPG(17, 3, 512, 0, 2, 17); ADD_BFPG(18, 3, 512, 0, 2, 18); ADD_BFPG(18, 3, 512, 0, 2, 9); ADD_BFPG(18, 3, 512, 0, 2, 6);
    ADD_BFPG(19, 3, 512, 0, 2, 1); ADD_BFPG(20, 3, 512, 0, 2, 10); ADD_BFPG(20, 3, 512, 0, 2, 5); ADD_BFPG(20, 3, 512, 0, 2, 4);
    ADD_BFPG(21, 3, 512, 0, 2, 7); ADD_BFPG(22, 3, 512, 0, 2, 11); ADD_BFPG(22, 3, 512, 0, 2, 2); ADD_BFPG(23, 3, 512, 0, 2, 1);
    ADD_BFPG(24, 3, 512, 0, 2, 12); ADD_BFPG(24, 3, 512, 0, 2, 8); ADD_BFPG(24, 3, 512, 0, 2, 6); ADD_BFPG(24, 3, 512, 0, 2, 4);
  } else if (getenv("TUNE_SET") && std::string(getenv("TUNE_SET")) == "fit") {
    // round 4: 16-32 MiB with the depth that fills ONE round of 256 workgroups exactly (U = ceil(n_vec / (256 x 512))), against the product's 128 x 8 / 512 x 16
    ADD_COPY(16, 7, 512, 0);
    ADD_BFPG(8, 3, 128, 0, 2, 8); ADD_BFPG(16, 3, 512, 0, 2, 16);
    ADD_BFPG(9, 3, 512, 0, 2, 9); ADD_BFPG(10, 3, 512, 0, 2, 10); ADD_BFPG(11, 3, 512, 0, 2, 11); ADD_BFPG(12, 3, 512, 0, 2, 12);
    ADD_BFPG(13, 3, 512, 0, 2, 13); ADD_BFPG(14, 3, 512, 0, 2, 14); ADD_BFPG(15, 3, 512, 0, 2, 15);
    ADD_BFPG(5, 3, 512, 0, 2, 5); ADD_BFPG(6, 3, 512, 0, 2, 6); ADD_BFPG(7, 3, 512, 0, 2, 7); ADD_BFPG(8, 3, 512, 0, 2, 8);
  } else if (getenv("TUNE_SET") && std::string(getenv("TUNE_SET")) == "oneround") {
    // round 4: every depth 5 .. 18 as ONE round (with TUNE_LDS=84: one workgroup per CU guaranteed) against the product's plans
    ADD_COPY(16, 7, 512, 0); ADD_BFPG(2, 3, 512, 0, 2, 2); ADD_BFPG(4, 3, 512, 0, 2, 4); ADD_BFPG(8, 3, 128, 0, 2, 8);
    ADD_BFPG(5, 3, 512, 0, 2, 5); ADD_BFPG(6, 3, 512, 0, 2, 6); ADD_BFPG(7, 3, 512, 0, 2, 7); ADD_BFPG(8, 3, 512, 0, 2, 8);
    ADD_BFPG(9, 3, 512, 0, 2, 9); ADD_BFPG(10, 3, 512, 0, 2, 10); ADD_BFPG(11, 3, 512, 0, 2, 11); ADD_BFPG(12, 3, 512, 0, 2, 12);
    ADD_BFPG(13, 3, 512, 0, 2, 13); ADD_BFPG(14, 3, 512, 0, 2, 14); ADD_BFPG(15, 3, 512, 0, 2, 15); ADD_BFPG(16, 3, 512, 0, 2, 16);
    ADD_BFPG(17, 3, 512, 0, 2, 17); ADD_BFPG(18, 3, 512, 0, 2, 18); ADD_BFPG(20, 3, 512, 0, 2, 10); ADD_BFPG(22, 3, 512, 0, 2, 11); ADD_BFPG(24, 3, 512, 0, 2, 12);
  } else if (getenv("TUNE_SET") && std::string(getenv("TUNE_SET")) == "sweep") {
    // tile-plan continuity (round 3): the one-round shapes against the multi-round 512x2 just above the 32 MiB headline size
    ADD_COPY(2, 7, 512, 0); ADD_COPY(16, 7, 512, 0);
    ADD_BFPG(2, 3, 512, 0, 2, 2); ADD_BFPG(4, 3, 512, 0, 2, 4); ADD_BFPG(8, 3, 512, 0, 2, 8); ADD_BFPG(16, 3, 512, 0, 2, 16);
    ADD_BFPG(2, 3, 256, 0, 2, 2); ADD_BFPG(4, 3, 256, 0, 2, 4); ADD_BFPG(8, 3, 256, 0, 2, 8);
    ADD_BFPG(16, 3, 512, 256, 2, 16); ADD_BFPG(8, 3, 512, 512, 2, 8); ADD_BFPG(8, 3, 512, 256, 2, 8);  // persistent: 1 or 2 workgroups per CU looping over tiles
  } else {
  ADD_COPY(16, 7, 512, 0);
  ADD_BFPG(16, 3, 512, 0, 2, 16); ADD_BFPL(16, 3, 512, 0, 2, 16, 2); ADD_BFPG(16, 3, 256, 0, 2, 16); ADD_BFPL(16, 3, 256, 0, 2, 16, 2);
  ADD_BFPG(2, 3, 512, 0, 2, 2); ADD_BFPL(2, 3, 512, 0, 2, 2, 2); ADD_BFPG(16, 3, 512, 0, 2, 16); ADD_BFPL(16, 3, 512, 0, 2, 16, 2);
  }
#endif
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (auto& v : vs) for (int i = 0; i < 10; i++) v.run(in[i % NBUF], out[i % NBUF], st);
  CK(hipStreamSynchronize(st));
  for (int r = 0; r < ROUNDS; r++)
    for (auto& v : vs) {
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < LAUNCHES; i++) v.run(in[i % NBUF], out[i % NBUF], st);
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      v.us.push_back(ms * 1e3f / LAUNCHES);
    }
  printf("%-28s %9s %9s %9s %8s\n", "variant", "min_us", "med_us", "TB/s(med)", "%8TB/s");
  for (auto& v : vs) {
    std::sort(v.us.begin(), v.us.end());
    float med = v.us[v.us.size() / 2], mn = v.us[0];
    double tbs = 4.0 * n / (med * 1e-6) / 1e12;  // n elements of this run
    printf("%-28s %9.2f %9.2f %9.3f %7.1f%%\n", v.name.c_str(), mn, med, tbs, 100.0 * tbs / 8.0);
  }
  return 0;
}
