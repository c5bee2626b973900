// tools/tune_rows.hip — on-GPU A/B harness for the register-resident row kernels of csrc/approx.hip (not part of the product library):
// LayerNorm rows of 768 (wave kernel), RMSNorm rows of 4096 (workgroup-per-row kernel), softmax rows of 1500 (wave kernel, 8-byte
// lane-vectors), plain and with the fused FLOAT16 casts, over (a) rows per wave / workgroup iteration and (b) the grid: persistent
// (as many workgroups as are resident, looping over rows -- the product's choice in rounds 1-3) against one pass per workgroup.
// Includes the product source with its entry points compiled out, so the kernels measured are the library's.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math -ffp-contract=off -fno-gpu-flush-denormals-to-zero -Iinclude \
//         -DDMXQ_EW_PART=9 tools/tune_rows.hip -o tools/tune_rows
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#include "../dmx-compressor_amd/csrc/approx.hip"

using namespace dmxq;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <int UNROLL, int THREADS>
__global__ __launch_bounds__(THREADS) void copy_tiles(const void* __restrict__ in, void* __restrict__ out, int64_t n_vec) {
  const int64_t v = (int64_t)blockIdx.x * THREADS * UNROLL + threadIdx.x;
  u32x4 raw[UNROLL];
#pragma unroll
  for (int u = 0; u < UNROLL; u++) if (v + u * THREADS < n_vec) raw[u] = load_raw16<true>(in, (v + u * THREADS) * 16);
#pragma unroll
  for (int u = 0; u < UNROLL; u++) if (v + u * THREADS < n_vec) __builtin_nontemporal_store(raw[u], (u32x4*)((char*)out + (v + u * THREADS) * 16));
}

struct Variant { std::string name; std::function<void(const void*, void*, hipStream_t)> run; std::vector<float> us; };

template <typename K>
static int resident_cap(K kernel) {
  int per_cu = 2;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kThreads, 0) != hipSuccess || per_cu < 1) per_cu = 2;
  return 256 * per_cu;
}

template <bool C> static RowCastArg<C> pick_cast(const RowCast& rc) { if constexpr (C) return rc; else return NoRowCast{}; }

int main(int argc, char** argv) {
  const int ROUNDS = argc > 1 ? atoi(argv[1]) : 7;
  const int64_t n = argc > 2 ? atoll(argv[2]) * 4096 : (int64_t)4096 * 4096;  // elements (bf16): every shape uses about this many
  const int NBUF = (int)std::max<int64_t>(2, std::min<int64_t>(48, (int64_t)1280 * 1024 * 1024 / (n * 4))), LAUNCHES = 50;
  std::vector<void*> in(NBUF), out(NBUF);
  std::vector<uint16_t> h(n);
  uint64_t s = 88172645463325252ull;
  for (int64_t i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint16_t)(((s >> 20) & 0x83FF) | 0x3C00) ^ (uint16_t)((s >> 40) & 0x0300); }
  for (int b = 0; b < NBUF; b++) { CK(hipMalloc(&in[b], n * 2)); CK(hipMalloc(&out[b], n * 2)); CK(hipMemcpy(in[b], h.data(), n * 2, hipMemcpyHostToDevice)); }
  void *d_w, *d_b;
  CK(hipMalloc(&d_w, 8192 * 2)); CK(hipMalloc(&d_b, 8192 * 2));
  CK(hipMemcpy(d_w, h.data(), 8192 * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(d_b, h.data() + 8192, 8192 * 2, hipMemcpyHostToDevice));
  hipStream_t st; CK(hipStreamCreate(&st));
  const dmxq_float_fmt f16{10, 5, 15, 1};
  RowCast rc;
  if (!rowcast_of(DMXQ_BF16, &f16, &f16, &rc)) { printf("rowcast_of failed\n"); return 1; }
  const int64_t n_vec = n / 8;
  const int64_t rows768 = n / 768, rows4096 = n / 4096, rows1500 = n / 1500;
  printf("# n %lld nbuf %d: layernorm %lld x 768, rmsnorm %lld x 4096, softmax %lld x 1500\n", (long long)n, NBUF, (long long)rows768, (long long)rows4096, (long long)rows1500);
  std::vector<Variant> vs;
#define ADD_COPY(T, U) vs.push_back({"copy   " #T "x" #U, [=](const void* i, void* o, hipStream_t q) { \
    hipLaunchKernelGGL((copy_tiles<U, T>), dim3((unsigned)((n_vec + (int64_t)T * U - 1) / ((int64_t)T * U))), dim3(T), 0, q, i, o, n_vec); }, {}})
  // GRID: 0 = persistent (resident cap), 1 = one pass per workgroup, 2 = two passes
#define GRID_OF(KERN, WANT, MODE) ((MODE) == 0 ? (unsigned)std::min<int64_t>((WANT), resident_cap(KERN)) : (unsigned)(((WANT) + (MODE) - 1) / (MODE)))
#define ADD_LN(CASTV, RPW, MODE) { auto k = layernorm_wave_kernel<DMXQ_BF16, 8, 3, 32, false, CASTV, RPW>; const int64_t want = (rows768 + 4 * RPW * 2 - 1) / (4 * RPW * 2); \
    const unsigned g = GRID_OF(k, want, MODE); const RowCastArg<CASTV> a = pick_cast<CASTV>(rc); \
    vs.push_back({std::string(CASTV ? "ln768_m" : "ln768  ") + " rpw" #RPW " grid" #MODE, [=](const void* i, void* o, hipStream_t q) { \
      hipLaunchKernelGGL(k, dim3(g), dim3(kThreads), 0, q, i, o, rows768, (int64_t)768, (const void*)d_w, (const void*)d_b, 1e-5f, a); }, {}}); }
#define ADD_RMS(CASTV, RPW, MODE) { auto k = layernorm_block_kernel<DMXQ_BF16, 8, 2, true, CASTV, RPW>; const int64_t want = (rows4096 + RPW - 1) / RPW; \
    const unsigned g = GRID_OF(k, want, MODE); const RowCastArg<CASTV> a = pick_cast<CASTV>(rc); \
    vs.push_back({std::string(CASTV ? "rms4k_m" : "rms4k  ") + " rpw" #RPW " grid" #MODE, [=](const void* i, void* o, hipStream_t q) { \
      hipLaunchKernelGGL(k, dim3(g), dim3(kThreads), 0, q, i, o, rows4096, (int64_t)4096, (const void*)d_w, (const void*)nullptr, 1e-5f, a); }, {}}); }
#define ADD_LNH(CASTV, RPW, MODE, H) { auto k = layernorm_wave_kernel<DMXQ_BF16, 8, 3, 32, false, CASTV, RPW, H>; const int64_t want = (rows768 + 4 * RPW * 2 - 1) / (4 * RPW * 2); \
    const unsigned g = GRID_OF(k, want, MODE); const RowCastArg<CASTV> a = pick_cast<CASTV>(rc); \
    vs.push_back({std::string(CASTV ? "ln768_m" : "ln768  ") + " rpw" #RPW " grid" #MODE " hoist" #H, [=](const void* i, void* o, hipStream_t q) { \
      hipLaunchKernelGGL(k, dim3(g), dim3(kThreads), 0, q, i, o, rows768, (int64_t)768, (const void*)d_w, (const void*)d_b, 1e-5f, a); }, {}}); }
#define ADD_LNP(CASTV, RPW, MODE, H, PF) { auto k = layernorm_wave_kernel<DMXQ_BF16, 8, 3, 32, false, CASTV, RPW, H, PF>; const int64_t want = (rows768 + 4 * RPW * 2 - 1) / (4 * RPW * 2); \
    const unsigned g = GRID_OF(k, want, MODE); const RowCastArg<CASTV> a = pick_cast<CASTV>(rc); \
    vs.push_back({std::string(CASTV ? "ln768_m" : "ln768  ") + " rpw" #RPW " grid" #MODE " hoist" #H " early" #PF, [=](const void* i, void* o, hipStream_t q) { \
      hipLaunchKernelGGL(k, dim3(g), dim3(kThreads), 0, q, i, o, rows768, (int64_t)768, (const void*)d_w, (const void*)d_b, 1e-5f, a); }, {}}); }
#define ADD_LNB(CASTV, RPW, MODE) { auto k = layernorm_block_kernel<DMXQ_BF16, 8, 2, false, CASTV, RPW>; const int64_t want = (rows4096 + RPW - 1) / RPW; \
    const unsigned g = GRID_OF(k, want, MODE); const RowCastArg<CASTV> a = pick_cast<CASTV>(rc); \
    vs.push_back({std::string(CASTV ? "ln4k_m " : "ln4k   ") + " rpw" #RPW " grid" #MODE, [=](const void* i, void* o, hipStream_t q) { \
      hipLaunchKernelGGL(k, dim3(g), dim3(kThreads), 0, q, i, o, rows4096, (int64_t)4096, (const void*)d_w, (const void*)d_b, 1e-5f, a); }, {}}); }
#define ADD_RMSW(CASTV, RPW, MODE) { auto k = layernorm_wave_kernel<DMXQ_BF16, 8, 8, 64, true, CASTV, RPW>; const int64_t want = (rows4096 + 4 * RPW - 1) / (4 * RPW); \
    const unsigned g = GRID_OF(k, want, MODE); const RowCastArg<CASTV> a = pick_cast<CASTV>(rc); \
    vs.push_back({std::string(CASTV ? "rms4kw_m" : "rms4kw ") + " rpw" #RPW " grid" #MODE, [=](const void* i, void* o, hipStream_t q) { \
      hipLaunchKernelGGL(k, dim3(g), dim3(kThreads), 0, q, i, o, rows4096, (int64_t)4096, (const void*)d_w, (const void*)nullptr, 1e-5f, a); }, {}}); }
#define ADD_SM(CASTV, RPW, MODE) { auto k = softmax_wave_kernel<DMXQ_BF16, 4, 6, 64, false, CASTV, false, RPW>; const int64_t want = (rows1500 + 4 * RPW - 1) / (4 * RPW); \
    const unsigned g = GRID_OF(k, want, MODE); const RowCastArg<CASTV> a = pick_cast<CASTV>(rc); \
    vs.push_back({std::string(CASTV ? "sm1500_m" : "sm1500 ") + " rpw" #RPW " grid" #MODE, [=](const void* i, void* o, hipStream_t q) { \
      hipLaunchKernelGGL(k, dim3(g), dim3(kThreads), 0, q, i, o, rows1500, (int64_t)1500, -INFINITY, a); }, {}}); }
#define ADD_GRIDS(M, C, R) M(C, R, 0) M(C, R, 1) M(C, R, 2)
  ADD_COPY(512, 16); ADD_COPY(256, 16); ADD_COPY(512, 2);
#ifdef TUNE_ROWS_SHORT
  ADD_GRIDS(ADD_LN, false, 1) ADD_GRIDS(ADD_LN, false, 2) ADD_GRIDS(ADD_LN, false, 4) ADD_GRIDS(ADD_LN, false, 8)
  ADD_GRIDS(ADD_LN, true, 1) ADD_GRIDS(ADD_LN, true, 2) ADD_GRIDS(ADD_LN, true, 4) ADD_GRIDS(ADD_LN, true, 8)
  ADD_GRIDS(ADD_RMS, true, 1) ADD_GRIDS(ADD_RMS, true, 2) ADD_GRIDS(ADD_RMS, true, 4)
  ADD_GRIDS(ADD_SM, false, 1) ADD_GRIDS(ADD_SM, false, 2) ADD_GRIDS(ADD_SM, false, 4)
  ADD_GRIDS(ADD_SM, true, 1) ADD_GRIDS(ADD_SM, true, 2) ADD_GRIDS(ADD_SM, true, 4)
#else
  ADD_LNP(false, 1, 0, 0, 0) ADD_LNP(false, 1, 0, 0, 1) ADD_LNP(false, 2, 0, 0, 0) ADD_LNP(false, 2, 0, 0, 1) ADD_LNP(false, 1, 0, 1, 0) ADD_LNP(false, 1, 0, 1, 1) ADD_LNP(false, 2, 0, 1, 1) ADD_LNP(false, 4, 0, 0, 1)
  ADD_LNP(true, 1, 0, 0, 0) ADD_LNP(true, 1, 0, 0, 1) ADD_LNP(true, 1, 0, 1, 1) ADD_LNP(true, 2, 0, 0, 1)
  ADD_LNH(false, 1, 0, 1) ADD_LNH(false, 1, 1, 1) ADD_LNH(false, 2, 0, 1) ADD_LNH(false, 2, 1, 1) ADD_LNH(false, 1, 0, 2) ADD_LNH(false, 1, 1, 2) ADD_LNH(false, 2, 0, 2) ADD_LNH(false, 2, 1, 2)
  ADD_LNH(true, 1, 0, 1) ADD_LNH(true, 1, 1, 1) ADD_LNH(true, 2, 0, 1) ADD_LNH(true, 1, 0, 2) ADD_LNH(true, 1, 1, 2)
  ADD_GRIDS(ADD_LN, false, 1) ADD_GRIDS(ADD_LN, false, 2) ADD_GRIDS(ADD_LN, false, 4) ADD_GRIDS(ADD_LN, false, 8)
  ADD_GRIDS(ADD_LN, true, 1) ADD_GRIDS(ADD_LN, true, 2) ADD_GRIDS(ADD_LN, true, 4)
  ADD_GRIDS(ADD_RMS, false, 1) ADD_GRIDS(ADD_RMS, false, 2) ADD_GRIDS(ADD_RMS, false, 4) ADD_GRIDS(ADD_RMS, false, 8)
  ADD_GRIDS(ADD_RMS, true, 1) ADD_GRIDS(ADD_RMS, true, 2) ADD_GRIDS(ADD_RMS, true, 4) ADD_GRIDS(ADD_RMS, true, 8)
  ADD_GRIDS(ADD_LNB, false, 2) ADD_GRIDS(ADD_LNB, false, 4) ADD_GRIDS(ADD_LNB, false, 8) ADD_GRIDS(ADD_LNB, true, 2) ADD_GRIDS(ADD_LNB, true, 8)
  ADD_GRIDS(ADD_SM, false, 1) ADD_GRIDS(ADD_SM, false, 2) ADD_GRIDS(ADD_SM, false, 4)
  ADD_GRIDS(ADD_SM, true, 1) ADD_GRIDS(ADD_SM, true, 2)
#endif
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (auto& v : vs) for (int i = 0; i < 10; i++) v.run(in[i % NBUF], out[i % NBUF], st);
  CK(hipStreamSynchronize(st));
  CK(hipGetLastError());
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
    double tbs = 4.0 * n / (med * 1e-6) / 1e12;
    printf("%-28s %9.2f %9.2f %9.3f %7.1f%%\n", v.name.c_str(), mn, med, tbs, 100.0 * tbs / 8.0);
  }
  return 0;
}
