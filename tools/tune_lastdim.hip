// tools/tune_lastdim.hip — on-GPU A/B harness for the per-channel-along-the-contiguous-dim kernels (not part of the product library):
// lastdim_kernel<FixedOp<kLast, true>> (INT8 per channel) and lastdim_kernel<ScaleOp<., kLast>> (SmoothQuant's x / s, x * s) over
// workgroup size, rows in flight per lane and rows per workgroup, against the flat-stream copy of the same box.  Includes the product
// source itself with its entry points compiled out, so the kernels measured are the library's.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math -ffp-contract=off -fno-gpu-flush-denormals-to-zero -Iinclude \
//         -DDMXQ_EW_PART=9 tools/tune_lastdim.hip -o tools/tune_lastdim
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#include "../dmx-compressor_amd/csrc/elementwise.hip"

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

int main(int argc, char** argv) {
  const int ROUNDS = argc > 1 ? atoi(argv[1]) : 7;
  const int64_t rows = argc > 2 ? atoll(argv[2]) : 4096, C = argc > 3 ? atoll(argv[3]) : 4096, n = rows * C, n_vec = n / 8;
  const int NBUF = (int)std::max<int64_t>(2, std::min<int64_t>(48, (int64_t)1280 * 1024 * 1024 / (n * 4))), LAUNCHES = 50;
  std::vector<void*> in(NBUF), out(NBUF);
  std::vector<uint16_t> h(n);
  uint64_t s = 88172645463325252ull;
  for (int64_t i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint16_t)(((s >> 20) & 0x8FFF) | 0x3000) ^ (uint16_t)((s >> 40) & 0x0F00); }
  for (int b = 0; b < NBUF; b++) { CK(hipMalloc(&in[b], n * 2)); CK(hipMalloc(&out[b], n * 2)); CK(hipMemcpy(in[b], h.data(), n * 2, hipMemcpyHostToDevice)); }
  std::vector<float> hs(C);
  std::vector<int64_t> hz(C);
  for (int64_t c = 0; c < C; c++) { hs[c] = 0.002f + 0.0001f * (float)(c % 97); hz[c] = (c % 7) - 3; }
  float* d_scale; int64_t* d_zp;
  CK(hipMalloc(&d_scale, C * 4)); CK(hipMalloc(&d_zp, C * 8));
  CK(hipMemcpy(d_scale, hs.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_zp, hz.data(), C * 8, hipMemcpyHostToDevice));
  printf("# rows %lld C %lld nbuf %d\n", (long long)rows, (long long)C, NBUF);
  hipStream_t st; CK(hipStreamCreate(&st));
  const FixedFmt ff{0, 1, DMXQ_ROUND_NEAREST, -128.0f, 127.0f, 0ull};  // INT8: fraction 0, clamped, nearest
  const ChannelMap cm = make_channel_map(C, 1, 1, n);
  const FixedOp<kLast, true> fop{ff, cm, d_scale, d_zp};
  const ScaleOp<true, kLast> dop{cm, d_scale};
  const ScaleOp<false, kLast> mop{cm, d_scale};
  const int cv = (int)(C / 8);
  std::vector<Variant> vs;
#define ADD_COPY(U, T) vs.push_back({"copy " #T "x" #U, [=](const void* i, void* o, hipStream_t q) { \
    hipLaunchKernelGGL((copy_tiles<U, T>), dim3((unsigned)((n_vec + (int64_t)T * U - 1) / ((int64_t)T * U))), dim3(T), 0, q, i, o, n_vec); }, {}})
  // T threads, R rows per lane (rows per workgroup = rpp * R), one pass per workgroup
#define ADD_LD(NAME, OPV, T, R) vs.push_back({std::string(NAME) + " T" #T " R" #R, [=](const void* i, void* o, hipStream_t q) { \
    const int lpr = cv < T ? cv : T, rpp = T / lpr, strips = (cv + lpr - 1) / lpr; \
    int64_t gx = (rows + (int64_t)rpp * R - 1) / ((int64_t)rpp * R); \
    hipLaunchKernelGGL((lastdim_kernel<DMXQ_BF16, DMXQ_BF16, decltype(OPV), T, R>), dim3((unsigned)gx, (unsigned)strips), dim3(T), 0, q, i, o, rows, C, cv, make_fastdiv_u32(lpr), rpp, OPV); }, {}})
#define ADD_ALL(NAME, OPV) \
  ADD_LD(NAME, OPV, 128, 16); ADD_LD(NAME, OPV, 128, 32); ADD_LD(NAME, OPV, 256, 4); ADD_LD(NAME, OPV, 256, 8); ADD_LD(NAME, OPV, 256, 16); ADD_LD(NAME, OPV, 256, 32); \
  ADD_LD(NAME, OPV, 512, 4); ADD_LD(NAME, OPV, 512, 8); ADD_LD(NAME, OPV, 512, 16); ADD_LD(NAME, OPV, 512, 32); ADD_LD(NAME, OPV, 1024, 8); ADD_LD(NAME, OPV, 1024, 16);
  ADD_COPY(16, 512); ADD_COPY(8, 256); ADD_COPY(2, 512);
  const char* set = getenv("TUNE_SET");
  if (!set || std::string(set) == "fixed") { ADD_ALL("int8", fop); }
  if (!set || std::string(set) == "scale") { ADD_ALL("div ", dop); ADD_ALL("mul ", mop); }
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
