// tools/tune_stream.hip — on-GPU A/B harness for the flat-stream skeleton (csrc/stream.hpp; not part of the product library): the
// library's own per-element ops (INT8 with a per-group scale, minifloat cast, SiLU, QuickGELU, erf GELU, the fused GELU module)
// over tile geometries THREADS x UNROLL, against plain copies on the same box.  Includes the product sources with their entry points
// compiled out (elementwise.hip) or unused (act_cast.hip), so the kernels measured are the library's.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math -ffp-contract=off -fno-gpu-flush-denormals-to-zero -Iinclude \
//         -DDMXQ_EW_PART=9 tools/tune_stream.hip -o tools/tune_stream
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#include "../dmx-compressor_amd/csrc/elementwise.hip"
#include "../dmx-compressor_amd/csrc/act_cast.hip"
#include "../dmx-compressor_amd/csrc/blockfmt.hip"

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
  const int64_t G = (rows + 127) / 128;
  std::vector<float> hs(G);
  std::vector<int64_t> hz(G);
  for (int64_t c = 0; c < G; c++) { hs[c] = 0.002f + 0.0001f * (float)(c % 97); hz[c] = (c % 7) - 3; }
  float* d_scale; int64_t* d_zp;
  CK(hipMalloc(&d_scale, G * 4)); CK(hipMalloc(&d_zp, G * 8));
  CK(hipMemcpy(d_scale, hs.data(), G * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_zp, hz.data(), G * 8, hipMemcpyHostToDevice));
  int64_t* d_zp0; CK(hipMalloc(&d_zp0, G * 8)); CK(hipMemset(d_zp0, 0, G * 8));  // symmetric schemes: every zero point 0
  printf("# rows %lld C %lld nbuf %d\n", (long long)rows, (long long)C, NBUF);
  hipStream_t st; CK(hipStreamCreate(&st));
  const FixedFmt fx{0, 1, DMXQ_ROUND_NEAREST, -128.0f, 127.0f, 0ull};  // INT8: fraction 0, clamped, nearest
  const FixedOp<kUniform, true> int8g{fx, make_channel_map(rows, C, 128, n), d_scale, d_zp};   // [1, rows, C], groups of 128 rows
  const FixedOp<kUniform, true> int8g0{fx, make_channel_map(rows, C, 128, n), d_scale, d_zp0};
  const FixedOp<kNone, true> int8n{fx, make_channel_map(1, 1, 1, n), nullptr, nullptr};
  const FloatFmt e4{3, 4, 7, 0, 0, DMXQ_ROUND_NEAREST, 0ull};
  const FloatOp<DMXQ_ROUND_NEAREST> e4m3{e4, make_float_fast(3, 4, 7), make_flush_fast(3, 4, 7, 0)};
  const UnaryOp<DMXQ_UNARY_SILU, DMXQ_BF16, true> silu{1.0f};
  const UnaryOp<DMXQ_UNARY_QUICK_GELU, DMXQ_BF16, true> qgelu{1.702f};
  const GeluOp<true, false> gelu{};
  const dmxq_float_fmt f16{10, 5, 15, 1};
  ActCasts ac{};
  if (!range16_of(&f16, DMXQ_BF16, &ac.ri) || !range16_of(&f16, DMXQ_BF16, &ac.ro)) { printf("range16_of failed\n"); return 1; }
  const CastedOp<GeluOp<true, false>, DMXQ_BF16> gelum{gelu, ac.ri, ac.ro, ac.gi, ac.go};
  const CastedOp<UnaryOp<DMXQ_UNARY_SILU, DMXQ_BF16, true, true>, DMXQ_BF16> silum{{1.0f}, ac.ri, ac.ro, ac.gi, ac.go};
  // composite block formats (blockfmt.hip): MXFP8[E4M3]{32} (4 lanes per block), SBFP12_16 (2 lanes per block)
  const MxfpFmt mxf{3, 4, 7, (float)ldexp(1.0, 8), make_float_fast(3, 4, 7), 8, 1, 1};
  const BlockOp<MxfpFmt, MxfpBlock> mxfp{mxf, 4};
  const SbfpFmt sbf{4, 1, -7.0f, 7.0f, 7.0f, 4, 4, 7, 0};
  const BlockOp<SbfpFmt, SbfpBlock> sbfp{sbf, 2};
  // float32 tensors (the same bytes viewed as n / 2 floats): FLOAT16 cast, the fused GELU / SiLU modules with FLOAT16 casts
  const FloatFmt h16{10, 5, 15, 1, 0, DMXQ_ROUND_NEAREST, 0ull};
  const FloatOp<DMXQ_ROUND_NEAREST> f16cast{h16, make_float_fast(10, 5, 15), make_flush_fast(10, 5, 15, 1)};
  ActCasts ac32{};
  if (!castg_of(&f16, &ac32.gi) || !castg_of(&f16, &ac32.go)) { printf("castg_of failed\n"); return 1; }
  const CastedOp<GeluOp<true, false>, DMXQ_F32> gelum32{gelu, ac32.ri, ac32.ro, ac32.gi, ac32.go};
  const CastedOp<UnaryOp<DMXQ_UNARY_SILU, DMXQ_F32, true>, DMXQ_F32> silum32{{1.0f}, ac32.ri, ac32.ro, ac32.gi, ac32.go};
  std::vector<Variant> vs;
#define ADD_ST32(NAME, OPV, T, U) vs.push_back({std::string(NAME) + " " #T "x" #U, [=](const void* i, void* o, hipStream_t q) { \
    hipLaunchKernelGGL((stream_kernel<DMXQ_F32, DMXQ_F32, U, T, std::remove_const_t<decltype(OPV)>>), dim3((unsigned)((n_vec + (int64_t)T * U - 1) / ((int64_t)T * U))), dim3(T), 0, q, i, o, n / 2, OPV); }, {}})
#define ADD_ALL32(NAME, OPV) \
  ADD_ST32(NAME, OPV, 64, 8); ADD_ST32(NAME, OPV, 64, 16); ADD_ST32(NAME, OPV, 128, 4); ADD_ST32(NAME, OPV, 128, 8); ADD_ST32(NAME, OPV, 128, 16); \
  ADD_ST32(NAME, OPV, 256, 2); ADD_ST32(NAME, OPV, 256, 4); ADD_ST32(NAME, OPV, 256, 8); ADD_ST32(NAME, OPV, 256, 16); \
  ADD_ST32(NAME, OPV, 512, 2); ADD_ST32(NAME, OPV, 512, 4); ADD_ST32(NAME, OPV, 512, 8); ADD_ST32(NAME, OPV, 512, 16);
#define ADD_COPY(T, U) vs.push_back({"copy   " #T "x" #U, [=](const void* i, void* o, hipStream_t q) { \
    hipLaunchKernelGGL((copy_tiles<U, T>), dim3((unsigned)((n_vec + (int64_t)T * U - 1) / ((int64_t)T * U))), dim3(T), 0, q, i, o, n_vec); }, {}})
#define ADD_ST(NAME, OPV, T, U) vs.push_back({std::string(NAME) + " " #T "x" #U, [=](const void* i, void* o, hipStream_t q) { \
    hipLaunchKernelGGL((stream_kernel<DMXQ_BF16, DMXQ_BF16, U, T, std::remove_const_t<decltype(OPV)>>), dim3((unsigned)((n_vec + (int64_t)T * U - 1) / ((int64_t)T * U))), dim3(T), 0, q, i, o, n, OPV); }, {}})
#define ADD_ALL(NAME, OPV) \
  ADD_ST(NAME, OPV, 64, 16); ADD_ST(NAME, OPV, 64, 32); ADD_ST(NAME, OPV, 128, 8); ADD_ST(NAME, OPV, 128, 16); ADD_ST(NAME, OPV, 128, 32); \
  ADD_ST(NAME, OPV, 256, 2); ADD_ST(NAME, OPV, 256, 4); ADD_ST(NAME, OPV, 256, 8); ADD_ST(NAME, OPV, 256, 16); \
  ADD_ST(NAME, OPV, 512, 2); ADD_ST(NAME, OPV, 512, 4); ADD_ST(NAME, OPV, 512, 8); ADD_ST(NAME, OPV, 512, 16); ADD_ST(NAME, OPV, 1024, 4); ADD_ST(NAME, OPV, 1024, 8);
#define ADD_SMALL(NAME, OPV) \
  ADD_ST(NAME, OPV, 64, 2); ADD_ST(NAME, OPV, 64, 4); ADD_ST(NAME, OPV, 128, 1); ADD_ST(NAME, OPV, 128, 2); ADD_ST(NAME, OPV, 128, 4); ADD_ST(NAME, OPV, 256, 1); ADD_ST(NAME, OPV, 512, 1);
#define ADD_HEAVY(NAME, OPV) \
  ADD_ST(NAME, OPV, 64, 8); ADD_ST(NAME, OPV, 128, 4); ADD_ST(NAME, OPV, 128, 8); \
  ADD_ST(NAME, OPV, 256, 2); ADD_ST(NAME, OPV, 256, 4); ADD_ST(NAME, OPV, 256, 8); \
  ADD_ST(NAME, OPV, 512, 2); ADD_ST(NAME, OPV, 512, 4); ADD_ST(NAME, OPV, 512, 8); ADD_ST(NAME, OPV, 1024, 2); ADD_ST(NAME, OPV, 1024, 4);
  ADD_COPY(512, 16); ADD_COPY(256, 16); ADD_COPY(128, 16); ADD_COPY(128, 32); ADD_COPY(256, 8); ADD_COPY(512, 2);
  const char* set = getenv("TUNE_SET");
  const std::string ss = set ? set : "";
  if (ss == "small") { ADD_COPY(256, 1); ADD_COPY(256, 2); ADD_COPY(128, 2);
    ADD_SMALL("int8g ", int8g); ADD_HEAVY("int8g ", int8g); ADD_SMALL("silu  ", silu); ADD_HEAVY("silu  ", silu); ADD_SMALL("gelu_m", gelum); ADD_HEAVY("gelu_m", gelum); }
  if (ss == "block") { ADD_HEAVY("mxfp8 ", mxfp); ADD_HEAVY("sbfp  ", sbfp); }
  if (ss == "f32") { ADD_ALL32("f16c32", f16cast); ADD_ALL32("gelm32", gelum32); ADD_ALL32("silm32", silum32); }
  if (ss == "int8g") { ADD_ST("int8g ", int8g, 64, 16); ADD_ST("int8g ", int8g, 128, 8); ADD_ST("int8g ", int8g, 128, 16); ADD_ST("int8g ", int8g, 256, 8); ADD_ST("int8g ", int8g, 256, 16); ADD_ST("int8g ", int8g, 512, 16); ADD_ST("int8g ", int8g, 256, 2); ADD_ST("int8g ", int8g, 512, 2);
    ADD_ST("int8g0", int8g0, 64, 16); ADD_ST("int8g0", int8g0, 128, 8); ADD_ST("int8g0", int8g0, 128, 16); ADD_ST("int8g0", int8g0, 256, 8); ADD_ST("int8g0", int8g0, 256, 16); ADD_ST("int8g0", int8g0, 512, 16); ADD_ST("int8g0", int8g0, 256, 2); ADD_ST("int8g0", int8g0, 512, 2);
    ADD_ST("int8n ", int8n, 64, 16); ADD_ST("int8n ", int8n, 128, 16); }
  if (ss.empty() || ss == "fixed") { ADD_ALL("int8g ", int8g); ADD_ALL("int8n ", int8n); ADD_HEAVY("e4m3  ", e4m3); }
  if (ss.empty() || ss == "unary") { ADD_ALL("silu  ", silu); ADD_HEAVY("qgelu ", qgelu); ADD_ALL("gelu  ", gelu); ADD_ALL("gelu_m", gelum); ADD_ALL("silu_m", silum); }
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
