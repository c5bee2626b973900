// tools/tune_pace.hip — on-GPU A/B harness (not part of the product library): idle issue cycles between the loads of a tile
// (stream.hpp PACE: N x `s_nop 7` after each of a wave's loads) for the library's own ops at their product geometries, 4096 x 4096 bf16.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math -ffp-contract=off -fno-gpu-flush-denormals-to-zero -mllvm -amdgpu-kernarg-preload-count=16 \
//         -Iinclude -DDMXQ_EW_PART=9 tools/tune_pace.hip -o tools/tune_pace
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
struct Variant { std::string name; std::function<void(const void*, void*, hipStream_t)> run; std::vector<float> us; };

int main(int argc, char** argv) {
  const int ROUNDS = argc > 1 ? atoi(argv[1]) : 7;
  const int64_t rows = 4096, C = 4096, n = rows * C, n_vec = n / 8;
  const int NBUF = 20, LAUNCHES = 50;
  std::vector<void*> in(NBUF), out(NBUF);
  std::vector<uint16_t> h(n);
  uint64_t s = 88172645463325252ull;
  for (int64_t i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint16_t)(((s >> 20) & 0x8FFF) | 0x3000) ^ (uint16_t)((s >> 40) & 0x0F00); }
  for (int b = 0; b < NBUF; b++) { CK(hipMalloc(&in[b], n * 2)); CK(hipMalloc(&out[b], n * 2)); CK(hipMemcpy(in[b], h.data(), n * 2, hipMemcpyHostToDevice)); }
  const int64_t G = (rows + 127) / 128;
  std::vector<float> hs(G); std::vector<int64_t> hz(G);
  for (int64_t c = 0; c < G; c++) { hs[c] = 0.002f + 0.0001f * (float)(c % 97); hz[c] = (c % 7) - 3; }
  float* d_scale; int64_t* d_zp;
  CK(hipMalloc(&d_scale, G * 4)); CK(hipMalloc(&d_zp, G * 8));
  CK(hipMemcpy(d_scale, hs.data(), G * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_zp, hz.data(), G * 8, hipMemcpyHostToDevice));
  hipStream_t st; CK(hipStreamCreate(&st));
  const FixedFmt fx{0, 1, DMXQ_ROUND_NEAREST, -128.0f, 127.0f, 0ull};
  const FixedOp<kUniform, true> int8g{fx, make_channel_map(rows, C, 128, n), d_scale, d_zp};
  const FixedOp<kNone, true> int8n{fx, make_channel_map(1, 1, 1, n), nullptr, nullptr};
  const FloatFmt e4{3, 4, 7, 0, 0, DMXQ_ROUND_NEAREST, 0ull};
  const FloatOp<DMXQ_ROUND_NEAREST> e4m3{e4, make_float_fast(3, 4, 7), make_flush_fast(3, 4, 7, 0)};
  const FloatFmt f16f{10, 5, 15, 1, 0, DMXQ_ROUND_NEAREST, 0ull};
  const FloatOp<DMXQ_ROUND_NEAREST> f16c{f16f, make_float_fast(10, 5, 15), make_flush_fast(10, 5, 15, 1)};
  const UnaryOp<DMXQ_UNARY_SILU, DMXQ_BF16, true> silu{1.0f};
  const UnaryOp<DMXQ_UNARY_QUICK_GELU, DMXQ_BF16, true> qgelu{1.702f};
  const GeluOp<true, false> gelu{};
  const MxfpFmt mxf{3, 4, 7, (float)ldexp(1.0, 8), make_float_fast(3, 4, 7), 8, 1, 1};
  const BlockOp<MxfpFmt, MxfpBlock> mxfp{mxf, 4};
  const SbfpFmt sbf{4, 1, -7.0f, 7.0f, 7.0f, 4, 4, 7, 0};
  const BlockOp<SbfpFmt, SbfpBlock> sbfp{sbf, 2};
  std::vector<Variant> vs;
#define ADD(NAME, OPV, T, U, P) vs.push_back({std::string(NAME) + " " #T "x" #U " pace " #P, [=](const void* i, void* o, hipStream_t q) { \
    hipLaunchKernelGGL((stream_kernel<DMXQ_BF16, DMXQ_BF16, U, T, std::remove_const_t<decltype(OPV)>, false, 16, P>), dim3((unsigned)((n_vec + (int64_t)T * U - 1) / ((int64_t)T * U))), dim3(T), 0, q, i, o, n, OPV); }, {}})
#define SWEEP(NAME, OPV, T, U) ADD(NAME, OPV, T, U, 0); ADD(NAME, OPV, T, U, 4); ADD(NAME, OPV, T, U, 6); ADD(NAME, OPV, T, U, 8); ADD(NAME, OPV, T, U, 12); ADD(NAME, OPV, T, U, 16); ADD(NAME, OPV, T, U, 24); ADD(NAME, OPV, T, U, 32)
  const char* set = getenv("TUNE_SET");
  const std::string ss = set ? set : "";
  SWEEP("int8g", int8g, 128, 16); SWEEP("int8g", int8g, 256, 8); SWEEP("int8n", int8n, 256, 8);
  SWEEP("e4m3 ", e4m3, 256, 8); SWEEP("f16c ", f16c, 256, 8); SWEEP("silu ", silu, 256, 8); SWEEP("qgelu", qgelu, 256, 8); SWEEP("gelu ", gelu, 256, 8); SWEEP("gelu ", gelu, 512, 4);
  SWEEP("mxfp8", mxfp, 256, 8); SWEEP("sbfp ", sbfp, 128, 8); SWEEP("sbfp ", sbfp, 256, 8);
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
    printf("%-28s min %6.2f med %6.2f  %5.1f%%\n", v.name.c_str(), v.us[0], med, 100.0 * 4.0 * n / (med * 1e-6) / 8e12); }
  return 0;
}
