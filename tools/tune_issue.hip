// tools/tune_issue.hip — on-GPU A/B harness (not part of the product library) behind profiles/r05_tune_pace.txt sections 1, 2, 4, 11:
// what bounds a one-round launch besides its bytes?  y = x * s on 4096 x 4096 bf16 through (a) `ldx`, a 60-line kernel with lastdim_kernel's
// geometry (T lanes, R rows per lane, CG column groups, 2-D or 1-D grid, table or constants, whole-tile wait, BURST = addresses first and
// N - 1 x 8 idle cycles between the loads), (b) `flat` / `flatw`, contiguous tiles in the stream skeleton's and in a wave-contiguous layout,
// with paced loads, (c) the product's lastdim_kernel itself (x * s, x / s, INT8 per channel) over pace.  The variant list in main() is the
// LAST experiment run (paced product kernels); earlier ones are in the git history of this file's sections.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math -ffp-contract=off -fno-gpu-flush-denormals-to-zero -mllvm -amdgpu-kernarg-preload-count=16 \
//         -Iinclude -DDMXQ_EW_PART=9 tools/tune_issue.hip -o tools/tune_issue
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

struct Pad { int64_t a[12]; };
template <int T, int R, int CG, int ORDER, int TABLE, int WAIT, int BURST = 0>
__global__ __launch_bounds__(T) void ldx(const void* __restrict__ in, void* __restrict__ out, int rows, int cv, const float* __restrict__ sc) {
  const int strips = cv / (T * CG);
  int bx, by;
  if (ORDER == 0) { bx = blockIdx.x; by = blockIdx.y; }
  else { by = blockIdx.x % strips; bx = blockIdx.x / strips; }
  const int t = threadIdx.x;
  const int r0 = bx * R;
  float s[CG][8];
#pragma unroll
  for (int g = 0; g < CG; g++) {
    const int cb = (by * CG + g) * T + t;
    if (TABLE) {
      const f32x4 a = *(const f32x4*)(sc + (int64_t)cb * 8), b = *(const f32x4*)(sc + (int64_t)cb * 8 + 4);
      s[g][0] = a.x; s[g][1] = a.y; s[g][2] = a.z; s[g][3] = a.w; s[g][4] = b.x; s[g][5] = b.y; s[g][6] = b.z; s[g][7] = b.w;
    } else {
#pragma unroll
      for (int k = 0; k < 8; k++) s[g][k] = 0.5f + 0.001f * (float)((cb * 8 + k) & 127);
    }
  }
  u32x4 raw[R][CG];
  const char* src = (const char*)in + ((int64_t)r0 * cv + (int64_t)by * CG * T) * 16;
  if (BURST) {
    const char* rp[R];
#pragma unroll
    for (int j = 0; j < R; j++) { rp[j] = src + ((int64_t)j * cv) * 16 + (uint32_t)t * 16u; asm volatile("" : "+v"(rp[j])); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < R; j++) { raw[j][0] = load_raw16<true, uint32_t>(rp[j], 0u);
      if (BURST > 1) { __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < BURST - 1; q++) asm volatile("s_nop 7"); __builtin_amdgcn_sched_barrier(0); } }
  } else {
#pragma unroll
  for (int j = 0; j < R; j++)
#pragma unroll
    for (int g = 0; g < CG; g++) raw[j][g] = load_raw16<true, uint32_t>(src + ((int64_t)j * cv + g * T) * 16, (uint32_t)t * 16u);
  }
  __builtin_amdgcn_sched_barrier(0);
  if (WAIT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  OutVec<DMXQ_BF16, 8> o[R][CG];
#pragma unroll
  for (int j = 0; j < R; j++)
#pragma unroll
    for (int g = 0; g < CG; g++) {
      float x[8], y[8];
      widen<DMXQ_BF16, 8>(raw[j][g], x);
#pragma unroll
      for (int k = 0; k < 8; k++) y[k] = x[k] * s[g][k];
      o[j][g] = pack_vec<DMXQ_BF16, 8>(y);
      __builtin_amdgcn_sched_barrier(0);
    }
  char* dst = (char*)out + ((int64_t)r0 * cv + (int64_t)by * CG * T) * 16;
#pragma unroll
  for (int j = 0; j < R; j++)
#pragma unroll
    for (int g = 0; g < CG; g++) store_out<DMXQ_BF16, 8, true>(dst + ((int64_t)j * cv + g * T) * 16 + (uint32_t)t * 16u, o[j][g]);
}

template <int T, int R, int CG, int ORDER, int TABLE, int WAIT>
__global__ __launch_bounds__(T) void ldxp(const void* __restrict__ in, void* __restrict__ out, int rows, int cv, Pad pad, const float* __restrict__ sc) {
  const int strips = cv / (T * CG);
  int bx, by;
  if (ORDER == 0) { bx = blockIdx.x; by = blockIdx.y; }
  else { by = blockIdx.x % strips; bx = blockIdx.x / strips; }
  const int t = threadIdx.x;
  const int r0 = bx * R;
  float s[CG][8];
#pragma unroll
  for (int g = 0; g < CG; g++) {
    const int cb = (by * CG + g) * T + t;
    if (TABLE) {
      const f32x4 a = *(const f32x4*)(sc + (int64_t)cb * 8), b = *(const f32x4*)(sc + (int64_t)cb * 8 + 4);
      s[g][0] = a.x; s[g][1] = a.y; s[g][2] = a.z; s[g][3] = a.w; s[g][4] = b.x; s[g][5] = b.y; s[g][6] = b.z; s[g][7] = b.w;
    } else {
#pragma unroll
      for (int k = 0; k < 8; k++) s[g][k] = 0.5f + 0.001f * (float)((cb * 8 + k) & 127);
    }
  }
  u32x4 raw[R][CG];
  const char* src = (const char*)in + ((int64_t)r0 * cv + (int64_t)by * CG * T) * 16;
#pragma unroll
  for (int j = 0; j < R; j++)
#pragma unroll
    for (int g = 0; g < CG; g++) raw[j][g] = load_raw16<true, uint32_t>(src + ((int64_t)j * cv + g * T) * 16, (uint32_t)t * 16u);
  __builtin_amdgcn_sched_barrier(0);
  if (WAIT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  OutVec<DMXQ_BF16, 8> o[R][CG];
#pragma unroll
  for (int j = 0; j < R; j++)
#pragma unroll
    for (int g = 0; g < CG; g++) {
      float x[8], y[8];
      widen<DMXQ_BF16, 8>(raw[j][g], x);
#pragma unroll
      for (int k = 0; k < 8; k++) y[k] = x[k] * s[g][k];
      o[j][g] = pack_vec<DMXQ_BF16, 8>(y);
      __builtin_amdgcn_sched_barrier(0);
    }
  char* dst = (char*)out + ((int64_t)r0 * cv + (int64_t)by * CG * T) * 16;
#pragma unroll
  for (int j = 0; j < R; j++)
#pragma unroll
    for (int g = 0; g < CG; g++) store_out<DMXQ_BF16, 8, true>(dst + ((int64_t)j * cv + g * T) * 16 + (uint32_t)t * 16u, o[j][g]);
}

template <int T, int R, int CG, int ORDER, int TABLE, int WAIT>
__global__ __launch_bounds__(T) void ldxq(const float* __restrict__ sc, const void* __restrict__ in, int rows, int cv, Pad pad, void* __restrict__ out) {
  const int strips = cv / (T * CG);
  int bx, by;
  if (ORDER == 0) { bx = blockIdx.x; by = blockIdx.y; }
  else { by = blockIdx.x % strips; bx = blockIdx.x / strips; }
  const int t = threadIdx.x;
  const int r0 = bx * R;
  float s[CG][8];
#pragma unroll
  for (int g = 0; g < CG; g++) {
    const int cb = (by * CG + g) * T + t;
    if (TABLE) {
      const f32x4 a = *(const f32x4*)(sc + (int64_t)cb * 8), b = *(const f32x4*)(sc + (int64_t)cb * 8 + 4);
      s[g][0] = a.x; s[g][1] = a.y; s[g][2] = a.z; s[g][3] = a.w; s[g][4] = b.x; s[g][5] = b.y; s[g][6] = b.z; s[g][7] = b.w;
    } else {
#pragma unroll
      for (int k = 0; k < 8; k++) s[g][k] = 0.5f + 0.001f * (float)((cb * 8 + k) & 127);
    }
  }
  u32x4 raw[R][CG];
  const char* src = (const char*)in + ((int64_t)r0 * cv + (int64_t)by * CG * T) * 16;
#pragma unroll
  for (int j = 0; j < R; j++)
#pragma unroll
    for (int g = 0; g < CG; g++) raw[j][g] = load_raw16<true, uint32_t>(src + ((int64_t)j * cv + g * T) * 16, (uint32_t)t * 16u);
  __builtin_amdgcn_sched_barrier(0);
  if (WAIT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  OutVec<DMXQ_BF16, 8> o[R][CG];
#pragma unroll
  for (int j = 0; j < R; j++)
#pragma unroll
    for (int g = 0; g < CG; g++) {
      float x[8], y[8];
      widen<DMXQ_BF16, 8>(raw[j][g], x);
#pragma unroll
      for (int k = 0; k < 8; k++) y[k] = x[k] * s[g][k];
      o[j][g] = pack_vec<DMXQ_BF16, 8>(y);
      __builtin_amdgcn_sched_barrier(0);
    }
  char* dst = (char*)out + ((int64_t)r0 * cv + (int64_t)by * CG * T) * 16;
#pragma unroll
  for (int j = 0; j < R; j++)
#pragma unroll
    for (int g = 0; g < CG; g++) store_out<DMXQ_BF16, 8, true>(dst + ((int64_t)j * cv + g * T) * 16 + (uint32_t)t * 16u, o[j][g]);
}

// the flat-stream tile of the same arithmetic (uniform multiplier): T lanes x U vectors, contiguous
template <int T, int U, int WAIT, int PACE = 0>
__global__ __launch_bounds__(T) void flat(const void* __restrict__ in, void* __restrict__ out, float s) {
  const char* src = (const char*)in + (int64_t)blockIdx.x * (T * U * 16);
  char* dst = (char*)out + (int64_t)blockIdx.x * (T * U * 16);
  u32x4 raw[U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    raw[u] = load_raw16<true, uint32_t>(src + u * (T * 16), threadIdx.x * 16u);
    if (PACE) { __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < PACE; q++) asm volatile("s_nop 7"); __builtin_amdgcn_sched_barrier(0); }
  }
  __builtin_amdgcn_sched_barrier(0);
  if (WAIT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  OutVec<DMXQ_BF16, 8> o[U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    float x[8], y[8];
    widen<DMXQ_BF16, 8>(raw[u], x);
#pragma unroll
    for (int k = 0; k < 8; k++) y[k] = x[k] * s;
    o[u] = pack_vec<DMXQ_BF16, 8>(y);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int u = 0; u < U; u++) store_out<DMXQ_BF16, 8, true>(dst + u * (T * 16) + threadIdx.x * 16u, o[u]);
}

template <int T, int U, int PACE>
__global__ __launch_bounds__(T) void flatw(const void* __restrict__ in, void* __restrict__ out, float s) {
  const uint32_t w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const char* src = (const char*)in + (int64_t)blockIdx.x * (T * U * 16) + (int64_t)w * (U * 1024);
  char* dst = (char*)out + (int64_t)blockIdx.x * (T * U * 16) + (int64_t)w * (U * 1024);
  u32x4 raw[U];
#pragma unroll
  for (int u = 0; u < U; u++) { raw[u] = load_raw16<true, uint32_t>(src + u * 1024, l * 16u); if (u + 1 < U) pace_issue<PACE>(); }
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  OutVec<DMXQ_BF16, 8> o[U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    float x[8], y[8];
    widen<DMXQ_BF16, 8>(raw[u], x);
#pragma unroll
    for (int k = 0; k < 8; k++) y[k] = x[k] * s;
    o[u] = pack_vec<DMXQ_BF16, 8>(y);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int u = 0; u < U; u++) store_out<DMXQ_BF16, 8, true>(dst + u * 1024 + l * 16u, o[u]);
}
struct Variant { std::string name; std::function<void(const void*, void*, hipStream_t)> run; std::vector<float> us; };
int main(int argc, char** argv) {
  const int ROUNDS = argc > 1 ? atoi(argv[1]) : 7;
  const int rows = 4096, C = 4096, cv = C / 8; const int64_t n = (int64_t)rows * C;
  const int NBUF = 20, LAUNCHES = 50;
  std::vector<void*> in(NBUF), out(NBUF);
  std::vector<uint16_t> h(n);
  uint64_t s = 88172645463325252ull;
  for (int64_t i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint16_t)(((s >> 20) & 0x8FFF) | 0x3000) ^ (uint16_t)((s >> 40) & 0x0F00); }
  for (int b = 0; b < NBUF; b++) { CK(hipMalloc(&in[b], n * 2)); CK(hipMalloc(&out[b], n * 2)); CK(hipMemcpy(in[b], h.data(), n * 2, hipMemcpyHostToDevice)); }
  std::vector<float> hs(C); for (int c = 0; c < C; c++) hs[c] = 0.5f + 0.001f * (float)(c % 97);
  float* d_scale; CK(hipMalloc(&d_scale, C * 4)); CK(hipMemcpy(d_scale, hs.data(), C * 4, hipMemcpyHostToDevice));
  hipStream_t st; CK(hipStreamCreate(&st));
  std::vector<Variant> vs;
  std::vector<int64_t> hz(C); for (int c = 0; c < C; c++) hz[c] = (c % 7) - 3;
  int64_t* d_zp; CK(hipMalloc(&d_zp, C * 8)); CK(hipMemcpy(d_zp, hz.data(), C * 8, hipMemcpyHostToDevice));
  const FixedFmt ff{0, 1, DMXQ_ROUND_NEAREST, -128.0f, 127.0f, 0ull};
  const ChannelMap cm = make_channel_map(C, 1, 1, n);
  const FixedOp<kLast, true> fop{ff, cm, d_scale, d_zp};
  const ScaleOp<true, kLast> dop{cm, d_scale};
  const ScaleOp<false, kLast> mop{cm, d_scale};
  const int64_t rows64 = rows, C64 = C;
#define ADD_LD(NAME, OPV, T, R) vs.push_back({std::string(NAME) + " product T" #T " R" #R, [=](const void* i, void* o, hipStream_t q) { \
    const int lpr = cv < T ? cv : T, rpp = T / lpr, strips = (cv + lpr - 1) / lpr; \
    int64_t gx = (rows64 + (int64_t)rpp * R - 1) / ((int64_t)rpp * R); \
    hipLaunchKernelGGL((lastdim_kernel<DMXQ_BF16, DMXQ_BF16, decltype(OPV), T, R>), dim3((unsigned)gx, (unsigned)strips), dim3(T), 0, q, i, o, rows64, C64, cv, make_fastdiv_u32(lpr), rpp, OPV); }, {}})
#define ADD_LDP(NAME, OPV, T, R, P) vs.push_back({std::string(NAME) + " product T" #T " R" #R " pace " #P, [=](const void* i, void* o, hipStream_t q) { \
    const int lpr = cv < T ? cv : T, rpp = T / lpr, strips = (cv + lpr - 1) / lpr; \
    int64_t gx = (rows64 + (int64_t)rpp * R - 1) / ((int64_t)rpp * R); \
    hipLaunchKernelGGL((lastdim_kernel<DMXQ_BF16, DMXQ_BF16, decltype(OPV), T, R, 16, P>), dim3((unsigned)gx, (unsigned)strips), dim3(T), 0, q, i, o, rows64, C64, cv, make_fastdiv_u32(lpr), rpp, OPV); }, {}})
#define SWP(NAME, OPV, T, R) ADD_LDP(NAME, OPV, T, R, 0); ADD_LDP(NAME, OPV, T, R, 1); ADD_LDP(NAME, OPV, T, R, 2); ADD_LDP(NAME, OPV, T, R, 3); ADD_LDP(NAME, OPV, T, R, 4); ADD_LDP(NAME, OPV, T, R, 6); ADD_LDP(NAME, OPV, T, R, 12)

#define LDX(T, R, CG, ORDER, TABLE, WAIT) vs.push_back({"ldx T" #T " R" #R " CG" #CG " ord" #ORDER " tab" #TABLE " wait" #WAIT, [=](const void* i, void* o, hipStream_t q) { \
    const int strips = cv / (T * CG), gx = rows / R; \
    if (ORDER == 0) hipLaunchKernelGGL((ldx<T, R, CG, ORDER, TABLE, WAIT>), dim3(gx, strips), dim3(T), 0, q, i, o, rows, cv, d_scale); \
    else hipLaunchKernelGGL((ldx<T, R, CG, ORDER, TABLE, WAIT>), dim3(gx * strips), dim3(T), 0, q, i, o, rows, cv, d_scale); }, {}})
#define FLATP(T, U, PACE) vs.push_back({"flat " #T "x" #U " wait1 pace" #PACE, [=](const void* i, void* o, hipStream_t q) { \
    hipLaunchKernelGGL((flat<T, U, 1, PACE>), dim3((unsigned)(n / 8 / (T * U))), dim3(T), 0, q, i, o, 0.75f); }, {}})
#define FLAT(T, U, WAIT) vs.push_back({"flat " #T "x" #U " wait" #WAIT, [=](const void* i, void* o, hipStream_t q) { \
    hipLaunchKernelGGL((flat<T, U, WAIT>), dim3((unsigned)(n / 8 / (T * U))), dim3(T), 0, q, i, o, 0.75f); }, {}})
#define FLATW(T, U, P) vs.push_back({"flatw (wave-contiguous) " #T "x" #U " pace " #P, [=](const void* i, void* o, hipStream_t q) { \
    hipLaunchKernelGGL((flatw<T, U, P>), dim3((unsigned)(n / 8 / (T * U))), dim3(T), 0, q, i, o, 0.75f); }, {}})
#define FLATP2(T, U, P) vs.push_back({"flat " #T "x" #U " pace " #P, [=](const void* i, void* o, hipStream_t q) { \
    hipLaunchKernelGGL((flat<T, U, 1, P>), dim3((unsigned)(n / 8 / (T * U))), dim3(T), 0, q, i, o, 0.75f); }, {}})
#define BOTH(T, U) FLATP2(T, U, 0); FLATP2(T, U, 2); FLATP2(T, U, 4); FLATW(T, U, 0); FLATW(T, U, 2); FLATW(T, U, 4)
#define ADD_LDS(NAME, OPV, T, R, S) vs.push_back({std::string(NAME) + " product T" #T " R" #R " pace " #S, [=](const void* i, void* o, hipStream_t q) { \
    const int lpr = cv < T ? cv : T, rpp = T / lpr, strips = (cv + lpr - 1) / lpr; \
    int64_t gx = (rows64 + (int64_t)rpp * R - 1) / ((int64_t)rpp * R); \
    hipLaunchKernelGGL((lastdim_kernel<DMXQ_BF16, DMXQ_BF16, decltype(OPV), T, R, 16, S>), dim3((unsigned)gx, (unsigned)strips), dim3(T), 0, q, i, o, rows64, C64, cv, make_fastdiv_u32(lpr), rpp, OPV); }, {}})
#define SWS(NAME, OPV, T, R) ADD_LDS(NAME, OPV, T, R, 0); ADD_LDS(NAME, OPV, T, R, 1); ADD_LDS(NAME, OPV, T, R, 2); ADD_LDS(NAME, OPV, T, R, 3); ADD_LDS(NAME, OPV, T, R, 4); ADD_LDS(NAME, OPV, T, R, 6); ADD_LDS(NAME, OPV, T, R, 12)
  SWS("int8", fop, 256, 16); SWS("div", dop, 256, 16); SWS("int8", fop, 256, 8); SWS("div", dop, 256, 8); SWS("mul", mop, 256, 16);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 30; w++) for (auto& v : vs) for (int i = 0; i < 10; i++) v.run(in[i % NBUF], out[i % NBUF], st);
  CK(hipStreamSynchronize(st)); CK(hipGetLastError());
  for (int r = 0; r < ROUNDS; r++)
    for (auto& v : vs) {
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < LAUNCHES; i++) v.run(in[i % NBUF], out[i % NBUF], st);
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); v.us.push_back(ms * 1e3f / LAUNCHES);
    }
  for (auto& v : vs) { std::sort(v.us.begin(), v.us.end()); float med = v.us[v.us.size() / 2];
    printf("%-44s min %6.2f med %6.2f  %5.1f%%\n", v.name.c_str(), v.us[0], med, 100.0 * 4.0 * n / (med * 1e-6) / 8e12); }
  return 0;
}
