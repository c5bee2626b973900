// tools/tune_pack.hip — on-GPU A/B of bfp_pack_rows_kernel's store forms (not part of the product library): VAR 0 = 8-byte code
// stores + one byte store per block, 1 = 16-byte code stores (neighbour-lane exchange), 2 = exponents through LDS, 3 = both.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math -ffp-contract=off -fno-gpu-flush-denormals-to-zero -Iinclude tools/tune_pack.hip -o tools/tune_pack
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#include "../dmx-compressor_amd/csrc/bfp_pack.hip"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

struct Variant { std::string name; std::function<void(const void*, int8_t*, uint8_t*, hipStream_t)> run; std::vector<float> us; };

int main(int argc, char** argv) {
  const int ROUNDS = argc > 1 ? atoi(argv[1]) : 7;
  const int64_t rows = argc > 2 ? atoll(argv[2]) : 4096, C = argc > 3 ? atoll(argv[3]) : 4096, n = rows * C, n_vec = n / 8;
  const int NBUF = 16, LAUNCHES = 50;
  std::vector<void*> in(NBUF);
  std::vector<int8_t*> mant(NBUF);
  std::vector<uint8_t*> exps(NBUF);
  std::vector<uint16_t> h(n);
  uint64_t s = 88172645463325252ull;
  for (int64_t i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint16_t)(((s >> 20) & 0x8FFF) | 0x3000) ^ (uint16_t)((s >> 40) & 0x0F00); }
  for (int b = 0; b < NBUF; b++) { CK(hipMalloc(&in[b], n * 4)); CK(hipMalloc((void**)&mant[b], n)); CK(hipMalloc((void**)&exps[b], n / 8)); CK(hipMemcpy(in[b], h.data(), n * 2, hipMemcpyHostToDevice)); }
  hipStream_t st; CK(hipStreamCreate(&st));
  std::vector<Variant> vs;
#define ADDG(B_, V_, T_, U_) vs.push_back({"B=" #B_ " VAR " #V_ " " #T_ "x" #U_, [=](const void* i, int8_t* m, uint8_t* e, hipStream_t q) { \
    int lpb = B_ / 8, lg = 0; while ((1 << lg) < lpb) lg++; \
    hipLaunchKernelGGL((bfp_pack_rows_kernel<DMXQ_BF16, V_, T_, U_>), dim3((unsigned)((n_vec + T_ * U_ - 1) / (T_ * U_))), dim3(T_), 0, q, i, m, e, n_vec, lpb, lg, 8, 0); }, {}})
#define ADD(B_, V_) ADDG(B_, V_, 256, 4)
  ADD(16, 0); ADD(16, 1); ADD(16, 2); ADD(16, 3); ADD(64, 0); ADD(64, 3);
  ADDG(16, 3, 256, 2); ADDG(16, 3, 256, 8); ADDG(16, 3, 512, 2); ADDG(16, 3, 512, 4); ADDG(16, 3, 128, 4); ADDG(16, 3, 128, 8); ADDG(16, 3, 64, 8); ADDG(16, 3, 1024, 2);
  ADDG(64, 3, 256, 2); ADDG(64, 3, 256, 8); ADDG(64, 3, 512, 2); ADDG(64, 3, 128, 8);
  // unpack: the codes / exponents buffers written by the warm-up of the pack variants above; out -> the (then unused) input buffers
  int bshift16 = 1, bshift64 = 3;
#define ADDU(P_, U_) vs.push_back({"unpack B=16 " #P_ " x" #U_, [=](const void* i, int8_t* m, uint8_t* e, hipStream_t q) { \
    hipLaunchKernelGGL((bfp_unpack_vec_kernel<DMXQ_BF16, P_, U_>), dim3((unsigned)((n_vec + kThreads * U_ - 1) / (kThreads * U_))), dim3(kThreads), 0, q, m, e, (void*)i, n_vec, bshift16, 8); }, {}})
#define ADDUF(P_, U_, BS_) vs.push_back({"unpack f32 bs" #BS_ " " #P_ " x" #U_, [=](const void* i, int8_t* m, uint8_t* e, hipStream_t q) { \
    hipLaunchKernelGGL((bfp_unpack_vec_kernel<DMXQ_F32, P_, U_>), dim3((unsigned)((n_vec + kThreads * U_ - 1) / (kThreads * U_))), dim3(kThreads), 0, q, m, e, (void*)i, n_vec, BS_, 8); }, {}})
#define ADDUR(U_, BL_) vs.push_back({"unpack f32 regions B=2^" #BL_ " x" #U_, [=](const void* i, int8_t* m, uint8_t* e, hipStream_t q) { \
    hipLaunchKernelGGL((bfp_unpack_f32_kernel<U_>), dim3((unsigned)((n_vec + kThreads * U_ - 1) / (kThreads * U_))), dim3(kThreads), 0, q, m, e, (float*)i, n_vec * 8, BL_, 8); }, {}})
  if (getenv("TUNE_UNPACK_F32")) { vs.clear(); ADDUR(1, 4); ADDUR(2, 4); ADDUR(4, 4); ADDUR(8, 4); ADDUR(2, 6); ADDUR(4, 6); ADDUF(false, 8, 1); ADDUF(true, 8, 1); ADDUF(false, 4, 1); ADDUF(true, 4, 1); ADDUF(false, 2, 1); ADDUF(true, 2, 1); ADDUF(true, 2, 3); ADDUF(true, 4, 3); ADDUF(true, 8, 3); }
  else if (getenv("TUNE_UNPACK")) { vs.clear(); ADDU(false, 8); ADDU(true, 8); ADDU(false, 4); ADDU(true, 4); ADDU(true, 2); ADDU(true, 16); (void)bshift64; }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (auto& v : vs) for (int i = 0; i < 10; i++) v.run(in[i % NBUF], mant[i % NBUF], exps[i % NBUF], st);
  CK(hipStreamSynchronize(st));
  CK(hipGetLastError());
  for (int r = 0; r < ROUNDS; r++)
    for (auto& v : vs) {
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < LAUNCHES; i++) v.run(in[i % NBUF], mant[i % NBUF], exps[i % NBUF], st);
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      v.us.push_back(ms * 1e3f / LAUNCHES);
    }
  printf("# bf16 [%lld, %lld] -> int8 codes + uint8 exponents\n%-16s %9s %9s %9s\n", (long long)rows, (long long)C, "variant", "min_us", "med_us", "%8TB/s");
  for (auto& v : vs) {
    std::sort(v.us.begin(), v.us.end());
    const float med = v.us[v.us.size() / 2];
    const double bytes = (v.name.find("f32") != std::string::npos ? 5.0 : 3.0) * n + (double)n / (v.name[0] == 'u' || v.name[2] == '1' ? 16 : 64);
    printf("%-16s %9.2f %9.2f %8.1f%%\n", v.name.c_str(), v.us[0], med, 100.0 * bytes / (med * 1e-6) / 8e12);
  }
  return 0;
}
