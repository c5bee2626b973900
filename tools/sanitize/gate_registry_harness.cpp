// tools/sanitize/gate_registry_harness.cpp — the HOST side of the reductions' init gate (csrc/gate_registry.hpp: per-device flag slots,
// one per stream, behind one mutex) under ThreadSanitizer / AddressSanitizer on the CPU, no GPU and no HIP runtime: the five HIP entry
// points the registry calls are defined HERE (plain host memory, fake stream handles whose device is encoded in the handle).
//   g++ -std=c++17 -g -O1 -fsanitize=thread  -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include gate_registry_harness.cpp -o gate_tsan -ldl -lpthread
//   g++ -std=c++17 -g -O1 -fsanitize=address,undefined ...                                                    -o gate_asan
// What runs: T host threads, each with its own streams plus streams SHARED with the other threads (the concurrent-host-threads pattern of
// tests/test_gpu_round5.py), on two devices, interleaved with mode switches (dmxq_internal_gate_mode) and with a capture / a failing
// allocation now and then.  Checked at the end: a stream keeps ONE slot for life, no two streams of a device share a slot, a slot's
// epochs handed out are all different and never 0, nothing is handed out while the gate is off.
#include <atomic>
#include <cstdio>
#include <cstring>
#include <map>
#include <set>
#include <thread>

#include "../../dmx-compressor_amd/csrc/gate_registry.hpp"

// ---- stand-ins for the HIP runtime (host memory; a stream handle is (device << 20 | id) << 4) -------------------------------------
static std::atomic<int> g_capturing{0}, g_fail_malloc{0};
static thread_local int t_device = 0;
static int dev_of(hipStream_t s) { return (int)(((uintptr_t)s >> 4) >> 20); }
extern "C" {
hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* st) {
  *st = g_capturing.load() ? hipStreamCaptureStatusActive : hipStreamCaptureStatusNone;
  return hipSuccess;
}
hipError_t hipStreamGetDevice(hipStream_t s, hipDevice_t* d) { *d = dev_of(s); return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = t_device; return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipMalloc(void** p, size_t n) {
  if (g_fail_malloc.load()) return hipErrorOutOfMemory;
  *p = malloc(n);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
}

int main() {
  constexpr int T = 8, OWN = 24, SHARED = 40, ROUNDS = 4000;
  struct Seen { unsigned* flag; unsigned epoch; hipStream_t s; };
  std::vector<std::vector<Seen>> seen(T);
  std::atomic<long> handed{0}, refused{0};
  auto stream = [](int dev, int id) { return (hipStream_t)(uintptr_t)((((uintptr_t)dev << 20) | (uintptr_t)(id + 1)) << 4); };
  std::vector<std::thread> th;
  for (int t = 0; t < T; t++)
    th.emplace_back([&, t] {
      unsigned rng = 12345u + 977u * (unsigned)t;
      for (int r = 0; r < ROUNDS; r++) {
        rng = rng * 1664525u + 1013904223u;
        const int dev = (rng >> 8) & 1;
        t_device = ((rng >> 9) & 7) ? dev : 1 - dev;   // (now and then the thread's current device is NOT the stream's)
        const bool own = (rng >> 12) & 1;
        const hipStream_t s = own ? stream(dev, 1000 * (t + 1) + (int)((rng >> 16) % OWN)) : stream(dev, (int)((rng >> 16) % SHARED));
        if ((rng & 0x3FF) == 7) dmxq_internal_gate_mode((rng >> 20) % 3);
        if ((rng & 0x7FF) == 11) g_capturing.store(1);
        if ((rng & 0x7FF) == 13) g_capturing.store(0);
        if ((rng & 0xFFF) == 17) g_fail_malloc.store(1);
        if ((rng & 0xFFF) == 19) g_fail_malloc.store(0);
        const InitGate g = take_gate(s, (rng >> 4) % 9000);
        if (g.on) { seen[t].push_back(Seen{g.flag, g.epoch, s}); handed++; } else refused++;
      }
    });
  for (auto& x : th) x.join();
  // ---- invariants
  std::map<hipStream_t, unsigned*> slot_of;
  std::map<unsigned*, hipStream_t> owner;
  std::map<unsigned*, std::set<unsigned>> epochs;
  long bad = 0;
  for (auto& v : seen)
    for (auto& e : v) {
      if (e.epoch == 0u) bad++;
      auto it = slot_of.find(e.s);
      if (it == slot_of.end()) slot_of[e.s] = e.flag; else if (it->second != e.flag) bad++;
      auto ow = owner.find(e.flag);
      if (ow == owner.end()) owner[e.flag] = e.s; else if (ow->second != e.s) bad++;
      if (!epochs[e.flag].insert(e.epoch).second) bad++;   // the same epoch handed out twice for one slot
    }
  printf("gate registry: %d threads x %d calls, %ld gates handed out over %zu streams, %ld refused (gate off / capture / no memory / "
         "foreign device / too many outputs), invariant violations: %ld\n", T, ROUNDS, handed.load(), slot_of.size(), refused.load(), bad);
  return bad ? 1 : 0;
}
