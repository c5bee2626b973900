// csrc/gate_registry.hpp — HOST side of the reductions' init gate (csrc/reduce.hip): the flag slots per device, one per stream, behind
// one mutex.  A header of its own (round 6) so that it can be compiled WITHOUT the device code and without a GPU: tools/sanitize/
// gate_registry_harness.cpp includes it with stand-in definitions of the five HIP entry points it calls and runs concurrent host
// threads through it under ThreadSanitizer and AddressSanitizer (profiles/r06_sanitizers.txt).  Included by reduce.hip only.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdlib.h>

#include <mutex>
#include <unordered_map>
#include <vector>

namespace dmxq {
struct InitGate { unsigned* flag; unsigned epoch; int on; };
constexpr int kGateSlots = 1024, kGateStride = 16 /* words: one slot per 64-byte line */, kGateMaxOut = 8192;
}  // namespace dmxq

using namespace dmxq;

// host side of the init gate: flag slots per device, one per stream
namespace {
struct GateDevice {
  unsigned* flags = nullptr;
  bool failed = false;
  std::unordered_map<unsigned long long, int> slot_of;   // stream key (see take_gate) -> slot
  std::vector<unsigned> epoch;
};
constexpr int kGateDevices = 64;
std::mutex g_gate_mu;
GateDevice g_gate[kGateDevices];
// 0: gate on; 1: off (the fill launch in front of every reduction); 2: on, workgroup (0, 0) does not volunteer (tests: every launch
// takes the takeover path).  Initial value from DMXQ_NO_INIT_GATE (set and not "0": off).
int g_gate_mode = [] { const char* e = getenv("DMXQ_NO_INIT_GATE"); return (e && e[0] && !(e[0] == '0' && !e[1])) ? 1 : 0; }();

InitGate take_gate(hipStream_t s, int64_t n_out) {
  const InitGate none{nullptr, 0u, 0};
  if (n_out > kGateMaxOut || s == hipStreamPerThread) return none;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); return none; }
  if (cap != hipStreamCaptureStatusNone) return none;
  // the slot belongs to the STREAM's device (not the thread's current one) and to the stream's id
  hipDevice_t dev = -1;
  if (hipStreamGetDevice(s, &dev) != hipSuccess || dev < 0 || dev >= kGateDevices) { (void)hipGetLastError(); return none; }
  // hipStreamGetId (HIP 7.1) where the loaded runtime has it -- resolved at run time: torch's bundled libamdhip64 is 7.0 and a link-time
  // reference would keep the library from loading there --, else the handle (a handle is only reused after its stream's work is done)
  typedef hipError_t (*stream_id_fn)(hipStream_t, unsigned long long*);
  static const stream_id_fn get_id = (stream_id_fn)dlsym(RTLD_DEFAULT, "hipStreamGetId");
  unsigned long long sid = (unsigned long long)(uintptr_t)s;
  if (get_id && s != nullptr) {
    unsigned long long id = 0ull;
    if (get_id(s, &id) == hipSuccess) sid = (id << 1) | 1ull;   // (odd: never equal to a handle, which is at least 2-byte aligned)
    else (void)hipGetLastError();
  }
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); return none; }
  std::lock_guard<std::mutex> lk(g_gate_mu);
  if (g_gate_mode == 1) return none;
  GateDevice& G = g_gate[dev];
  if (G.failed) return none;
  if (!G.flags) {
    if (cur != (int)dev) return none;   // (the flag words are allocated from a call whose current device is the stream's)
    // (another thread capturing in global mode makes hipMalloc fail: the fill launch serves until a later call succeeds)
    unsigned* p = nullptr;
    const size_t bytes = (size_t)kGateSlots * kGateStride * sizeof(unsigned);
    if (hipMalloc((void**)&p, bytes) != hipSuccess) { (void)hipGetLastError(); return none; }
    if (hipMemset(p, 0, bytes) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(p); G.failed = true; return none; }
    G.flags = p;
    G.epoch.assign(kGateSlots, 0u);
  }
  int slot;
  auto it = G.slot_of.find(sid);
  if (it != G.slot_of.end()) slot = it->second;
  else {
    if ((int)G.slot_of.size() >= kGateSlots) return none;
    slot = (int)G.slot_of.size();
    G.slot_of.emplace(sid, slot);
  }
  unsigned e = ++G.epoch[slot];
  if (e == 0u) e = ++G.epoch[slot];
  return InitGate{G.flags + (size_t)slot * kGateStride, e, g_gate_mode == 2 ? 2 : 1};
}
}  // namespace

// Test / diagnosis hook (not part of include/dmxq.h): 0 gate on, 1 gate off, 2 gate on without the volunteer; returns the old mode
extern "C" int dmxq_internal_gate_mode(int mode) {
  std::lock_guard<std::mutex> lk(g_gate_mu);
  const int old = g_gate_mode;
  if (mode >= 0 && mode <= 2) g_gate_mode = mode;
  return old;
}

