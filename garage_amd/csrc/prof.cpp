// Optional per-launch timing with HIP events (used by bench.py's roofline leg).
// Disabled by default.  The events are attached to the kernel dispatch itself
// (hipExtLaunchKernelGGL), on the stream the kernel is launched on, so the
// elapsed time is the kernel's own duration on the device.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

#include "prof.h"

namespace {
struct Sample {
  hipEvent_t start, stop;
  int kind;
  double work;
};
bool g_on = false;
// launches per kind since the library was loaded, counted whether timing is on or
// not (tests assert which kernels an update took: ga_launch_count)
int64_t g_launches[GA_PROF_KINDS] = {0};
std::vector<Sample> g_samples;
std::vector<hipEvent_t> g_pool;
constexpr size_t kMaxSamples = 200000;

hipEvent_t get_event() {
  if (!g_pool.empty()) {
    hipEvent_t e = g_pool.back();
    g_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}
}  // namespace

void ga_prof_events(int kind, double work, hipEvent_t* start, hipEvent_t* stop) {
  *start = nullptr;
  *stop = nullptr;
  if (kind >= 0 && kind < GA_PROF_KINDS) ++g_launches[kind];
  if (!g_on || g_samples.size() >= kMaxSamples) return;
  Sample s;
  s.start = get_event();
  s.stop = get_event();
  s.kind = kind;
  s.work = work;
  g_samples.push_back(s);
  *start = s.start;
  *stop = s.stop;
}

void ga_prof_count(int kind) {
  if (kind >= 0 && kind < GA_PROF_KINDS) ++g_launches[kind];
}

extern "C" int64_t ga_launch_count(int kind) {
  return (kind >= 0 && kind < GA_PROF_KINDS) ? g_launches[kind] : -1;
}

extern "C" int ga_prof_enable(int on) {
  g_on = on != 0;
  return 0;
}

// out[kind * 3 + {0,1,2}] = {total milliseconds, total work, launches}
extern "C" int ga_prof_collect(double* out_host, int n_kinds) {
  for (int i = 0; i < n_kinds * 3; ++i) out_host[i] = 0.0;
  for (auto& s : g_samples) {
    (void)hipEventSynchronize(s.stop);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, s.start, s.stop);
    if (s.kind >= 0 && s.kind < n_kinds) {
      out_host[s.kind * 3 + 0] += (double)ms;
      out_host[s.kind * 3 + 1] += s.work;
      out_host[s.kind * 3 + 2] += 1.0;
    }
    g_pool.push_back(s.start);
    g_pool.push_back(s.stop);
  }
  g_samples.clear();
  return 0;
}
