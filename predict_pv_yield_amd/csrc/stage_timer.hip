// Opt-in per-stage device timing of the multi-kernel entry points (pv_farneback_batch_u8 and the streaming stages of
// the optical-flow advection): bench.py asks for it to report a roofline PER STAGE from inside
// the run, with HIP events on the stream the kernels are launched on.
//
//   pv_stage_timing_begin()                      arm (drops earlier marks)
//   ... any pv_* calls ...                       each stage boundary records one event on the call's stream
//   pv_stage_timing_end(names, ms, cap, &n)      waits for the last event; per distinct stage label (first-seen order):
//                                                summed milliseconds and number of occurrences
// Disarmed (the default) a mark is one relaxed load and a branch; nothing is recorded while a stream is being captured
// into a hipGraph.  The facility is process-global and not re-entrant: one timing session at a time, one stream.
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "pv_common.h"

namespace pv {

namespace {
std::atomic<int> g_armed{0};
std::mutex g_mu;
struct Mark { const char* label; hipEvent_t ev; };
std::vector<Mark> g_marks;
std::vector<hipEvent_t> g_pool;

hipEvent_t take_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
}  // namespace

// Called by the launchers: "everything enqueued on `st` from here up to the next mark belongs to `label`".
// label == nullptr closes the current stage (end of an entry point).
void stage_mark(const char* label, hipStream_t st) {
  if (!g_armed.load(std::memory_order_relaxed)) return;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return;
  std::lock_guard<std::mutex> lk(g_mu);
  hipEvent_t e = take_event();
  if (!e) return;
  if (hipEventRecord(e, st) != hipSuccess) { g_pool.push_back(e); return; }
  g_marks.push_back({label, e});
}

}  // namespace pv

extern "C" {

int pv_stage_timing_begin(void) {
  std::lock_guard<std::mutex> lk(pv::g_mu);
  for (auto& m : pv::g_marks) pv::g_pool.push_back(m.ev);
  pv::g_marks.clear();
  pv::g_armed.store(1);
  return PV_OK;
}

int pv_stage_timing_end(const char** names, float* ms, int32_t* counts, int32_t capacity, int32_t* n_out) {
  PV_REQUIRE(n_out && (capacity == 0 || (names && ms && counts)), PV_EINVAL, "pv_stage_timing_end: null pointer");
  pv::g_armed.store(0);
  std::lock_guard<std::mutex> lk(pv::g_mu);
  int n = 0;
  if (!pv::g_marks.empty()) {
    hipError_t e = hipEventSynchronize(pv::g_marks.back().ev);
    PV_REQUIRE(e == hipSuccess, PV_ELAUNCH, "pv_stage_timing_end: %s", hipGetErrorString(e));
    for (size_t i = 0; i + 1 < pv::g_marks.size(); ++i) {
      const char* label = pv::g_marks[i].label;
      if (!label) continue;   // a closing mark: the gap up to the next entry point belongs to nobody
      float dt = 0.f;
      if (hipEventElapsedTime(&dt, pv::g_marks[i].ev, pv::g_marks[i + 1].ev) != hipSuccess) continue;
      int j = 0;
      for (; j < n; ++j)
        if (names[j] == label || std::string(names[j]) == label) break;
      if (j == n) {
        if (n >= capacity) continue;
        names[n] = label; ms[n] = 0.f; counts[n] = 0; ++n;
      }
      ms[j] += dt;
      counts[j] += 1;
    }
  }
  for (auto& m : pv::g_marks) pv::g_pool.push_back(m.ev);
  pv::g_marks.clear();
  *n_out = n;
  return PV_OK;
}

}  // extern "C"
