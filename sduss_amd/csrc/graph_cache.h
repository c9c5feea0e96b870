// hipGraph replay of a step plan (unet_sdxl.cpp, mmdit_sd3.cpp) -- OPT-IN (MX_GRAPH=1).
//
// A forward is ~1.3k (SDXL) / ~0.5k (MMDiT) kernel launches and the plan is a pure function of its arguments (weights,
// arena workspace, caller tensors -- all addressed through pointers that repeat from step to step under torch's caching
// allocator), so the launch sequence can be captured once per argument tuple and replayed with one hipGraphLaunch.
// Measured on MI355X (same box, bench.py --batch 1 / 4, 12 replays of 1 capture): 30.3 vs 29.7 ms/step at one 1024 px request,
// 64.3 vs 64.1 ms at four -- the step is NOT host-launch-bound even at batch 1 (its floor there is the per-kernel prologue /
// epilogue of ~1.3k launches that fill a quarter of the chip), so the replay buys nothing today and stays off by default.
// It is kept for hosts whose launch path is slower (many replicas per CPU socket).
//   * key = every scalar and pointer argument of the forward; a different tuple captures a new graph (LRU of 8);
//   * 16 misses in a row (pointers that never repeat) switch the cache off for the handle: the plan then runs eagerly;
//   * capture needs a non-legacy stream: calls on the null stream are forked to a private non-blocking stream and joined
//     back with events (asynchronous on both sides);
//   * off while mx_profile is enabled (per-launch events).  tests/test_unet_gpu.py::test_unet_graph_replay_* checks that a
//     replay reads the current buffer contents (run with MX_GRAPH=1 to exercise the replay path).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "common.h"

namespace mx {

struct GraphCache {
  struct Entry { std::vector<uint64_t> key; hipGraphExec_t exec; uint64_t stamp; hipStream_t last; };   // last: stream of the latest launch
  std::vector<Entry> entries;
  uint64_t clock = 0;
  int miss_streak = 0;
  bool disabled = false;
  hipStream_t side = nullptr;
  hipEvent_t ev_in = nullptr, ev_out = nullptr;
  long n_replay = 0, n_capture = 0, n_eager = 0;   // MX_GRAPH_DEBUG=1 prints them when the handle is destroyed

  static bool env_on() {
    static const bool on = [] { const char* e = getenv("MX_GRAPH"); return e && e[0] == '1'; }();
    return on;
  }
  ~GraphCache() {
    if (getenv("MX_GRAPH_DEBUG")) fprintf(stderr, "[mx graph] replays %ld captures %ld eager %ld disabled %d\n", n_replay, n_capture, n_eager, (int)disabled);
    clear();
    if (ev_in) (void)hipEventDestroy(ev_in);
    if (ev_out) (void)hipEventDestroy(ev_out);
    if (side) (void)hipStreamDestroy(side);
  }

  // An exec may still be running when it is dropped (the launch is asynchronous): wait for the stream it last ran on first.
  static void retire(Entry& e) {
    if (e.last) (void)hipStreamSynchronize(e.last);
    (void)hipGraphExecDestroy(e.exec);
  }
  // drop every captured graph: called when the handle's weight table changes (the key holds pointers, not the table)
  void clear() {
    for (auto& e : entries) retire(e);
    entries.clear();
    miss_streak = 0;
  }

  // body(stream) enqueues the plan on `stream` and returns true on success.  Returns true if the work was enqueued (by
  // replay or eagerly); false if body failed (its error is already recorded).
  // capture_on_miss == false (the mixed-resolution forwards: their key holds every group's pointers and batch sizes, which under continuous batching of
  // a mixed stream rarely repeat): a miss runs eagerly, captures nothing and does NOT count towards the streak that switches the handle's cache off --
  // otherwise a mixed stream would disable the replay of the fixed-composition forwards of the same handle (advisor, round 3).
  template <class F>
  bool run(hipStream_t user, const std::vector<uint64_t>& key, F&& body, bool capture_on_miss = true) {
    if (disabled || !env_on() || prof_enabled()) { ++n_eager; return body(user); }
    hipStream_t s = user;
    const bool forked = (user == nullptr);
    if (forked) {
      if (!side) {
        if (hipStreamCreateWithFlags(&side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ev_in, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ev_out, hipEventDisableTiming) != hipSuccess) {
          (void)hipGetLastError(); disabled = true; return body(user);
        }
      }
      s = side;
    }
    hipGraphExec_t exec = nullptr;
    for (auto& e : entries)
      if (e.key == key) { exec = e.exec; e.stamp = ++clock; e.last = s; break; }
    if (!exec && !capture_on_miss) { ++n_eager; return body(user); }
    if (!exec) {
      if (++miss_streak > 16) { disabled = true; return body(user); }
      if (hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) != hipSuccess) { (void)hipGetLastError(); disabled = true; return body(user); }
      const bool ok = body(s);
      hipGraph_t g = nullptr;
      const hipError_t ec = hipStreamEndCapture(s, &g);
      if (!ok) { if (g) (void)hipGraphDestroy(g); (void)hipGetLastError(); return false; }
      if (ec != hipSuccess || !g || hipGraphInstantiate(&exec, g, nullptr, nullptr, 0) != hipSuccess) {
        if (g) (void)hipGraphDestroy(g);
        (void)hipGetLastError(); disabled = true; return body(user);
      }
      (void)hipGraphDestroy(g);
      if (entries.size() >= 8) {
        size_t lru = 0;
        for (size_t i = 1; i < entries.size(); ++i) if (entries[i].stamp < entries[lru].stamp) lru = i;
        retire(entries[lru]);
        entries.erase(entries.begin() + lru);
      }
      entries.push_back(Entry{key, exec, ++clock, s});
      ++n_capture;
    } else {
      miss_streak = 0;
      ++n_replay;
    }
    if (forked) { (void)hipEventRecord(ev_in, user); (void)hipStreamWaitEvent(s, ev_in, 0); }
    if (hipGraphLaunch(exec, s) != hipSuccess) { (void)hipGetLastError(); disabled = true; return body(user); }
    if (forked) { (void)hipEventRecord(ev_out, s); (void)hipStreamWaitEvent(user, ev_out, 0); }
    return true;
  }
};

}  // namespace mx
