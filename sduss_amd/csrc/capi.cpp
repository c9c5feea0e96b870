// Error plumbing and the optional launch profiler of the C ABI (include/mxdenoise.h).
#include <string>
#include <vector>

#include "../../include/mxdenoise.h"
#include "common.h"

namespace mx {
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
int cu_count() {
  static const int ncu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) n = prop.multiProcessorCount;
    }
    return (n & ~7) > 0 ? (n & ~7) : 256;         // whole XCD groups, so tile % 8 stays the workgroup's XCD (gemm_tile_of_block)
  }();
  return ncu;
}

struct ProfRec { hipEvent_t a, b; int kind; double flops, bytes; int m, n, k; float ms; };
static std::vector<ProfRec> g_last;
static bool g_prof = false;
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;

bool prof_enabled() { return g_prof; }
static hipEvent_t get_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e = nullptr; (void)hipEventCreate(&e); return e;
}
void prof_begin(hipStream_t s, int kind, double flops, double bytes, int m, int n, int k) {
  if (!g_prof) return;
  ProfRec r; r.a = get_event(); r.b = get_event(); r.kind = kind; r.flops = flops; r.bytes = bytes; r.m = m; r.n = n; r.k = k; r.ms = 0.f;
  (void)hipEventRecord(r.a, s);
  g_recs.push_back(r);
}
void prof_end(hipStream_t s) {
  if (!g_prof || g_recs.empty()) return;
  (void)hipEventRecord(g_recs.back().b, s);
}
}  // namespace mx

extern "C" const char* mx_last_error(void) { return mx::g_err.c_str(); }
extern "C" int mx_version(void) { return 1; }

extern "C" int mx_profile_enable(int on) {
  mx::g_prof = on != 0;
  if (on) {
    for (auto& r : mx::g_recs) { mx::g_pool.push_back(r.a); mx::g_pool.push_back(r.b); }
    mx::g_recs.clear();
  }
  return 0;
}

// Synchronises the recorded events (call after the stream has been synchronised) and sums, per kernel kind,
// launches / milliseconds / algorithmic flops / algorithmic bytes.  out: double[4 * 10] = kind-major {n, ms, flops, bytes}.
extern "C" int mx_profile_collect(double* out) {
  MX_CHECK(out != nullptr, "profile_collect: null output");
  for (int i = 0; i < 4 * mx::PROF_KINDS; ++i) out[i] = 0.0;
  mx::g_last.clear();
  for (auto& r : mx::g_recs) {
    MX_HIP(hipEventSynchronize(r.b));
    float ms = 0.f;
    MX_HIP(hipEventElapsedTime(&ms, r.a, r.b));
    double* o = out + 4 * r.kind;
    o[0] += 1.0; o[1] += ms; o[2] += r.flops; o[3] += r.bytes;
    r.ms = ms; mx::g_last.push_back(r);
    mx::g_pool.push_back(r.a); mx::g_pool.push_back(r.b);
  }
  mx::g_recs.clear();
  return 0;
}

// per-launch records of the last mx_profile_collect: out[i*6 .. i*6+5] = {kind, M, N, K, ms, flops}; returns the count
extern "C" int mx_profile_records(double* out, int max_records) {
  int n = 0;
  for (auto& r : mx::g_last) {
    if (n >= max_records) break;
    double* o = out + 6 * n;
    o[0] = r.kind; o[1] = r.m; o[2] = r.n; o[3] = r.k; o[4] = r.ms; o[5] = r.flops;
    ++n;
  }
  return n;
}

/* ---- host-side evaluation of a fitted random forest (the block-skip predictor; include/mxdenoise.h) ---- */
extern "C" int mx_forest_predict(const int32_t* left, const int32_t* right, const int32_t* feature, const double* threshold, const double* p1,
                                 const int32_t* roots, int n_trees, const float* X, int n_rows, int n_feat, unsigned char* out) {
  MX_CHECK(left && right && feature && threshold && p1 && roots && X && out && n_trees > 0 && n_rows >= 0 && n_feat > 0,
           "forest_predict: bad arguments");
  for (int r = 0; r < n_rows; ++r) {
    const float* x = X + (size_t)r * n_feat;
    double acc = 0.0;
    for (int t = 0; t < n_trees; ++t) {
      int node = roots[t];
      while (left[node] >= 0) {                            // leaves carry -1 (sklearn TREE_LEAF)
        const int f = feature[node];
        if (f < 0 || f >= n_feat) { mx::set_error("forest_predict: feature index outside the row"); return 1; }
        node = (double)x[f] <= threshold[node] ? left[node] : right[node];
      }
      acc += p1[node];
    }
    out[r] = acc / n_trees > 0.5 ? 1 : 0;                  // argmax of the mean class probabilities (ties: class 0)
  }
  return 0;
}
