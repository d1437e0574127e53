// In-library kernel profiler: HIP events recorded on the launch stream around the hot kernels
// (torch.cuda.Event would only see torch's current stream; these see the stream the kernel really runs on).
// Enabled by jv_profile_enable(1); jv_profile_report() synchronises and returns per-kernel totals as JSON.
#include <map>
#include <mutex>
#include <sstream>
#include <string.h>
#include <vector>

#include "../../include/jyutvoice_hip.h"
#include "jv_common.h"

namespace jv {

namespace {
struct Span {
  hipEvent_t a, b;
  const char* name;
  const char* group;      // null, or a stage tag set by the caller around a run of launches (prof_group)
  double flops, bytes;
};
struct Prof {
  bool on = false;
  std::vector<Span> spans;
  std::vector<hipEvent_t> pool;
  hipEvent_t pending = nullptr;
  const char* group = nullptr;
  std::mutex mu;
};
Prof g_prof;

hipEvent_t get_event() {
  if (!g_prof.pool.empty()) {
    hipEvent_t e = g_prof.pool.back();
    g_prof.pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
}  // namespace

bool prof_on() { return g_prof.on; }

// launches recorded until the next call are also summed under "_group:<tag>" in the report (tag: a string literal; null: none)
void prof_group(const char* tag) {
  std::lock_guard<std::mutex> lk(g_prof.mu);
  g_prof.group = tag;
}

void prof_begin(hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_prof.mu);
  g_prof.pending = get_event();
  (void)hipEventRecord(g_prof.pending, st);
}

void prof_end(hipStream_t st, const char* name, double flops, double bytes) {
  std::lock_guard<std::mutex> lk(g_prof.mu);
  hipEvent_t b = get_event();
  (void)hipEventRecord(b, st);
  g_prof.spans.push_back(Span{g_prof.pending, b, name, g_prof.group, flops, bytes});
  g_prof.pending = nullptr;
}

}  // namespace jv

extern "C" {

int jv_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(jv::g_prof.mu);
  jv::g_prof.on = on != 0;
  return JV_OK;
}

int jv_profile_report(char* json, int64_t cap) {
  using namespace jv;
  if (!json || cap < 64) return fail(JV_ERR_ARG, "jv_profile_report: buffer too small");
  JV_HIP(hipDeviceSynchronize());
  std::lock_guard<std::mutex> lk(g_prof.mu);
  struct Agg { long n = 0; double ms = 0, flops = 0, bytes = 0; };
  std::map<std::string, Agg> agg;
  for (const Span& s : g_prof.spans) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, s.a, s.b) != hipSuccess) ms = 0.f;
    Agg& a = agg[s.name];
    a.n += 1; a.ms += ms; a.flops += s.flops; a.bytes += s.bytes;
    if (s.group) {
      Agg& gsum = agg[std::string("_group:") + s.group];
      gsum.n += 1; gsum.ms += ms; gsum.flops += s.flops; gsum.bytes += s.bytes;
    }
    g_prof.pool.push_back(s.a);
    g_prof.pool.push_back(s.b);
  }
  g_prof.spans.clear();
  // what an event pair measures with nothing between its two records (the fixed cost every span above includes):
  // reported so that the caller can set the event durations beside a profiler's kernel durations
  double empty_ms = 0.0;
  {
    const int n = 64;
    std::vector<hipEvent_t> ev(2 * n);
    for (auto& e : ev) e = get_event();
    for (int i = 0; i < n; ++i) {
      (void)hipEventRecord(ev[2 * i], nullptr);
      (void)hipEventRecord(ev[2 * i + 1], nullptr);
    }
    (void)hipDeviceSynchronize();
    for (int i = 0; i < n; ++i) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]) == hipSuccess) empty_ms += ms;
    }
    empty_ms /= n;
    for (auto& e : ev) g_prof.pool.push_back(e);
  }
  std::ostringstream os;
  os.precision(9);
  os << "{\"_empty_event_pair\":{\"launches\":1,\"ms\":" << empty_ms << ",\"flops\":0,\"bytes\":0}";
  bool first = false;
  for (const auto& kv : agg) {
    if (!first) os << ",";
    first = false;
    os << "\"" << kv.first << "\":{\"launches\":" << kv.second.n << ",\"ms\":" << kv.second.ms << ",\"flops\":" << kv.second.flops
       << ",\"bytes\":" << kv.second.bytes << "}";
  }
  os << "}";
  const std::string out = os.str();
  if ((int64_t)out.size() + 1 > cap) return fail(JV_ERR_ARG, "jv_profile_report: buffer too small");
  memcpy(json, out.c_str(), out.size() + 1);
  return JV_OK;
}

}  // extern "C"
