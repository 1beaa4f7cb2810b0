// Host-side glue of libmjv_hip.so: error reporting, launch checking and the opt-in event profiler.
#include "mjv_common.h"

#include <atomic>
#include <mutex>
#include <string>
#include <vector>
#include <string.h>

namespace {
thread_local char g_err[512] = "";

struct ProfTag {
  std::string name;
  int64_t launches = 0;
  double ms = 0, flops = 0, bytes = 0;
};
struct ProfRec {
  int tag;
  hipEvent_t start, stop;
};
std::mutex g_mu;
bool g_prof_on = false;
std::string g_prof_only;  // when non-empty only this tag is recorded
std::vector<ProfTag> g_tags;
std::vector<ProfRec> g_recs;
std::vector<hipEvent_t> g_pool;

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

void mjv_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int mjv_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    mjv_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return MJV_E_LAUNCH;
  }
  return MJV_OK;
}

int mjv_device_cus() {
  static std::atomic<int> cus[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  int n = cus[dev & 63].load(std::memory_order_relaxed);
  if (n <= 0) {
    hipDeviceProp_t prop;
    n = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    cus[dev & 63].store(n, std::memory_order_relaxed);
  }
  return n;
}

MjvProfScope::MjvProfScope(const char* tag, hipStream_t s, double flops, double bytes) : slot(-1), stream(s) {
  if (!g_prof_on) return;
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_prof_only.empty() && g_prof_only != tag) return;
  int t = -1;
  for (size_t i = 0; i < g_tags.size(); ++i)
    if (g_tags[i].name == tag) { t = (int)i; break; }
  if (t < 0) {
    g_tags.push_back(ProfTag());
    g_tags.back().name = tag;
    t = (int)g_tags.size() - 1;
  }
  g_tags[t].launches += 1;
  g_tags[t].flops += flops;
  g_tags[t].bytes += bytes;
  ProfRec r;
  r.tag = t;
  r.start = get_event();
  r.stop = get_event();
  (void)hipEventRecord(r.start, s);
  g_recs.push_back(r);
  slot = (int)g_recs.size() - 1;
}

MjvProfScope::~MjvProfScope() {
  if (slot < 0) return;
  std::lock_guard<std::mutex> lk(g_mu);
  (void)hipEventRecord(g_recs[slot].stop, stream);
}

extern "C" {

int mjv_abi_version(void) { return MJV_ABI_VERSION; }
const char* mjv_last_error(void) { return g_err; }
const char* mjv_arch(void) { return "gfx950"; }

int mjv_prof_enable(int32_t on) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_prof_on = on != 0;
  return MJV_OK;
}

int mjv_prof_filter(const char* tag) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_prof_only = tag ? tag : "";
  return MJV_OK;
}

int mjv_prof_collect(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto& r : g_recs) {
    (void)hipEventSynchronize(r.stop);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.start, r.stop) == hipSuccess) g_tags[r.tag].ms += ms;
    g_pool.push_back(r.start);
    g_pool.push_back(r.stop);
  }
  g_recs.clear();
  return MJV_OK;
}

int mjv_prof_reset(void) {
  mjv_prof_collect();
  std::lock_guard<std::mutex> lk(g_mu);
  g_tags.clear();
  return MJV_OK;
}

int mjv_prof_count(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  return (int)g_tags.size();
}

int mjv_prof_get(int32_t i, const char** name, int64_t* launches, double* ms, double* flops, double* bytes) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (i < 0 || i >= (int)g_tags.size()) {
    mjv_set_error("prof_get: index %d out of range", i);
    return MJV_E_ARG;
  }
  if (name) *name = g_tags[i].name.c_str();
  if (launches) *launches = g_tags[i].launches;
  if (ms) *ms = g_tags[i].ms;
  if (flops) *flops = g_tags[i].flops;
  if (bytes) *bytes = g_tags[i].bytes;
  return MJV_OK;
}

}  // extern "C"
