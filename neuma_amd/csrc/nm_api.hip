// Error plumbing, version and device self-tests of libneuma_hip.
#include <stdarg.h>

#include "nm_common.h"

static thread_local char g_err[512] = "";

void nm_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* nm_last_error(void) { return g_err; }
extern "C" int nm_version(void) { return 100; }

// ---------------------------------------------------------------- kernel timing
#include <map>
#include <string>
#include <vector>

int g_nm_prof_on = 0;
static std::string g_only;
struct ProfSample { std::string name; hipEvent_t a, b; };
static std::vector<ProfSample> g_samples;
static std::vector<hipEvent_t> g_pool;
static hipEvent_t g_cur = nullptr;
static bool g_cur_active = false;

static hipEvent_t prof_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
static bool prof_match(const char* name) {
  if (g_only.empty()) return true;
  if (*name == '(') ++name;      // (a template instance with a comma is a parenthesised macro argument)
  // template instantiations arrive as "k_material_fwd<NM_ELASTICITY>": match on the prefix
  return strncmp(name, g_only.c_str(), g_only.size()) == 0;
}
static int g_stride = 1;        // time every g_stride-th matching launch (nm_prof_enable(on > 1))
static long g_seen = 0;
void nm_prof_begin(const char* name, hipStream_t s) {
  g_cur_active = false;
  if (!prof_match(name) || g_samples.size() > 2000000) return;
  if (g_stride > 1 && (g_seen++ % g_stride) != 0) return;
  g_cur = prof_event();
  if (!g_cur) return;
  (void)hipEventRecord(g_cur, s);
  g_cur_active = true;
}
void nm_prof_end(const char* name, hipStream_t s) {
  if (!g_cur_active) return;
  hipEvent_t b = prof_event();
  if (!b) return;
  (void)hipEventRecord(b, s);
  g_samples.push_back({name, g_cur, b});
  g_cur_active = false;
}
extern "C" int nm_prof_enable(int32_t on, const char* only_kernel) {
  g_only = only_kernel ? only_kernel : "";
  g_nm_prof_on = on ? 1 : 0;
  g_stride = on > 1 ? on : 1;
  g_seen = 0;
  return NM_OK;
}
extern "C" int nm_prof_reset(void) {
  for (auto& s : g_samples) { g_pool.push_back(s.a); g_pool.push_back(s.b); }
  g_samples.clear();
  return NM_OK;
}
extern "C" int nm_prof_report(char* out, size_t cap) {
  NM_REQUIRE(out && cap > 0, "null report buffer");
  std::map<std::string, std::pair<long, double>> agg;
  for (auto& s : g_samples) {
    NM_HIP_CHECK(hipEventSynchronize(s.b));
    float ms = 0.f;
    NM_HIP_CHECK(hipEventElapsedTime(&ms, s.a, s.b));
    auto& e = agg[s.name];
    e.first += 1;
    e.second += ms;
  }
  std::string txt;
  for (auto& kv : agg) {
    char line[256];
    snprintf(line, sizeof(line), "%s %ld %.6f\n", kv.first.c_str(), kv.second.first, kv.second.second);
    txt += line;
  }
  size_t n = txt.size() < cap - 1 ? txt.size() : cap - 1;
  memcpy(out, txt.data(), n);
  out[n] = 0;
  return NM_OK;
}
