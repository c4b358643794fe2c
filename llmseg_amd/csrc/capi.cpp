// C-ABI glue shared by all kernels: error string, version, optional per-GEMM HIP-event timing.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <array>
#include <map>
#include <mutex>
#include <vector>
#include "llmseg_hip.h"

static thread_local char g_err[512] = "";

void llmseg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* llmseg_last_error(void) { return g_err; }
extern "C" int llmseg_version(void) { return 1; }

// ---- GEMM timing: one (start, stop) event pair per launch, recorded on the launch stream -----------------------
namespace {
struct ProfRec { hipEvent_t a, b; double flops; long tag[4]; };
std::mutex g_mu;
bool g_prof_on = false;
std::vector<ProfRec> g_recs;
std::vector<ProfRec> g_pool;
hipEvent_t g_cur = nullptr;
}  // namespace

void llmseg_prof_begin(hipStream_t s) {
  if (!g_prof_on) return;
  std::lock_guard<std::mutex> lk(g_mu);
  ProfRec r;
  if (!g_pool.empty()) { r = g_pool.back(); g_pool.pop_back(); }
  else { hipEventCreate(&r.a); hipEventCreate(&r.b); }
  r.flops = 0;
  hipEventRecord(r.a, s);
  g_recs.push_back(r);
}

void llmseg_prof_tag(long a, long b, long c, long d) {
  if (!g_prof_on) return;
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_recs.empty()) return;
  g_recs.back().tag[0] = a; g_recs.back().tag[1] = b; g_recs.back().tag[2] = c; g_recs.back().tag[3] = d;
}

void llmseg_prof_end(hipStream_t s, double flops) {
  if (!g_prof_on) return;
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_recs.empty()) return;
  g_recs.back().flops = flops;
  hipEventRecord(g_recs.back().b, s);
}

extern "C" int llmseg_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_prof_on = on != 0;
  return LLMSEG_OK;
}

extern "C" int llmseg_prof_collect(double* total_ms, double* total_flops, int64_t* launches, double* dom_ms, double* dom_flops, int64_t* dom_launches) {
  std::lock_guard<std::mutex> lk(g_mu);
  double ms = 0, fl = 0, dms = 0, dfl = 0;
  int64_t dn = 0;
  std::map<std::array<long, 4>, std::array<double, 3>> by_shape;      // (M,N,K,variant) -> {ms, flops, count}
  for (auto& r : g_recs) {
    hipEventSynchronize(r.b);
    float t = 0;
    hipEventElapsedTime(&t, r.a, r.b);
    ms += t;
    fl += r.flops;
    if (r.tag[3] / 1000 == 2 && (r.tag[3] & 1) == 0) { dms += t; dfl += r.flops; ++dn; }   // gemm_bf16_tn_glds_kernel<false, 2, 1>
    auto& e = by_shape[{r.tag[0], r.tag[1], r.tag[2], r.tag[3]}];
    e[0] += t; e[1] += r.flops; e[2] += 1;
    g_pool.push_back(r);
  }
  if (getenv("LLMSEG_PROF_TABLE")) {                                   // per-shape table on stderr (tuning aid)
    std::vector<std::pair<double, std::array<long, 4>>> order;
    for (auto& kv : by_shape) order.push_back({-kv.second[0], kv.first});
    std::sort(order.begin(), order.end());
    fprintf(stderr, "%8s %8s %8s %4s %8s %10s %9s %6s\n", "M", "N", "K", "var", "calls", "total_ms", "TFLOP/s", "%time");
    for (size_t i = 0; i < order.size() && i < 40; ++i) {
      auto& e = by_shape[order[i].second];
      fprintf(stderr, "%8ld %8ld %8ld %4ld %8.0f %10.2f %9.1f %6.1f\n", order[i].second[0], order[i].second[1], order[i].second[2], order[i].second[3],
              e[2], e[0], e[1] / (e[0] * 1e-3) / 1e12, 100.0 * e[0] / ms);
    }
  }
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = fl;
  if (launches) *launches = (int64_t)g_recs.size();
  if (dom_ms) *dom_ms = dms;
  if (dom_flops) *dom_flops = dfl;
  if (dom_launches) *dom_launches = dn;
  g_recs.clear();
  return LLMSEG_OK;
}
