// C-ABI glue shared by all kernels: error string, version, optional per-GEMM HIP-event timing.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <array>
#include <map>
#include <mutex>
#include <string>
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

#include <atomic>
static std::atomic<long> g_launches{0};
void llmseg_count_launch() { g_launches.fetch_add(1, std::memory_order_relaxed); }
extern "C" int64_t llmseg_launch_count(void) { return (int64_t)g_launches.load(std::memory_order_relaxed); }
extern "C" int llmseg_version(void) { return LLMSEG_ABI_VERSION; }
extern "C" int64_t llmseg_struct_size(int which) {
  switch (which) {
    case 0: return (int64_t)sizeof(llmseg_gemm_args);
    case 1: return (int64_t)sizeof(llmseg_attn_args);
    case 2: return (int64_t)sizeof(llmseg_attn_bwd_args);
    case 3: return (int64_t)sizeof(llmseg_dropout);
    default: return -1;
  }
}

// ---- GEMM timing: one (start, stop) event pair per launch, recorded on the launch stream -----------------------
namespace {
struct ProfRec { hipEvent_t a, b; double flops; long tag[4]; };
std::mutex g_mu;
bool g_prof_on = false;
std::vector<ProfRec> g_recs;
std::vector<ProfRec> g_pool;
hipEvent_t g_cur = nullptr;
char g_dom_name[128] = "";
double g_dom_bytes = 0;          // algorithmic bytes (A + W + C once, bf16) of the dominant class in the last collected window
int64_t g_dom_info[4] = {0, 0, 0, 0};   // of that class: user-level GEMM calls, calls run as K-slices, K-slices summed over those, kernel launches
}  // namespace

void llmseg_prof_begin(hipStream_t s) {
  if (!g_prof_on) return;
  std::lock_guard<std::mutex> lk(g_mu);
  ProfRec r;
  if (!g_pool.empty()) { r = g_pool.back(); g_pool.pop_back(); }
  else { (void)hipEventCreate(&r.a); (void)hipEventCreate(&r.b); }
  r.flops = 0;
  (void)hipEventRecord(r.a, s);
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
  (void)hipEventRecord(g_recs.back().b, s);
}

extern "C" int llmseg_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_prof_on = on != 0;
  return LLMSEG_OK;
}

extern "C" const char* llmseg_prof_dominant_kernel(void) { return g_dom_name; }
extern "C" double llmseg_prof_dominant_bytes(void) { return g_dom_bytes; }
extern "C" int llmseg_prof_dominant_info(int64_t* out4) {
  if (!out4) return LLMSEG_EINVAL;
  for (int i = 0; i < 4; ++i) out4[i] = g_dom_info[i];
  return LLMSEG_OK;
}

extern "C" int llmseg_prof_collect(double* total_ms, double* total_flops, int64_t* launches, double* dom_ms, double* dom_flops, int64_t* dom_launches) {
  std::lock_guard<std::mutex> lk(g_mu);
  double ms = 0, fl = 0;
  std::map<long, std::array<double, 4>> by_class;                      // kernel class (staging variant, fp32-out) -> {ms, flops, count, bytes}
  std::map<std::array<long, 4>, std::array<double, 3>> by_shape;      // (M,N,K,variant) -> {ms, flops, count}
  std::map<long, std::array<int64_t, 3>> split_by_class;              // class -> {calls run as K-slices, their slices summed, calls with the extension K-tile}
  for (auto& r : g_recs) {
    (void)hipEventSynchronize(r.b);
    float t = 0;
    (void)hipEventElapsedTime(&t, r.a, r.b);
    ms += t;
    fl += r.flops;
    auto& c = by_class[((r.tag[3] % 10000) / 1000) * 2 + (r.tag[3] & 1)];    // tag = split * 10000 + variant * 1000 + flags
    c[0] += t; c[1] += r.flops; c[2] += 1;
    {
      const long S = r.tag[3] / 10000;                                      // K-slices of this call (0 / 1 = none)
      auto& sp = split_by_class[((r.tag[3] % 10000) / 1000) * 2 + (r.tag[3] & 1)];
      if (S > 1) { sp[0] += 1; sp[1] += S; }
    }
    c[3] += 2.0 * ((double)r.tag[0] * r.tag[2] + (double)r.tag[1] * r.tag[2] + (double)r.tag[0] * r.tag[1]);      // M K + N K + M N elements, 2 bytes each
    auto& e = by_shape[{r.tag[0], r.tag[1], r.tag[2], r.tag[3]}];
    e[0] += t; e[1] += r.flops; e[2] += 1;
    g_pool.push_back(r);
  }
  if (getenv("LLMSEG_PROF_TABLE")) {                                   // per-shape table on stderr (tuning aid)
    std::vector<std::pair<double, std::array<long, 4>>> order;
    for (auto& kv : by_shape) order.push_back({-kv.second[0], kv.first});
    std::sort(order.begin(), order.end());
    fprintf(stderr, "%8s %8s %8s %6s %8s %10s %9s %6s\n", "M", "N", "K", "var", "calls", "total_ms", "TFLOP/s", "%time");
    const size_t top = (size_t)std::max(40, atoi(getenv("LLMSEG_PROF_TABLE")));
    for (size_t i = 0; i < order.size() && i < top; ++i) {
      auto& e = by_shape[order[i].second];
      fprintf(stderr, "%8ld %8ld %8ld %6ld %8.0f %10.2f %9.1f %6.1f\n", order[i].second[0], order[i].second[1], order[i].second[2], order[i].second[3],
              e[2], e[0], e[1] / (e[0] * 1e-3) / 1e12, 100.0 * e[0] / ms);
    }
  }
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = fl;
  if (launches) *launches = (int64_t)g_recs.size();
  // dominant kernel = the GEMM kernel class with the largest total time in this window
  long dom = -1;
  for (auto& kv : by_class) if (dom < 0 || kv.second[0] > by_class[dom][0]) dom = kv.first;
  static const char* names[] = {"gemm_bf16_tn_kernel<*, ...> (register staging, 128x128)", "?", "gemm_bf16_tn_glds_kernel<*, 2, 1>", "gemm_skinny_kernel<M>", "?", "?", "?", "gemm_bf16_tn_pp2_kernel<*, false>",
                                "gemm_bf16_tn_pp_kernel<*, false, 4>", "gemm_bf16_tn_pp_kernel<*, false, 2>"};
  if (dom >= 0) {
    const long v = dom / 2;
    std::string nm = (v >= 0 && v < 10) ? names[v] : "?";
    const size_t star = nm.find('*');
    if (star != std::string::npos) nm.replace(star, 1, (dom & 1) ? "true" : "false");
    snprintf(g_dom_name, sizeof(g_dom_name), "%s", nm.c_str());
  } else g_dom_name[0] = 0;
  if (dom_ms) *dom_ms = dom >= 0 ? by_class[dom][0] : 0;
  if (dom_flops) *dom_flops = dom >= 0 ? by_class[dom][1] : 0;
  if (dom_launches) *dom_launches = dom >= 0 ? (int64_t)by_class[dom][2] : 0;
  g_dom_bytes = dom >= 0 ? by_class[dom][3] : 0;
  if (dom >= 0) {
    // a record is one user-level GEMM CALL: one launch of the tile kernel (the K-slices are its batch index), plus one splitk_reduce_kernel
    // launch when it ran as K-slices -- what rocprofv3 lists as separate kernel rows
    const auto& sp = split_by_class[dom];
    g_dom_info[0] = (int64_t)by_class[dom][2]; g_dom_info[1] = sp[0]; g_dom_info[2] = sp[1]; g_dom_info[3] = (int64_t)by_class[dom][2] + sp[0];
  } else g_dom_info[0] = g_dom_info[1] = g_dom_info[2] = g_dom_info[3] = 0;
  g_recs.clear();
  return LLMSEG_OK;
}
