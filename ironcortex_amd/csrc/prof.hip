// Live per-kernel timing for bench.py: HIP events recorded on the stream a kernel is launched
// on, read back (with a synchronise) by ic_prof_read().  Off by default; when off the launch
// path pays one predictable branch.
#include <mutex>
#include <vector>

#include "ic_common.h"

bool g_ic_prof_on = false;
int g_ic_prof_only = -1;  // -1: every kernel; otherwise only this kernel id is bracketed
static int g_stride = 1;  // bracket 1 launch in g_stride (of the launches that qualify)
static unsigned g_tick = 0;

// Algorithmic bytes of the launches whose traffic follows from their shapes alone (GEMV chains, heads,
// gate): counted per kernel id on the host at launch time while profiling is on, every launch
// (independent of the event sampling), read back with ic_prof_bytes_read().
static int64_t g_nlaunch[IC_KID_COUNT] = {};
static double g_bytes[IC_KID_COUNT] = {};
void ic_prof_count(int kid, double bytes) {
  g_nlaunch[kid] += 1;
  g_bytes[kid] += bytes;
}

bool ic_prof_sampled() { return g_stride <= 1 || (g_tick++ % (unsigned)g_stride) == 0; }

namespace {
struct Pair {
  hipEvent_t a, b;
};
std::mutex g_mu;
std::vector<Pair> g_open[IC_KID_COUNT];   // recorded, not yet read
std::vector<Pair> g_free;                 // event pool
Pair g_cur[IC_KID_COUNT];
const char* const g_names[] = IC_KID_NAMES;
}  // namespace

void ic_prof_begin(int kid, hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_mu);
  Pair p;
  if (!g_free.empty()) {
    p = g_free.back();
    g_free.pop_back();
  } else {
    if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return;
  }
  g_cur[kid] = p;
  (void)hipEventRecord(p.a, st);
}

void ic_prof_end(int kid, hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_mu);
  (void)hipEventRecord(g_cur[kid].b, st);
  g_open[kid].push_back(g_cur[kid]);
}

extern "C" int ic_prof_enable(int on) {
  g_ic_prof_on = on != 0;
  return IC_OK;
}
extern "C" int ic_prof_select(int kid) {
  IC_REQUIRE(kid >= -1 && kid < IC_KID_COUNT, "ic_prof_select: kid=%d", kid);
  g_ic_prof_only = kid;
  return IC_OK;
}
extern "C" int ic_prof_stride(int n) {
  IC_REQUIRE(n >= 1, "ic_prof_stride: n=%d", n);
  g_stride = n;
  g_tick = 0;
  return IC_OK;
}
extern "C" int ic_prof_kernel_count(void) { return IC_KID_COUNT; }
extern "C" const char* ic_prof_kernel_name(int kid) {
  return (kid >= 0 && kid < IC_KID_COUNT) ? g_names[kid] : "";
}
extern "C" int ic_prof_read(int kid, int64_t* launches, double* total_ms) {
  IC_REQUIRE(kid >= 0 && kid < IC_KID_COUNT && launches && total_ms, "ic_prof_read: bad argument");
  std::lock_guard<std::mutex> lk(g_mu);
  double ms = 0.0;
  int64_t n = 0;
  for (Pair& p : g_open[kid]) {
    float t = 0.f;
    if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&t, p.a, p.b) == hipSuccess) {
      ms += t;
      ++n;
    }
    g_free.push_back(p);
  }
  g_open[kid].clear();
  *launches = n;
  *total_ms = ms;
  return IC_OK;
}
extern "C" int ic_prof_bytes_read(int kid, int64_t* launches, double* bytes) {
  IC_REQUIRE(kid >= 0 && kid < IC_KID_COUNT && launches && bytes, "ic_prof_bytes_read: bad argument");
  std::lock_guard<std::mutex> lk(g_mu);
  *launches = g_nlaunch[kid];
  *bytes = g_bytes[kid];
  g_nlaunch[kid] = 0;
  g_bytes[kid] = 0.0;
  return IC_OK;
}
