// Error reporting + optional per-launch HIP-event profiling for libnnl_hip.so.
#include "nnl_common.h"
#include <stdarg.h>
#include <mutex>
#include <vector>

static thread_local char g_err[512] = "";

char* nnl_err_buf() { return g_err; }

int nnl_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" int nnl_version(void) { return 200; }

static int g_env_generation = 0;
int nnl_env_cached(const char* name, int dflt, int* value, int* generation) {
  if (*generation != g_env_generation) {
    const char* e = getenv(name);
    *value = e ? atoi(e) : dflt;
    *generation = g_env_generation;
  }
  return *value;
}
extern "C" int nnl_reload_env(void) { ++g_env_generation; return NNL_OK; }
int nnl_env_generation() { return g_env_generation; }
extern "C" const char* nnl_last_error(void) { return g_err; }
#include "stamp.inc"
extern "C" const char* nnl_source_stamp(void) { return NNL_SOURCE_STAMP; }

// ---- profiling: a bounded pool of event pairs, recorded on the stream each kernel is launched on ----
namespace {
struct ProfRec { int kind; hipEvent_t a, b; double work, exec; };
std::mutex g_mu;
bool g_enabled = false;
std::vector<ProfRec> g_recs;
std::vector<hipEvent_t> g_free;
thread_local hipEvent_t g_open_a = nullptr;
thread_local double g_exec_frac = 1.0;      // executed / algorithmic multiplies of the launch in flight (Winograd kernels < 1)
const size_t kMaxRecs = 200000;

hipEvent_t get_event() {
  if (!g_free.empty()) { hipEvent_t e = g_free.back(); g_free.pop_back(); return e; }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
}  // namespace

void nnl_prof_begin(int kind, hipStream_t s) {
  if (!g_enabled) return;
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_recs.size() >= kMaxRecs) { g_open_a = nullptr; return; }
  g_open_a = get_event();
  g_exec_frac = 1.0;
  if (g_open_a) (void)hipEventRecord(g_open_a, s);
}

void nnl_prof_end(int kind, hipStream_t s, double work) {
  if (!g_enabled || !g_open_a) return;
  std::lock_guard<std::mutex> lk(g_mu);
  hipEvent_t b = get_event();
  if (!b) { g_free.push_back(g_open_a); g_open_a = nullptr; return; }
  (void)hipEventRecord(b, s);
  g_recs.push_back({kind, g_open_a, b, work, work * g_exec_frac});
  g_open_a = nullptr;
}

void nnl_prof_exec_frac(double f) { g_exec_frac = f; }

extern "C" int nnl_prof_enable(int enable) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_enabled = enable != 0;
  return NNL_OK;
}

extern "C" int nnl_prof_collect(int64_t* launches, double* total_ms, double* total_work) { return nnl_prof_collect2(launches, total_ms, total_work, nullptr); }

extern "C" int nnl_prof_collect2(int64_t* launches, double* total_ms, double* total_work, double* total_exec) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (int k = 0; k < NNL_PROF_KINDS; ++k) { launches[k] = 0; total_ms[k] = 0; total_work[k] = 0; if (total_exec) total_exec[k] = 0; }
  for (auto& r : g_recs) {
    hipError_t e = hipEventSynchronize(r.b);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, r.a, r.b);
    if (e == hipSuccess && r.kind >= 0 && r.kind < NNL_PROF_KINDS) {
      launches[r.kind] += 1; total_ms[r.kind] += ms; total_work[r.kind] += r.work;
      if (total_exec) total_exec[r.kind] += r.exec;
    }
    g_free.push_back(r.a); g_free.push_back(r.b);
  }
  g_recs.clear();
  return NNL_OK;
}

// ---- data-parallel overlap under hipGraph replay (dist.py: GradSync.reduce_overlapped) -------------------------------------------
// torch forbids external events on ROCm, so a replayed graph cannot record an event another stream waits for.  Instead the captured
// backward contains, right after the last gradient of bucket k has been written, a one-lane SIGNAL kernel that publishes the replay
// counter (a device word the graph itself bumps at its start) into flag[k]; after the replay has been enqueued, the host launches on
// a side stream a one-lane WAIT kernel per bucket that polls flag[k] (relaxed system-scope loads, s_sleep between polls, BOUNDED in
// WALL TIME — the constant-rate counter `wall_clock64()`, not a poll count, so a slow or pre-empted replay cannot trip it: on time-out
// it raises *err and returns, so the stream always drains) followed, in stream order, by the bucket's all-reduce.  The
// gradient bytes are visible to the collective by ordinary kernel-boundary semantics: they were written by kernels that completed
// before the signal kernel started, and the collective's kernels start after the wait kernel ended.
namespace {
__global__ void dp_bump_kernel(int* step) { *step = *step + 1; }
__global__ void dp_signal_kernel(int* flag, const int* step) {
  __hip_atomic_store(flag, *step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void dp_wait_kernel(const int* flag, int value, long long timeout_ticks, int* err) {
  const long long t0 = (long long)wall_clock64();
  for (;;) {
    for (int i = 0; i < 64; ++i) {
      if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - value >= 0) {    // (wrap-safe comparison)
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        return;
      }
      __builtin_amdgcn_s_sleep(32);
    }
    if ((long long)wall_clock64() - t0 > timeout_ticks) break;
  }
  __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace

extern "C" int nnl_dp_bump(int32_t* step, void* stream) {
  NNL_CHECK_ARG(step, "dp_bump: null pointer");
  hipLaunchKernelGGL(dp_bump_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}
extern "C" int nnl_dp_signal(int32_t* flag, const int32_t* step, void* stream) {
  NNL_CHECK_ARG(flag && step, "dp_signal: null pointer");
  hipLaunchKernelGGL(dp_signal_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, flag, step);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}
extern "C" int nnl_dp_wait(const int32_t* flag, int32_t value, int64_t timeout_us, int32_t* err, void* stream) {
  NNL_CHECK_ARG(flag && err && timeout_us > 0, "dp_wait: bad argument");
  static thread_local int rate_khz = 0;                   // wall_clock64() ticks per millisecond (100 000 on gfx950)
  if (rate_khz <= 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || rate_khz <= 0)
      rate_khz = 100000;
  }
  const long long ticks = (long long)((double)timeout_us * 1e-3 * (double)rate_khz);
  hipLaunchKernelGGL(dp_wait_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, flag, value, ticks, err);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}
