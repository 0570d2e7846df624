// Shared helpers for the nnl HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/nnl.h"

#define NNL_WAVE 64

// thread-local last-error text, returned by nnl_last_error()
char* nnl_err_buf();
int nnl_set_error(int code, const char* fmt, ...);

#define NNL_CHECK_ARG(cond, ...)                                           \
  do {                                                                     \
    if (!(cond)) return nnl_set_error(NNL_ERR_INVALID_ARG, __VA_ARGS__);   \
  } while (0)

#define NNL_CHECK_HIP(expr)                                                         \
  do {                                                                              \
    hipError_t _e = (expr);                                                         \
    if (_e != hipSuccess)                                                           \
      return nnl_set_error(NNL_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
  } while (0)

#define NNL_CHECK_LAUNCH()                                                          \
  do {                                                                              \
    hipError_t _e = hipGetLastError();                                              \
    if (_e != hipSuccess)                                                           \
      return nnl_set_error(NNL_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(_e)); \
  } while (0)

static inline int64_t nnl_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Tuning / A-B switches (NNL_* environment variables): read ONCE per site, not per launch (a getenv per call sits on the host's
// launch path).  nnl_reload_env() — exported for tests and the A/B tools — makes every site read its variable again.
int nnl_env_cached(const char* name, int dflt, int* value, int* generation);
int nnl_env_generation();          // bumped by nnl_reload_env(): caches of env-dependent plans key on it
#define NNL_ENV_INT(name, dflt) ([]() -> int { static int v_ = 0, g_ = -1; return nnl_env_cached(name, dflt, &v_, &g_); }())
// A/B hooks of CLOSED experiments (tile / k-block / prefetch / planner-constant sweeps, timing hacks): compiled to their
// shipped value unless the library is built with -DNNL_AB (make AB=1: the tools/bench_*.py --ab sweeps need that build).  The shipped
// library reads only the switches that select between TESTED algorithm families (DESIGN.md section 5 lists them).
#ifdef NNL_AB
#define NNL_AB_INT(name, dflt) NNL_ENV_INT(name, dflt)
#else
#define NNL_AB_INT(name, dflt) (dflt)
#endif

// optional per-launch profiling with HIP events on the launch stream (bench.py roofline leg)
void nnl_prof_begin(int kind, hipStream_t s);
void nnl_prof_end(int kind, hipStream_t s, double work);
void nnl_prof_exec_frac(double f);   // the launch in flight executes f x its algorithmic multiplies (Winograd kernels: 1 / 1.5, 1 / 2.25)

struct NnlProfScope {
  int kind; hipStream_t s; double work;
  NnlProfScope(int k, hipStream_t st, double w) : kind(k), s(st), work(w) { nnl_prof_begin(kind, s); }
  ~NnlProfScope() { nnl_prof_end(kind, s, work); }
};

__device__ __forceinline__ float nnl_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float nnl_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
