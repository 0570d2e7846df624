// nn.Linear with 1 - 4 output features on [M, K] rows — the last layer of FullyConnectedNet (General/Layers.py:146; the tabular
// regression head `final_lin`: 500 -> 1) — round 4.  On the 64 x 64 MFMA tile a 1-column output is 16 workgroups each wasting
// 63/64 of the matrix core, and its backward was ten launches (channel pads, a filter transpose, dgrad, wgrad + split-K reduce,
// a two-stage bias column sum): 19 us forward + ~60 us backward of a 0.4 ms step.  Here: forward = one wave per row (fixed
// shuffle tree), backward = dX elementwise, dW / db by a fixed-order two-stage column reduction.  HBM/launch-bound:
// 4 (K + N) B per row each way.  Bitwise reproducible (no atomics).
#include "nnl_common.h"

namespace {

constexpr int kMaxN = 4;
constexpr int kSlabRows = 64;      // rows per first-stage block of the weight-gradient reduction

// y[m][n] = bias[n] + sum_k x[m][k] w[n][k]; one wave per row, lanes stride k, xor-shuffle tree
__global__ __launch_bounds__(256) void lin_small_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ y, long M, int K,
                                                            long ldx, int N) {
  const int lane = threadIdx.x & 63;
  const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  float acc[kMaxN] = {0.f, 0.f, 0.f, 0.f};
  const float* xr = x + m * ldx;
  for (int k0 = lane; k0 < K; k0 += 64 * 8) {                // 8 independent loads per lane in flight
    float xv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) xv[u] = k0 + 64 * u < K ? xr[k0 + 64 * u] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = k0 + 64 * u;
      if (k < K) {
#pragma unroll
        for (int n = 0; n < kMaxN; ++n)
          if (n < N) acc[n] += xv[u] * w[(long)n * K + k];
      }
    }
  }
#pragma unroll
  for (int n = 0; n < kMaxN; ++n) {
    float a = acc[n];
    for (int o = 1; o < 64; o <<= 1) a += __shfl_xor(a, o, 64);
    if (lane == 0 && n < N) y[m * N + n] = a + (bias ? bias[n] : 0.f);
  }
}

// dx[m][k] = sum_n dy[m][n] w[n][k]
__global__ __launch_bounds__(256) void lin_small_dx_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                           float* __restrict__ dx, long M, int K, int N) {
  const long total = M * K;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long m = i / K;
    const int k = (int)(i - m * K);
    float a = 0.f;
#pragma unroll
    for (int n = 0; n < kMaxN; ++n)
      if (n < N) a += dy[m * N + n] * w[(long)n * K + k];
    dx[i] = a;
  }
}

// stage 1: part[slab][n][k] = sum over the slab's 64 rows (in row order) of dy[m][n] x[m][k]   (k == K: the bias column, x = 1)
// block = 64 k columns x 4 row groups of 16 rows; the four groups meet in LDS in group order
__global__ __launch_bounds__(256) void lin_small_dw_part_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                float* __restrict__ part, long M, int K, long ldx, int N) {
  __shared__ float red[4][kMaxN][64];
  const int kl = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + kl;
  const long m0 = (long)blockIdx.y * kSlabRows + grp * 16;
  float acc[kMaxN] = {0.f, 0.f, 0.f, 0.f};
  if (k <= K) {
    float xv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) xv[r] = (m0 + r < M) ? (k < K ? x[(m0 + r) * ldx + k] : 1.f) : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (m0 + r < M) {
#pragma unroll
        for (int n = 0; n < kMaxN; ++n)
          if (n < N) acc[n] += dy[(m0 + r) * N + n] * xv[r];
      }
  }
#pragma unroll
  for (int n = 0; n < kMaxN; ++n) red[grp][n][kl] = acc[n];
  __syncthreads();
  if (grp == 0 && k <= K) {
#pragma unroll
    for (int n = 0; n < kMaxN; ++n)
      if (n < N) part[((long)blockIdx.y * N + n) * (K + 1) + k] = ((red[0][n][kl] + red[1][n][kl]) + red[2][n][kl]) + red[3][n][kl];
  }
}

// stage 2: dw[n][k] (k < K) / db[n] (k == K) = the slabs' partial sums in slab order
__global__ __launch_bounds__(256) void lin_small_dw_final_kernel(const float* __restrict__ part, int nslab, float* __restrict__ dw,
                                                                 float* __restrict__ db, int K, int N) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N * (K + 1)) return;
  const int n = i / (K + 1), k = i - n * (K + 1);
  float a = 0.f;
  int s = 0;
  for (; s + 8 <= nslab; s += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[((long)(s + u) * N + n) * (K + 1) + k];
#pragma unroll
    for (int u = 0; u < 8; ++u) a += v[u];
  }
  for (; s < nslab; ++s) a += part[((long)s * N + n) * (K + 1) + k];
  if (k < K) { if (dw) dw[(long)n * K + k] = a; }
  else if (db) db[n] = a;
}

}  // namespace

extern "C" int nnl_linear_small_supported(int64_t N) { return N >= 1 && N <= kMaxN ? 1 : 0; }

extern "C" int nnl_linear_small_fwd(const float* x, const float* w, const float* bias, float* y, int64_t M, int64_t K, int64_t ldx,
                                    int64_t N, void* stream) {
  NNL_CHECK_ARG(M >= 0 && K > 0 && K < (1 << 24) && ldx >= K && N >= 1 && N <= kMaxN, "linear_small_fwd: bad sizes");
  if (M == 0) return NNL_OK;
  NNL_CHECK_ARG(x && w && y, "linear_small_fwd: null pointer");
  hipStream_t s = (hipStream_t)stream;
  NnlProfScope prof(NNL_PROF_ELEMENTWISE, s, 4.0 * M * (double)(K + N));
  hipLaunchKernelGGL(lin_small_fwd_kernel, dim3((unsigned)nnl_cdiv(M, 4)), dim3(256), 0, s, x, w, bias, y, (long)M, (int)K, (long)ldx, (int)N);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

extern "C" size_t nnl_linear_small_bwd_workspace_bytes(int64_t M, int64_t K, int64_t N) {
  if (M <= 0 || K <= 0 || N <= 0) return 0;
  return (size_t)nnl_cdiv(M, kSlabRows) * N * (K + 1) * sizeof(float);
}

/* dx (may be NULL) [M,K] dense, dw (may be NULL) [N,K], db (may be NULL) [N] */
extern "C" int nnl_linear_small_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, float* db, int64_t M,
                                    int64_t K, int64_t ldx, int64_t N, void* workspace, size_t workspace_bytes, void* stream) {
  NNL_CHECK_ARG(M >= 0 && K > 0 && K < (1 << 24) && ldx >= K && N >= 1 && N <= kMaxN, "linear_small_bwd: bad sizes");
  if (M == 0) return NNL_OK;
  NNL_CHECK_ARG(dy && x && w, "linear_small_bwd: null pointer");
  hipStream_t s = (hipStream_t)stream;
  NnlProfScope prof(NNL_PROF_ELEMENTWISE, s, 4.0 * M * (double)(2 * K + N));
  if (dx) {
    long b = nnl_cdiv(M * K, 256);
    if (b > 2048) b = 2048;
    hipLaunchKernelGGL(lin_small_dx_kernel, dim3((unsigned)b), dim3(256), 0, s, dy, w, dx, (long)M, (int)K, (int)N);
    NNL_CHECK_LAUNCH();
  }
  if (dw || db) {
    if (workspace == nullptr || workspace_bytes < nnl_linear_small_bwd_workspace_bytes(M, K, N))
      return nnl_set_error(NNL_ERR_WORKSPACE, "linear_small_bwd: workspace too small");
    const int nslab = (int)nnl_cdiv(M, kSlabRows);
    float* part = (float*)workspace;
    hipLaunchKernelGGL(lin_small_dw_part_kernel, dim3((unsigned)nnl_cdiv(K + 1, 64), (unsigned)nslab), dim3(256), 0, s, dy, x, part, (long)M,
                       (int)K, (long)ldx, (int)N);
    NNL_CHECK_LAUNCH();
    hipLaunchKernelGGL(lin_small_dw_final_kernel, dim3((unsigned)nnl_cdiv(N * (K + 1), 256)), dim3(256), 0, s, (const float*)part, nslab, dw,
                       db, (int)K, (int)N);
    NNL_CHECK_LAUNCH();
  }
  return NNL_OK;
}
