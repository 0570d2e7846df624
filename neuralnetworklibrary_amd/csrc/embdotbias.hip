// K4 — fused EmbeddingDotBias forward / backward (CollabFilterNet, Applications/CollabFiltering.py:196-204).
//
// HBM/gather-bound: per sample fwd reads 16 B of indices + 2*D*4 B of rows + 8 B of biases and writes
// 4(+4) B; bwd re-reads the two rows and scatter-adds 2*D*4 + 8 B (516 B/sample at D=30).
// Layout: a 16-lane sub-group of a wave owns one sample (4 samples per 64-lane wave); lanes stride the
// embedding dimension so each sub-group reads its two rows as contiguous 64-B segments, and the dot
// product is reduced with 4 xor-shuffles inside the sub-group (no LDS).
#include "scatter_det.h"

namespace {

constexpr int kSub = 16;          // lanes per sample
constexpr int kBlock = 256;       // 4 waves, 16 samples per block

__device__ __forceinline__ float sub_sum(float v) {
#pragma unroll
  for (int o = kSub / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(kBlock) void embdotbias_fwd_kernel(
    const int64_t* __restrict__ x, const float* __restrict__ U, const float* __restrict__ M,
    const float* __restrict__ bu, const float* __restrict__ bi, float* __restrict__ y,
    float* __restrict__ z, int64_t n, int64_t n_user, int64_t n_item, int D, int has_range, float lo,
    float hi, int32_t* __restrict__ err_flag) {
  const int sub = threadIdx.x & (kSub - 1);
  const int64_t per_block = kBlock / kSub;
  for (int64_t b = (int64_t)blockIdx.x * per_block + (threadIdx.x / kSub); b < n;
       b += (int64_t)gridDim.x * per_block) {
    const int64_t u = x[2 * b], it = x[2 * b + 1];
    const bool ok = (u >= 0) & (u < n_user) & (it >= 0) & (it < n_item);
    float acc = 0.f;
    if (ok) {
      const float* __restrict__ ur = U + u * D;
      const float* __restrict__ mr = M + it * D;
      for (int d = sub; d < D; d += kSub) acc = fmaf(ur[d], mr[d], acc);
    }
    acc = sub_sum(acc);
    if (sub == 0) {
      if (!ok) {
        if (err_flag) *err_flag = 1;
        y[b] = 0.f;
        if (z) z[b] = 0.f;
      } else {
        const float zz = acc + bu[u] + bi[it];
        if (z) z[b] = zz;
        y[b] = has_range ? lo + (hi - lo) * (1.f / (1.f + expf(-zz))) : zz;
      }
    }
  }
}

__global__ __launch_bounds__(kBlock) void embdotbias_bwd_kernel(
    const int64_t* __restrict__ x, const float* __restrict__ U, const float* __restrict__ M,
    const float* __restrict__ z, const float* __restrict__ dy, float* __restrict__ dU,
    float* __restrict__ dM, float* __restrict__ dbu, float* __restrict__ dbi, int64_t n,
    int64_t n_user, int64_t n_item, int D, int has_range, float lo, float hi) {
  const int sub = threadIdx.x & (kSub - 1);
  const int64_t per_block = kBlock / kSub;
  for (int64_t b = (int64_t)blockIdx.x * per_block + (threadIdx.x / kSub); b < n;
       b += (int64_t)gridDim.x * per_block) {
    const int64_t u = x[2 * b], it = x[2 * b + 1];
    if ((u < 0) | (u >= n_user) | (it < 0) | (it >= n_item)) continue;
    float g = dy[b];
    if (has_range) {
      const float s = 1.f / (1.f + expf(-z[b]));
      g *= (hi - lo) * s * (1.f - s);
    }
    const float* __restrict__ ur = U + u * D;
    const float* __restrict__ mr = M + it * D;
    float* __restrict__ dur = dU + u * D;
    float* __restrict__ dmr = dM + it * D;
    for (int d = sub; d < D; d += kSub) {
      atomicAdd(dur + d, g * mr[d]);
      atomicAdd(dmr + d, g * ur[d]);
    }
    if (sub == 0) {
      atomicAdd(dbu + u, g);
      atomicAdd(dbi + it, g);
    }
  }
}

// g[b] = d loss / d (pre-sigmoid score) of sample b (0 for a sample with an out-of-range index)
__global__ void embdot_g_kernel(const int64_t* __restrict__ x, const float* __restrict__ z, const float* __restrict__ dy,
                                float* __restrict__ g, int64_t n, int64_t n_user, int64_t n_item, int has_range, float lo, float hi) {
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n; b += (int64_t)gridDim.x * blockDim.x) {
    const int64_t u = x[2 * b], it = x[2 * b + 1];
    float v = 0.f;
    if (!((u < 0) | (u >= n_user) | (it < 0) | (it >= n_item))) {
      v = dy[b];
      if (has_range) {
        const float s = 1.f / (1.f + expf(-z[b]));
        v *= (hi - lo) * s * (1.f - s);
      }
    }
    g[b] = v;
  }
}


// Small minibatches (n <= 256: the MovieLens-100K notebook's 64) — the whole backward in ONE launch, no sort, no zero fill (round 4):
// one thread per element of the contiguous gradient block [dU | dM | dbu | dbi] scans the minibatch (indices and g staged in LDS)
// in SAMPLE ORDER and adds g * partner row on a match.  Replaces memset + g + rank sort + segment sums (4 of the 11 nodes of the
// replayed collab step); same summation order as the sorted path for rows with few samples.
constexpr int kScanMaxN = 256;

__global__ __launch_bounds__(256) void embdot_scan_bwd_kernel(const int64_t* __restrict__ x, const float* __restrict__ U,
                                                              const float* __restrict__ M, const float* __restrict__ z,
                                                              const float* __restrict__ dy, float* __restrict__ out, int n, long n_user,
                                                              long n_item, int D, int has_range, float lo, float hi) {
  __shared__ int su[kScanMaxN], si[kScanMaxN];
  __shared__ float sg[kScanMaxN];
  const int t = threadIdx.x;
  if (t < n) {
    const int64_t u = x[2 * t], it = x[2 * t + 1];
    const bool ok = !((u < 0) | (u >= n_user) | (it < 0) | (it >= n_item));
    float g = 0.f;
    if (ok) {
      g = dy[t];
      if (has_range) {
        const float sgm = 1.f / (1.f + expf(-z[t]));
        g *= (hi - lo) * sgm * (1.f - sgm);
      }
    }
    su[t] = ok ? (int)u : -1;
    si[t] = ok ? (int)it : -1;
    sg[t] = g;
  }
  __syncthreads();
  const long nu_d = n_user * D, ni_d = n_item * D;
  const long total = nu_d + ni_d + n_user + n_item;
  const long f = (long)blockIdx.x * 256 + t;
  if (f >= total) return;
  float acc = 0.f;
  if (f < nu_d + ni_d) {
    const bool user = f < nu_d;
    const long ff = user ? f : f - nu_d;
    const int r = (int)(ff / D), d = (int)(ff - (long)r * D);
    const int* mine = user ? su : si;
    const int* other = user ? si : su;
    const float* P = user ? M : U;
    for (int s = 0; s < n; ++s)
      if (mine[s] == r) acc += sg[s] * P[(long)other[s] * D + d];
  } else {
    const bool user = f < nu_d + ni_d + n_user;
    const int r = (int)(user ? f - nu_d - ni_d : f - nu_d - ni_d - n_user);
    const int* mine = user ? su : si;
    for (int s = 0; s < n; ++s)
      if (mine[s] == r) acc += sg[s];
  }
  out[f] = acc;
}

int grid_for(int64_t n) {
  int64_t blocks = nnl_cdiv(n, kBlock / kSub);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

}  // namespace

extern "C" int nnl_embdotbias_fwd(const int64_t* x, const float* U, const float* M, const float* bu,
                                  const float* bi, float* y, float* z, int64_t n, int64_t n_user,
                                  int64_t n_item, int64_t D, int has_range, float lo, float hi,
                                  int32_t* err_flag, void* stream) {
  NNL_CHECK_ARG(n >= 0 && n_user > 0 && n_item > 0 && D > 0 && D < (1 << 30), "embdotbias_fwd: bad sizes");
  if (n == 0) return NNL_OK;
  NNL_CHECK_ARG(x && U && M && bu && bi && y, "embdotbias_fwd: null pointer");
  hipStream_t s = (hipStream_t)stream;
  NnlProfScope prof(NNL_PROF_EMBDOT, s, (double)n * (16 + 8.0 * D + 8 + 8));
  hipLaunchKernelGGL(embdotbias_fwd_kernel, dim3(grid_for(n)), dim3(kBlock), 0, s, x, U, M, bu, bi, y, z,
                     n, n_user, n_item, (int)D, has_range, lo, hi, err_flag);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

extern "C" size_t nnl_embdotbias_bwd_workspace_bytes(int64_t n) {
  return n > 0 ? nnl_det::order_bytes(n, 2) + (size_t)n * sizeof(float) : 0;
}

extern "C" int nnl_embdotbias_bwd(const int64_t* x, const float* U, const float* M, const float* z,
                                  const float* dy, float* dU, float* dM, float* dbu, float* dbi,
                                  int64_t n, int64_t n_user, int64_t n_item, int64_t D, int has_range,
                                  float lo, float hi, void* workspace, size_t workspace_bytes, void* stream) {
  NNL_CHECK_ARG(n >= 0 && n_user > 0 && n_item > 0 && D > 0 && D < (1 << 30), "embdotbias_bwd: bad sizes");
  NNL_CHECK_ARG(dU && dM && dbu && dbi, "embdotbias_bwd: null output");
  hipStream_t s = (hipStream_t)stream;
  const bool one_block = dM == dU + n_user * D && dbu == dM + n_item * D && dbi == dbu + n_user;
  // (the scan costs table elements x minibatch compares: fine at MovieLens cardinalities — 81 k x 64 — but milliseconds for tables
  // of 1e5 - 1e6 rows, ADVICE r4: above 2^27 element-sample pairs the sorted scatter below is the path)
  if (one_block && n >= 1 && n <= kScanMaxN && (n_user + n_item) * (D + 1) < (1L << 31) && (n_user + n_item) * (D + 1) * n <= (1L << 27) &&
      NNL_ENV_INT("NNL_EMBDOT_SCAN", 1) != 0 &&
      NNL_ENV_INT("NNL_SCATTER_ATOMIC", 0) == 0) {
    NNL_CHECK_ARG(x && U && M && dy && (z || !has_range), "embdotbias_bwd: null pointer");
    NnlProfScope prof(NNL_PROF_EMBDOT, s, (double)n * (16 + 16.0 * D + 8 + 8));
    const long total = (n_user + n_item) * (D + 1);
    hipLaunchKernelGGL(embdot_scan_bwd_kernel, dim3((unsigned)nnl_cdiv(total, 256)), dim3(256), 0, s, x, U, M, z, dy, dU, (int)n, (long)n_user,
                       (long)n_item, (int)D, has_range, lo, hi);
    NNL_CHECK_LAUNCH();
    return NNL_OK;
  }
  if (one_block) {
    // the four gradients are one allocation ([dU | dM | dbu | dbi], as ops.py hands them over): one fill instead of four
    NNL_CHECK_HIP(hipMemsetAsync(dU, 0, sizeof(float) * ((n_user + n_item) * (D + 1)), s));
  } else {
    NNL_CHECK_HIP(hipMemsetAsync(dU, 0, sizeof(float) * n_user * D, s));
    NNL_CHECK_HIP(hipMemsetAsync(dM, 0, sizeof(float) * n_item * D, s));
    NNL_CHECK_HIP(hipMemsetAsync(dbu, 0, sizeof(float) * n_user, s));
    NNL_CHECK_HIP(hipMemsetAsync(dbi, 0, sizeof(float) * n_item, s));
  }
  if (n == 0) return NNL_OK;
  NNL_CHECK_ARG(x && U && M && dy && (z || !has_range), "embdotbias_bwd: null pointer");
  NnlProfScope prof(NNL_PROF_EMBDOT, s, (double)n * (16 + 16.0 * D + 8 + 8));
  if (nnl_det::use_det(n, workspace) && workspace_bytes >= nnl_embdotbias_bwd_workspace_bytes(n)) {
    // deterministic: samples of a table row are added in sample order (scatter_det.h)
    int* order = (int*)workspace;
    float* g = (float*)((char*)workspace + nnl_det::order_bytes(n, 2));
    hipLaunchKernelGGL(embdot_g_kernel, dim3(grid_for(n)), dim3(kBlock), 0, s, x, z, dy, g, n, n_user, n_item, has_range, lo, hi);
    NNL_CHECK_LAUNCH();
    int st = nnl_det::sort_rows(x, 2, n, 2, order, s);
    if (st) return st;
    nnl_det::SegSumParams q{};
    q.idx = x; q.idx_stride = 2; q.n = (int)n; q.scale_i = g; q.scale_i_stride = 0; q.skip_row = -1;
    q.srcrow = x; q.srcrow_stride = 2;
    // dU[u] = sum g * M[item];  dM[item] = sum g * U[u];  dbu[u] = sum g;  dbi[item] = sum g — ONE launch, four parameter sets
    // (the "column" of a set is selected by offsetting idx / order)
    nnl_det::SegSumParams4 ps{};
    q.order = order; q.card = n_user; q.D = (int)D; q.dst = dU; q.src = M; q.ld = D; q.srcrow_col = 1; q.srcrow_card = n_item;
    ps.q[0] = q;
    nnl_det::SegSumParams r = q;
    r.idx = x + 1; r.order = order + n; r.card = n_item; r.dst = dM; r.src = U; r.srcrow = x; r.srcrow_col = 0; r.srcrow_card = n_user;
    ps.q[1] = r;
    nnl_det::SegSumParams b = q;
    b.D = 1; b.dst = dbu; b.src = g; b.ld = 1; b.srcrow = nullptr; b.srcrow_card = 0; b.scale_i = nullptr;
    ps.q[2] = b;
    b.idx = x + 1; b.order = order + n; b.card = n_item; b.dst = dbi;
    ps.q[3] = b;
    return nnl_det::segsum4(ps, 4, s);
  }
  hipLaunchKernelGGL(embdotbias_bwd_kernel, dim3(grid_for(n)), dim3(kBlock), 0, s, x, U, M, z, dy, dU, dM,
                     dbu, dbi, n, n_user, n_item, (int)D, has_range, lo, hi);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}
