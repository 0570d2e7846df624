// nn.MSELoss(reduction='mean') — the 'cont' loss of General/Learner.py:20 (`loss_func_dict['cont']`) that closes the collaborative-
// filtering and structured-data steps.  A 64 - 1024 sample step is launch-bound: ATen runs it as an elementwise
// kernel + a reduction forward and one more backward; here the forward is ONE launch for up to 65 536 samples (a single 1024-thread
// block, fixed-order tree: bitwise reproducible), two above that, and the backward one elementwise launch that takes the upstream
// scalar gradient from device memory (no host sync).
#include "nnl_common.h"

namespace {

constexpr int kLossBlock = 1024;
constexpr long kLossOneBlock = 65536;
constexpr int kLossMaxBlocks = 256;

__device__ __forceinline__ float block_sum(float v, float* red) {
  red[threadIdx.x] = v;
  __syncthreads();
  for (int w = kLossBlock / 2; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  return red[0];
}

// one block: out[0] = scale * sum (a - b)^2;  several blocks: part[blockIdx.x] = its share (scale applied by mse_final_kernel)
__global__ __launch_bounds__(kLossBlock) void mse_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                                              long n, float scale) {
  __shared__ float red[kLossBlock];
  float acc = 0.f;
  for (long i = (long)blockIdx.x * kLossBlock + threadIdx.x; i < n; i += (long)gridDim.x * kLossBlock) {
    const float d = a[i] - b[i];
    acc += d * d;
  }
  const float s = block_sum(acc, red);
  if (threadIdx.x == 0) out[blockIdx.x] = gridDim.x == 1 ? s * scale : s;
}

__global__ __launch_bounds__(kLossBlock) void mse_final_kernel(const float* __restrict__ part, int nparts, float* __restrict__ out, float scale) {
  __shared__ float red[kLossBlock];
  const float s = block_sum((int)threadIdx.x < nparts ? part[threadIdx.x] : 0.f, red);
  if (threadIdx.x == 0) out[0] = s * scale;
}

__global__ __launch_bounds__(256) void mse_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ gout,
                                                       float* __restrict__ da, long n, float scale) {
  const float g = (gout ? gout[0] : 1.f) * scale;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) da[i] = g * (a[i] - b[i]);
}

}  // namespace

extern "C" size_t nnl_mse_workspace_bytes(int64_t n) { return n > kLossOneBlock ? (size_t)kLossMaxBlocks * sizeof(float) : 0; }

extern "C" int nnl_mse_fwd(const float* pred, const float* target, float* loss, int64_t n, void* workspace, size_t workspace_bytes,
                           void* stream) {
  NNL_CHECK_ARG(pred && target && loss && n > 0, "mse_fwd: bad argument");
  hipStream_t s = (hipStream_t)stream;
  NnlProfScope prof(NNL_PROF_ELEMENTWISE, s, 8.0 * n);
  const float scale = 1.f / (float)n;
  if (n <= kLossOneBlock) {
    hipLaunchKernelGGL(mse_fwd_kernel, dim3(1), dim3(kLossBlock), 0, s, pred, target, loss, (long)n, scale);
    NNL_CHECK_LAUNCH();
    return NNL_OK;
  }
  if (workspace == nullptr || workspace_bytes < nnl_mse_workspace_bytes(n)) return nnl_set_error(NNL_ERR_WORKSPACE, "mse_fwd: workspace too small");
  long blocks = nnl_cdiv(n, (long)kLossBlock * 8);
  if (blocks > kLossMaxBlocks) blocks = kLossMaxBlocks;
  hipLaunchKernelGGL(mse_fwd_kernel, dim3((unsigned)blocks), dim3(kLossBlock), 0, s, pred, target, (float*)workspace, (long)n, scale);
  NNL_CHECK_LAUNCH();
  hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(kLossBlock), 0, s, (const float*)workspace, (int)blocks, loss, scale);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

extern "C" int nnl_mse_bwd(const float* pred, const float* target, const float* grad_out, float* dpred, int64_t n, void* stream) {
  NNL_CHECK_ARG(pred && target && dpred && n > 0, "mse_bwd: bad argument");
  hipStream_t s = (hipStream_t)stream;
  NnlProfScope prof(NNL_PROF_ELEMENTWISE, s, 12.0 * n);
  long blocks = nnl_cdiv(n, 256L * 4);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(mse_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, pred, target, grad_out, dpred, (long)n, 2.f / (float)n);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

// ---- FullyConnectedNet's 'sigmoidal' output activation (reference General/Layers.py:150-152): y = lo + (hi - lo) * sigmoid(x) ----
// three ATen kernels forward and three backward on a [bs, 1] tensor; here one each.  The forward also keeps s = sigmoid(x), so the
// backward is torch's own formula dx = dy * (hi - lo) * s * (1 - s) (re-deriving s from y would lose its low bits near lo).
namespace {
__global__ __launch_bounds__(256) void scaled_sigmoid_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ sg, long n,
                                                                  float lo, float hi) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float s = 1.f / (1.f + expf(-x[i]));
    sg[i] = s;
    y[i] = lo + (hi - lo) * s;
  }
}
__global__ __launch_bounds__(256) void scaled_sigmoid_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ sg, float* __restrict__ dx,
                                                                  long n, float lo, float hi) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float s = sg[i];
    dx[i] = (dy[i] * (hi - lo)) * ((1.f - s) * s);
  }
}
}  // namespace

extern "C" int nnl_scaled_sigmoid_fwd(const float* x, float* y, float* sig, int64_t n, float lo, float hi, void* stream) {
  NNL_CHECK_ARG(x && y && sig && n > 0 && hi != lo, "scaled_sigmoid_fwd: bad argument");
  hipStream_t s = (hipStream_t)stream;
  NnlProfScope prof(NNL_PROF_ELEMENTWISE, s, 8.0 * n);
  long blocks = nnl_cdiv(n, 256L * 4);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(scaled_sigmoid_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, y, sig, (long)n, lo, hi);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

extern "C" int nnl_scaled_sigmoid_bwd(const float* dy, const float* sig, float* dx, int64_t n, float lo, float hi, void* stream) {
  NNL_CHECK_ARG(dy && sig && dx && n > 0 && hi != lo, "scaled_sigmoid_bwd: bad argument");
  hipStream_t s = (hipStream_t)stream;
  NnlProfScope prof(NNL_PROF_ELEMENTWISE, s, 12.0 * n);
  long blocks = nnl_cdiv(n, 256L * 4);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(scaled_sigmoid_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dy, sig, dx, (long)n, lo, hi);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}
