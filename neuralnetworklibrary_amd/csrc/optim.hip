// K8 — fused multi-tensor Optimizer.step: decoupled weight decay X *= 1 - wd_g*lr_g, global-norm
// gradient clipping and the SGD-momentum / Adam update for EVERY parameter tensor in one or three launches, replacing the
// reference's per-parameter Python loops (General/Optimizer.py:60-70; ~110 tensors x 3-5 tiny launches for ResNet-34) and
// torch.optim's per-group foreach kernels.  HBM-bound: SGD-momentum reads p, g, buf and writes p, buf (20 B/element),
// Adam reads p, g, m, v and writes p, m, v (28 B/element).
//
// Layout: the host passes a device descriptor table (one entry per tensor: pointers, element count, per-tensor lr and
// decay factor) and a chunk table (tensor index, offset) that maps blockIdx.x to 4096-element pieces; both are tiny
// and uploaded per step (gradient pointers change every backward).
#include "nnl_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kChunk = 4096;      // elements per workgroup

__global__ __launch_bounds__(kBlock) void sqnorm_partial_kernel(const nnl_optim_tensor_t* __restrict__ tensors,
                                                                 const int32_t* __restrict__ chunk_tensor,
                                                                 const int64_t* __restrict__ chunk_off,
                                                                 float* __restrict__ partial) {
  __shared__ float red[4];
  const nnl_optim_tensor_t t = tensors[chunk_tensor[blockIdx.x]];
  const long off = chunk_off[blockIdx.x];
  const long end = off + kChunk < t.numel ? off + kChunk : t.numel;
  const float* __restrict__ g = t.grad;
  float acc = 0.f;
  if (g)
    for (long i = off + threadIdx.x; i < end; i += kBlock) { const float v = g[i]; acc += v * v; }
  acc = nnl_wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// coef[0] = min(1, max_norm / (sqrt(sum partial) + 1e-6))   (torch.nn.utils.clip_grad_norm_), fixed summation order
__global__ void clip_coef_kernel(const float* __restrict__ partial, int n, const float* __restrict__ hyper, float* __restrict__ coef) {
  const float max_norm = hyper[6];
  __shared__ double red[kBlock];
  double a = 0.0;
  for (int i = threadIdx.x; i < n; i += kBlock) a += (double)partial[i];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int w = kBlock / 2; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float total = (float)sqrt(red[0]);
    const float c = max_norm / (total + 1e-6f);
    coef[0] = c < 1.f ? c : 1.f;
    coef[1] = total;
  }
}

// kind 0: SGD with momentum (dampening 0, no nesterov): buf = mom*buf + g ; p = p*decay - lr*buf     (mom == 0: p -= lr*g)
// kind 1: Adam (no amsgrad): m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p = p*decay - (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ __launch_bounds__(kBlock) void optim_step_kernel(const nnl_optim_tensor_t* __restrict__ tensors,
                                                             const int32_t* __restrict__ chunk_tensor,
                                                             const int64_t* __restrict__ chunk_off, const float* __restrict__ coef,
                                                             int kind, const float* __restrict__ hyper, int write_clipped_grad) {
  // hyper-parameters live in device memory (uploaded with the descriptor table) so that a captured hipGraph of the whole
  // training step can be replayed with new momentum / betas / bias corrections
  const float momentum = hyper[0], beta1 = hyper[1], beta2 = hyper[2], eps = hyper[3], bc1 = hyper[4], sqrt_bc2 = hyper[5];
  const float omb1 = hyper[0], omb2 = hyper[7];             // Adam: 1 - beta1, 1 - beta2 rounded once on the host (slot 0 is the momentum for SGD)
  const nnl_optim_tensor_t t = tensors[chunk_tensor[blockIdx.x]];
  const long off = chunk_off[blockIdx.x];
  const long end = off + kChunk < t.numel ? off + kChunk : t.numel;
  if (t.grad == nullptr) {                        // no gradient this step: the reference still applies the weight decay
    if (t.decay != 1.f)
      for (long i = off + threadIdx.x; i < end; i += kBlock) t.param[i] *= t.decay;
    return;
  }
  const float cscale = coef ? coef[0] : 1.f;
  const float lr = t.lr, decay = t.decay;
  float* __restrict__ p = t.param;
  float* __restrict__ gptr = t.grad;
  float* __restrict__ s1 = t.state1;
  float* __restrict__ s2 = t.state2;
  for (long i = off + threadIdx.x; i < end; i += kBlock) {
    const float g = gptr[i] * cscale;
    if (write_clipped_grad) gptr[i] = g;          // clip_grad_norm_ scales .grad in place: keep that observable
    float x = p[i] * decay;
    if (kind == 0) {
      float d = g;
      if (momentum != 0.f) { d = momentum * s1[i] + g; s1[i] = d; }
      x -= lr * d;
    } else {
      const float m = beta1 * s1[i] + omb1 * g;               // torch: exp_avg.lerp_(g, 1 - beta1) / mul_(beta1).add_(g, alpha = 1 - beta1)
      const float v = beta2 * s2[i] + omb2 * g * g;           //        exp_avg_sq.mul_(beta2).addcmul_(g, g, value = 1 - beta2)
      s1[i] = m; s2[i] = v;
      x -= (lr / bc1) * (m / (sqrtf(v) / sqrt_bc2 + eps));
    }
    p[i] = x;
  }
}

}  // namespace

// Replayed steps (Learner.use_graphs): the captured upload node re-reads ONE pinned image of the table, which the host may therefore not
// rewrite while an earlier replay is still in flight.  The values that change from step to step (per-tensor lr / decay, the 8
// hyper-parameter floats) travel separately — an eager, stream-ordered upload from a ring of pinned buffers into `dyn` before each
// replay — and this captured kernel patches them into the table after the static image has landed, so fit()'s loop can run ahead
// of the GPU (round 4).  dyn = [n x {lr, decay}] then 8 floats.
__global__ void optim_patch_kernel(nnl_optim_tensor_t* __restrict__ tensors, float* __restrict__ hyper, const float* __restrict__ dyn, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { tensors[i].lr = dyn[2 * i]; tensors[i].decay = dyn[2 * i + 1]; }
  if (i < 8) hyper[i] = dyn[2 * n + i];
}

extern "C" int nnl_optim_patch(nnl_optim_tensor_t* tensors, float* hyper, const float* dyn, int64_t n, void* stream) {
  NNL_CHECK_ARG(tensors && hyper && dyn && n > 0 && n < (1 << 24), "optim_patch: bad arguments");
  hipLaunchKernelGGL(optim_patch_kernel, dim3((unsigned)nnl_cdiv(n > 8 ? n : 8, 256)), dim3(256), 0, (hipStream_t)stream, tensors, hyper, dyn, (int)n);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

extern "C" int64_t nnl_optim_chunk_elems(void) { return kChunk; }

extern "C" int nnl_optim_step(const nnl_optim_tensor_t* tensors, const int32_t* chunk_tensor, const int64_t* chunk_off,
                              int64_t n_chunks, int kind, const float* hyper, int use_clip, float* clip_workspace, void* stream) {
  NNL_CHECK_ARG(tensors && chunk_tensor && chunk_off && hyper && n_chunks > 0 && n_chunks < (1L << 31), "optim_step: bad table");
  NNL_CHECK_ARG(kind == 0 || kind == 1, "optim_step: kind must be 0 (SGD) or 1 (Adam)");
  NNL_CHECK_ARG(!use_clip || clip_workspace, "optim_step: clipping needs a workspace of n_chunks + 2 floats");
  hipStream_t s = (hipStream_t)stream;
  NnlProfScope prof(NNL_PROF_OPTIM, s, (double)n_chunks * kChunk * (kind == 0 ? 20.0 : 28.0));
  const float* coef = nullptr;
  if (use_clip) {
    hipLaunchKernelGGL(sqnorm_partial_kernel, dim3((unsigned)n_chunks), dim3(kBlock), 0, s, tensors, chunk_tensor, chunk_off,
                       clip_workspace + 2);
    NNL_CHECK_LAUNCH();
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(kBlock), 0, s, (const float*)(clip_workspace + 2), (int)n_chunks, hyper,
                       clip_workspace);
    NNL_CHECK_LAUNCH();
    coef = clip_workspace;
  }
  hipLaunchKernelGGL(optim_step_kernel, dim3((unsigned)n_chunks), dim3(kBlock), 0, s, tensors, chunk_tensor, chunk_off, coef, kind,
                     hyper, use_clip ? 1 : 0);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}
