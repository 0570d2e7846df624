// K5b — language-model side kernels (Applications/Text.py):
//  * embedding gather with a per-vocabulary-row dropout mask (EmbeddingDropout.forward :465-475:
//    F.embedding(x, W * mask[V,1], pad)) and its dense scatter-add backward (padding row gets no gradient);
//  * fused softmax + cross-entropy over the vocabulary (F.cross_entropy inside RegSeqCrossEntropyLoss :773):
//    one online-softmax pass per row forward (max and sum-exp together), one pass backward.
// Both are HBM-bound: CE reads 4*V B per token forward and reads+writes 8*V B backward (V = 47 343: 848 MB of
// logits per 4 480-token step).
#include "scatter_det.h"

namespace {

constexpr int kBlock = 256;

__global__ void emb_rowmask_fwd_kernel(const int64_t* __restrict__ x, const float* __restrict__ W,
                                       const float* __restrict__ rowmask, float* __restrict__ out, long n, int V, int D,
                                       int32_t* __restrict__ err_flag) {
  const long total = n * D;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / D;
    const int d = (int)(i - r * D);
    const int64_t idx = x[r];
    float v = 0.f;
    if (idx >= 0 && idx < V) {
      v = W[idx * D + d];
      if (rowmask) v *= rowmask[idx];
    } else if (err_flag && d == 0) {
      *err_flag = 1;
    }
    out[i] = v;
  }
}

__global__ void emb_rowmask_bwd_kernel(const int64_t* __restrict__ x, const float* __restrict__ rowmask,
                                       const float* __restrict__ dout, float* __restrict__ dW, long n, int V, int D,
                                       long padding_idx) {
  const long total = n * D;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / D;
    const int d = (int)(i - r * D);
    const int64_t idx = x[r];
    if (idx < 0 || idx >= V || idx == padding_idx) continue;
    const float m = rowmask ? rowmask[idx] : 1.f;
    if (m != 0.f) atomicAdd(dW + idx * D + d, dout[i] * m);
  }
}

struct MS { float m, s; };
__device__ __forceinline__ MS ms_merge(MS a, MS b) {
  const float m = fmaxf(a.m, b.m);
  MS r;
  r.m = m;
  r.s = (a.s == 0.f ? 0.f : a.s * expf(a.m - m)) + (b.s == 0.f ? 0.f : b.s * expf(b.m - m));
  return r;
}

// one block per row: lse[r] = log sum exp(logits[r,:]); loss[r] = lse[r] - logits[r, target[r]]
__global__ __launch_bounds__(kBlock) void softmax_ce_fwd_kernel(const float* __restrict__ logits,
                                                                 const int64_t* __restrict__ target,
                                                                 float* __restrict__ lse, float* __restrict__ loss, int V,
                                                                 int32_t* __restrict__ err_flag) {
  __shared__ float sm[4], ss[4];
  const long r = blockIdx.x;
  const float* __restrict__ row = logits + r * V;
  MS acc = {-INFINITY, 0.f};
  int v = threadIdx.x;
  for (; v + 3 * kBlock < V; v += 4 * kBlock) {          // 4 independent loads in flight per lane
    const float a = row[v], b = row[v + kBlock], c = row[v + 2 * kBlock], d = row[v + 3 * kBlock];
    const float m = fmaxf(fmaxf(fmaxf(a, b), fmaxf(c, d)), acc.m);
    acc.s = acc.s * expf(acc.m - m) + ((expf(a - m) + expf(b - m)) + (expf(c - m) + expf(d - m)));
    acc.m = m;
  }
  for (; v < V; v += kBlock) {
    const float a = row[v];
    const float m = fmaxf(a, acc.m);
    acc.s = acc.s * expf(acc.m - m) + expf(a - m);
    acc.m = m;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    MS other = {__shfl_xor(acc.m, o, 64), __shfl_xor(acc.s, o, 64)};
    acc = ms_merge(acc, other);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { sm[wave] = acc.m; ss[wave] = acc.s; }
  __syncthreads();
  if (threadIdx.x == 0) {
    MS t = {sm[0], ss[0]};
    for (int w = 1; w < 4; ++w) t = ms_merge(t, MS{sm[w], ss[w]});
    const float l = t.m + logf(t.s);
    lse[r] = l;
    const int64_t tg = target[r];
    if (tg >= 0 && tg < V) loss[r] = l - row[tg];
    else { loss[r] = 0.f; if (err_flag) *err_flag = 1; }
  }
}

// out[0] = mean(loss[0..rows))  (single block, fixed order)
__global__ void mean_kernel(const float* __restrict__ v, float* __restrict__ out, long n) {
  __shared__ float red[kBlock];
  float a = 0.f;
  for (long i = threadIdx.x; i < n; i += kBlock) a += v[i];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int w = kBlock / 2; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0] / (float)n;
}

// dlogits[r,v] = (exp(logits[r,v] - lse[r]) - [v == target[r]]) * gup / rows
__global__ __launch_bounds__(kBlock) void softmax_ce_bwd_kernel(const float* __restrict__ logits,
                                                                 const int64_t* __restrict__ target,
                                                                 const float* __restrict__ lse, const float* __restrict__ gup,
                                                                 float* __restrict__ dlogits, int V, long rows, long ld) {
  const long r = blockIdx.x;
  const float* __restrict__ row = logits + r * V;
  float* __restrict__ drow = dlogits + r * ld;
  const float l = lse[r];
  const float g = gup[0] / (float)rows;
  const int64_t tg = target[r];
  for (int v = threadIdx.x; v < V; v += kBlock) {
    float p = expf(row[v] - l);
    if (v == tg) p -= 1.f;
    drow[v] = p * g;
  }
  for (int v = V + threadIdx.x; v < ld; v += kBlock) drow[v] = 0.f;        // pad columns of a padded gradient buffer
}

int grid_for(long n) {
  long b = nnl_cdiv(n, kBlock);
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" int nnl_embedding_rowmask_fwd(const int64_t* x, const float* W, const float* rowmask, float* out, int64_t n,
                                         int64_t V, int64_t D, int32_t* err_flag, void* stream) {
  NNL_CHECK_ARG(n >= 0 && V > 0 && D > 0 && D < (1 << 24), "embedding_rowmask_fwd: bad sizes");
  if (n == 0) return NNL_OK;
  NNL_CHECK_ARG(x && W && out, "embedding_rowmask_fwd: null pointer");
  hipStream_t s = (hipStream_t)stream;
  NnlProfScope prof(NNL_PROF_ELEMENTWISE, s, (double)n * (8 + 8.0 * D));
  hipLaunchKernelGGL(emb_rowmask_fwd_kernel, dim3(grid_for(n * D)), dim3(kBlock), 0, s, x, W, rowmask, out, (long)n, (int)V, (int)D,
                     err_flag);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

extern "C" size_t nnl_embedding_rowmask_bwd_workspace_bytes(int64_t n) { return n > 0 ? nnl_det::order_bytes(n, 1) : 0; }

extern "C" int nnl_embedding_rowmask_bwd(const int64_t* x, const float* rowmask, const float* dout, float* dW, int64_t n,
                                         int64_t V, int64_t D, int64_t padding_idx, void* workspace, size_t workspace_bytes,
                                         void* stream) {
  NNL_CHECK_ARG(n >= 0 && V > 0 && D > 0 && D < (1 << 24), "embedding_rowmask_bwd: bad sizes");
  NNL_CHECK_ARG(dW, "embedding_rowmask_bwd: null output");
  hipStream_t s = (hipStream_t)stream;
  NNL_CHECK_HIP(hipMemsetAsync(dW, 0, sizeof(float) * V * D, s));
  if (n == 0) return NNL_OK;
  NNL_CHECK_ARG(x && dout, "embedding_rowmask_bwd: null pointer");
  NnlProfScope prof(NNL_PROF_ELEMENTWISE, s, (double)n * (8 + 8.0 * D) + 4.0 * V * D);
  if (nnl_det::use_det(n, workspace) && workspace_bytes >= nnl_embedding_rowmask_bwd_workspace_bytes(n)) {
    int* order = (int*)workspace;                 // deterministic: the tokens of a vocabulary row are added in token order
    int st = nnl_det::sort_rows(x, 1, n, 1, order, s);
    if (st) return st;
    nnl_det::SegSumParams q{};
    q.idx = x; q.idx_stride = 1; q.order = order; q.n = (int)n; q.card = V; q.D = (int)D; q.dst = dW; q.src = dout; q.ld = D;
    q.scale_row = rowmask; q.skip_row = padding_idx;
    return nnl_det::segsum(q, 1, s);
  }
  hipLaunchKernelGGL(emb_rowmask_bwd_kernel, dim3(grid_for(n * D)), dim3(kBlock), 0, s, x, rowmask, dout, dW, (long)n, (int)V, (int)D,
                     (long)padding_idx);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

extern "C" int nnl_softmax_ce_fwd(const float* logits, const int64_t* target, float* lse, float* loss_rows, float* loss_mean,
                                  int64_t rows, int64_t V, int32_t* err_flag, void* stream) {
  NNL_CHECK_ARG(rows > 0 && V > 0 && V < (1L << 31) && rows < (1L << 31), "softmax_ce_fwd: bad sizes");
  NNL_CHECK_ARG(logits && target && lse && loss_rows && loss_mean, "softmax_ce_fwd: null pointer");
  hipStream_t s = (hipStream_t)stream;
  NnlProfScope prof(NNL_PROF_SOFTMAX_CE, s, 4.0 * rows * V);
  hipLaunchKernelGGL(softmax_ce_fwd_kernel, dim3((unsigned)rows), dim3(kBlock), 0, s, logits, target, lse, loss_rows, (int)V, err_flag);
  NNL_CHECK_LAUNCH();
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(kBlock), 0, s, (const float*)loss_rows, loss_mean, (long)rows);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

extern "C" int nnl_softmax_ce_bwd(const float* logits, const int64_t* target, const float* lse, const float* grad_out,
                                  float* dlogits, int64_t rows, int64_t V, int64_t ld_dlogits, void* stream) {
  NNL_CHECK_ARG(rows > 0 && V > 0 && V < (1L << 31) && rows < (1L << 31) && ld_dlogits >= V, "softmax_ce_bwd: bad sizes");
  NNL_CHECK_ARG(logits && target && lse && grad_out && dlogits, "softmax_ce_bwd: null pointer");
  hipStream_t s = (hipStream_t)stream;
  NnlProfScope prof(NNL_PROF_SOFTMAX_CE, s, 8.0 * rows * V);
  hipLaunchKernelGGL(softmax_ce_bwd_kernel, dim3((unsigned)rows), dim3(kBlock), 0, s, logits, target, lse, grad_out, dlogits, (int)V,
                     (long)rows, (long)ld_dlogits);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

// ---- AR / TAR regularisers of RegSeqCrossEntropyLoss (reference Text.py:765-777) ---------------------------------------------
//   reg = alpha * mean(h^2) + beta * mean((h[1:] - h[:-1])^2),   h = enc_out [T, R] (R = bs * emb_dim), contiguous.
// HBM-bound: forward reads h once (4 B/element: a block owns a strip of columns and walks t, so h[t-1] stays in a register),
// backward reads h and writes dh (8 B/element; the t-1 / t+1 neighbours are L2 hits).  Two fixed-order stages: reproducible.
namespace {

constexpr int kRegBlock = 256;
constexpr int kRegMaxBlocks = 1024;

__global__ __launch_bounds__(kRegBlock) void seq_reg_partial_kernel(const float* __restrict__ h, float* __restrict__ part, int T, long R) {
  __shared__ float red[2][kRegBlock];
  float ar = 0.f, tar = 0.f;
  for (long c = (long)blockIdx.x * kRegBlock + threadIdx.x; c < R; c += (long)gridDim.x * kRegBlock) {
    float prev = h[c];
    ar += prev * prev;
    int t = 1;
    for (; t + 8 <= T; t += 8) {                  // eight loads in flight (a load -> wait -> use loop is T serialized round trips)
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = h[(long)(t + j) * R + c];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = v[j] - prev;
        ar += v[j] * v[j];
        tar += d * d;
        prev = v[j];
      }
    }
    for (; t < T; ++t) {
      const float v = h[(long)t * R + c];
      const float d = v - prev;
      ar += v * v;
      tar += d * d;
      prev = v;
    }
  }
  red[0][threadIdx.x] = ar; red[1][threadIdx.x] = tar;
  __syncthreads();
  for (int w = kRegBlock / 2; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) { red[0][threadIdx.x] += red[0][threadIdx.x + w]; red[1][threadIdx.x] += red[1][threadIdx.x + w]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part[2 * blockIdx.x] = red[0][0]; part[2 * blockIdx.x + 1] = red[1][0]; }
}

// out[0] = alpha*ar + beta*tar, out[1] = ar = mean(h^2), out[2] = tar
__global__ __launch_bounds__(kRegBlock) void seq_reg_final_kernel(const float* __restrict__ part, int nparts, float* __restrict__ out,
                                                                    float alpha, float beta, float inv_n_ar, float inv_n_tar) {
  __shared__ float red[2][kRegBlock];
  float ar = 0.f, tar = 0.f;
  for (int i = threadIdx.x; i < nparts; i += kRegBlock) { ar += part[2 * i]; tar += part[2 * i + 1]; }
  red[0][threadIdx.x] = ar; red[1][threadIdx.x] = tar;
  __syncthreads();
  for (int w = kRegBlock / 2; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) { red[0][threadIdx.x] += red[0][threadIdx.x + w]; red[1][threadIdx.x] += red[1][threadIdx.x + w]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float a = red[0][0] * inv_n_ar, t = red[1][0] * inv_n_tar;
    out[1] = a; out[2] = t;
    out[0] = alpha * a + beta * t;
  }
}

// dh[t] = g * ( ca * h[t] + cb * ((t > 0 ? h[t] - h[t-1] : 0) - (t < T-1 ? h[t+1] - h[t] : 0)) ),  ca = 2 alpha / N_ar, cb = 2 beta / N_tar
__global__ __launch_bounds__(kRegBlock) void seq_reg_bwd_kernel(const float* __restrict__ h, const float* __restrict__ gout,
                                                                  float* __restrict__ dh, int T, long R, float ca, float cb) {
  const float g = gout ? gout[0] : 1.f;
  const long total = (long)T * R;
  for (long i = (long)blockIdx.x * kRegBlock + threadIdx.x; i < total; i += (long)gridDim.x * kRegBlock) {
    const int t = (int)(i / R);
    const float v = h[i];
    float d = 0.f;
    if (t > 0) d += v - h[i - R];
    if (t < T - 1) d -= h[i + R] - v;
    dh[i] = g * (ca * v + cb * d);
  }
}

// ---- WeightDropLSTM1's weight drop (reference Text.py:495-513: `W = Dropout_p(W_raw)`, resampled once per forward, shared by all
// timesteps), fused with the zero-padding of the recurrent matrix to the persistent kernel's k granularity:
//   out[r][c] = raw[r][c] * m(r, c) for c < H, 0 for H <= c < ld_out;   m = mask[r*H + c] when a mask is given, else
//   keep(seed, r*H + c) / (1 - p) with keep = [u >= p], u = 24 uniform bits of a splitmix64 hash of (seed, index).
// The backward re-derives m the same way (no mask tensor is stored): draw[r][c] = dW[r*ld_dw + c] * m(r, c).
__device__ __forceinline__ float wd_keep(unsigned long long seed, unsigned long long idx, float p, float scale) {
  unsigned long long x = seed + idx * 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  x = x ^ (x >> 31);
  const float u = (float)(unsigned)(x >> 40) * (1.0f / 16777216.0f);
  return u >= p ? scale : 0.f;
}

__global__ __launch_bounds__(256) void weight_drop_kernel(const float* __restrict__ src, long ld_src, const float* __restrict__ mask,
                                                           float* __restrict__ out, long ld_out, long rows, int H,
                                                           unsigned long long seed, float p, float scale, int use_hash) {
  const long total = rows * ld_out;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / ld_out;
    const int c = (int)(i - r * ld_out);
    float v = 0.f;
    if (c < H) {
      const long e = r * H + c;
      const float m = mask ? mask[e] : (use_hash ? wd_keep(seed, (unsigned long long)e, p, scale) : 1.f);
      v = src[r * ld_src + c] * m;
    }
    out[i] = v;
  }
}

}  // namespace

extern "C" size_t nnl_seq_reg_workspace_bytes(int64_t T, int64_t R) {
  return (T > 0 && R > 0) ? (size_t)kRegMaxBlocks * 2 * sizeof(float) : 0;
}

extern "C" int nnl_seq_reg_fwd(const float* h, float* out3, int64_t T, int64_t R, float alpha, float beta, void* workspace,
                               size_t workspace_bytes, void* stream) {
  NNL_CHECK_ARG(h && out3 && T > 0 && R > 0 && T < (1 << 30), "seq_reg_fwd: bad argument");
  if (workspace == nullptr || workspace_bytes < nnl_seq_reg_workspace_bytes(T, R))
    return nnl_set_error(NNL_ERR_WORKSPACE, "seq_reg_fwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  NnlProfScope prof(NNL_PROF_ELEMENTWISE, s, 4.0 * T * R);
  long blocks = nnl_cdiv(R, kRegBlock);
  if (blocks > kRegMaxBlocks) blocks = kRegMaxBlocks;
  hipLaunchKernelGGL(seq_reg_partial_kernel, dim3((unsigned)blocks), dim3(kRegBlock), 0, s, h, (float*)workspace, (int)T, (long)R);
  NNL_CHECK_LAUNCH();
  const float inv_ar = 1.f / ((float)T * (float)R), inv_tar = T > 1 ? 1.f / ((float)(T - 1) * (float)R) : 0.f;
  hipLaunchKernelGGL(seq_reg_final_kernel, dim3(1), dim3(kRegBlock), 0, s, (const float*)workspace, (int)blocks, out3, alpha, beta, inv_ar, inv_tar);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

extern "C" int nnl_seq_reg_bwd(const float* h, const float* grad_out, float* dh, int64_t T, int64_t R, float alpha, float beta,
                               void* stream) {
  NNL_CHECK_ARG(h && dh && T > 0 && R > 0 && T < (1 << 30), "seq_reg_bwd: bad argument");
  hipStream_t s = (hipStream_t)stream;
  NnlProfScope prof(NNL_PROF_ELEMENTWISE, s, 8.0 * T * R);
  const float ca = 2.f * alpha / ((float)T * (float)R), cb = T > 1 ? 2.f * beta / ((float)(T - 1) * (float)R) : 0.f;
  long blocks = nnl_cdiv(T * R, kRegBlock * 4);
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(seq_reg_bwd_kernel, dim3((unsigned)blocks), dim3(kRegBlock), 0, s, h, grad_out, dh, (int)T, (long)R, ca, cb);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

extern "C" int nnl_weight_drop(const float* src, int64_t ld_src, const float* mask, float* out, int64_t ld_out, int64_t rows,
                               int64_t H, uint64_t seed, float p, void* stream) {
  NNL_CHECK_ARG(src && out && rows > 0 && H > 0 && ld_src >= H && ld_out >= H && H < (1L << 30), "weight_drop: bad argument");
  NNL_CHECK_ARG(p >= 0.f && p < 1.f, "weight_drop: p must be in [0, 1)");
  hipStream_t s = (hipStream_t)stream;
  NnlProfScope prof(NNL_PROF_ELEMENTWISE, s, 8.0 * rows * H);
  long blocks = nnl_cdiv(rows * ld_out, 256 * 4);
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(weight_drop_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, (long)ld_src, mask, out, (long)ld_out, (long)rows,
                     (int)H, (unsigned long long)seed, p, 1.f / (1.f - p), (mask == nullptr && p > 0.f) ? 1 : 0);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}
