// K2 — BatchNorm (train / eval) fused with the residual add and ReLU that follow it in the reference's ResNet
// blocks (retinanet.py:47-48,53-57,81-95: bn -> `out += residual` -> relu), and BatchNorm1d of the FC stacks
// (General/Layers.py:40, StructuredData.py:1078).  HBM-bound: a [rows, C] row-major matrix (NHWC activations:
// rows = N*H*W) is streamed with 16-B loads; algorithmic bytes per element: fwd = 4 (stats read) + 8 (apply
// read+write) [+4 residual]; bwd = 12 (reduce reads dy,y|x) + 12..16 (apply) — see DESIGN.md.
//
// Statistics use the shifted-data form: S1 = sum(x - K), S2 = sum((x - K)^2) with the pivot K = x[0][c], so
// var = (S2 - S1^2/n)/n has no catastrophic cancellation when |mean| >> std.  Per-block partial sums are combined
// in a fixed order by a second tiny kernel => bitwise reproducible run to run.
#include "nnl_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int VEC> struct VecT;
template <> struct VecT<4> { typedef f32x4 type; };
template <> struct VecT<1> { typedef float type; };

template <int VEC> __device__ __forceinline__ float vget(const typename VecT<VEC>::type& v, int i);
template <> __device__ __forceinline__ float vget<4>(const f32x4& v, int i) { return v[i]; }
template <> __device__ __forceinline__ float vget<1>(const float& v, int i) { return v; }
template <int VEC> __device__ __forceinline__ void vset(typename VecT<VEC>::type& v, int i, float x);
template <> __device__ __forceinline__ void vset<4>(f32x4& v, int i, float x) { v[i] = x; }
template <> __device__ __forceinline__ void vset<1>(float& v, int i, float x) { v = x; }

constexpr int kBlock = 256;
constexpr int kMaxRowBlocks = 1024;
constexpr int kFinLanes = 64;     // partial-sum lanes per channel in the finalize kernels (4 channels per block)

struct Shape {
  int L;        // lanes per row inside a wave (power of two <= 64)
  int rpb;      // rows covered by one block iteration = 4 waves * 64 / L
  int gx, gy;   // grid: row blocks x channel-group tiles
};

Shape make_shape(long rows, long CG) {
  Shape s;
  s.L = 1;
  while (s.L < 64 && s.L < CG) s.L <<= 1;
  s.rpb = kBlock / s.L;
  s.gy = (int)nnl_cdiv(CG, s.L);
  long gx = nnl_cdiv(rows, (long)s.rpb * 4);              // at least 4 rows per thread
  long cap = kMaxRowBlocks / (s.gy > 0 ? s.gy : 1);
  if (cap < 1) cap = 1;
  if (gx > cap) gx = cap;
  if (gx < 1) gx = 1;
  s.gx = (int)gx;
  return s;
}

// ---- pass 1 of training forward: partial (S1, S2) per block ------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(kBlock) void bn_stats_kernel(const float* __restrict__ x, float* __restrict__ part,
                                                           long rows, int C, int L) {
  typedef typename VecT<VEC>::type V;
  __shared__ float red[kBlock * 2 * VEC];
  const int CG = C / VEC;
  const int tx = threadIdx.x & (L - 1), ty = threadIdx.x / L;
  const int rpb = kBlock / L;
  const int g = blockIdx.y * L + tx;
  const bool ok = g < CG;
  float s1[VEC], s2[VEC], piv[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) { s1[e] = 0.f; s2[e] = 0.f; piv[e] = 0.f; }
  if (ok) {
    const V pv = *reinterpret_cast<const V*>(x + (long)g * VEC);
#pragma unroll
    for (int e = 0; e < VEC; ++e) piv[e] = vget<VEC>(pv, e);
    const long rstep = (long)gridDim.x * rpb;
    long r = (long)blockIdx.x * rpb + ty;
    const float* xp = x + (long)g * VEC;
    for (; r + 3 * rstep < rows; r += 4 * rstep) {          // 4 independent 16-B loads in flight per lane
      V v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const V*>(xp + (r + u * rstep) * C);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const float d = vget<VEC>(v[u], e) - piv[e];
          s1[e] += d;
          s2[e] += d * d;
        }
    }
    for (; r < rows; r += rstep) {
      const V v = *reinterpret_cast<const V*>(xp + r * C);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float d = vget<VEC>(v, e) - piv[e];
        s1[e] += d;
        s2[e] += d * d;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    red[(threadIdx.x * VEC + e) * 2 + 0] = s1[e];
    red[(threadIdx.x * VEC + e) * 2 + 1] = s2[e];
  }
  __syncthreads();
  if (ty == 0 && ok) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float a = 0.f, b = 0.f;
      for (int j = 0; j < rpb; ++j) {
        a += red[((j * L + tx) * VEC + e) * 2 + 0];
        b += red[((j * L + tx) * VEC + e) * 2 + 1];
      }
      const long c = (long)g * VEC + e;
      part[((long)blockIdx.x * C + c) * 2 + 0] = a;
      part[((long)blockIdx.x * C + c) * 2 + 1] = b;
    }
  }
}

// Sum the per-block partials of channel c: 64 lanes each add a strided subset (fixed order), then a fixed-order
// tree over the lanes in LDS.  Block = 256 threads = 4 channels x 64 lanes.  Returns the sums to lane 0.
template <int LANES = kFinLanes>
__device__ __forceinline__ void reduce_partials(const float* __restrict__ part, int nparts, int C, int c, bool cok,
                                                float (*red)[LANES][2], float& s1, float& s2) {
  const int lane = threadIdx.x & (LANES - 1), ch = threadIdx.x / LANES;
  float a = 0.f, b = 0.f;
  if (cok) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2* pp = reinterpret_cast<const f32x2*>(part) + c;          // (S1, S2) pairs, stride C
    int p = lane;
    constexpr int U = 8;                                                // independent 8-B loads in flight per lane: the partials
    for (; p + (U - 1) * LANES < nparts; p += U * LANES) {      // were written by other XCDs, each load is a ~1 us miss
      f32x2 v[U];                                                       // (3136 conv-epilogue partials per channel on the 56x56
#pragma unroll                                                          // stage: 14 -> 7 us per finalize); added in index order
      for (int u = 0; u < U; ++u) v[u] = pp[(long)(p + u * LANES) * C];
#pragma unroll
      for (int u = 0; u < U; ++u) { a += v[u][0]; b += v[u][1]; }
    }
    for (; p < nparts; p += LANES) {
      const f32x2 v = pp[(long)p * C];
      a += v[0]; b += v[1];
    }
  }
  red[ch][lane][0] = a;
  red[ch][lane][1] = b;
  __syncthreads();
  for (int w = LANES / 2; w > 0; w >>= 1) {
    if (lane < w) {
      red[ch][lane][0] += red[ch][lane + w][0];
      red[ch][lane][1] += red[ch][lane + w][1];
    }
    __syncthreads();
  }
  s1 = red[ch][0][0];
  s2 = red[ch][0][1];
}

// tail shared by the finalize kernels: invstd, running statistics (unbiased variance), per-channel scale & shift
__device__ __forceinline__ void finish_stats(int c, int C, float m, float var, float n, const float* __restrict__ gamma,
                                             const float* __restrict__ beta, float* __restrict__ mean,
                                             float* __restrict__ invstd, float* __restrict__ running_mean,
                                             float* __restrict__ running_var, float* __restrict__ scale,
                                             float* __restrict__ shift, float eps, float momentum,
                                             float* __restrict__ pivot_out = nullptr) {
  const float is = 1.f / sqrtf(var + eps);
  mean[c] = m;
  if (pivot_out) pivot_out[c] = m;                 // next step's pivot for the conv-epilogue statistics (read above, before this)
  invstd[c] = is;
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
  if (running_var) {
    const float unbiased = n > 1.f ? var * (n / (n - 1.f)) : var;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
  }
  const float gm = gamma ? gamma[c] : 1.f;
  const float sc = is * gm;
  scale[c] = sc;
  shift[c] = (beta ? beta[c] : 0.f) - m * sc;
}

// ---- finalize: mean / invstd, running statistics, per-channel scale & shift ----------------------------------
template <int LANES>                  // partial-sum lanes per channel: 64 (4 channels per block) or 256 (one channel per block: the
__global__ void bn_finalize_kernel(const float* __restrict__ x, const float* __restrict__ part, int nparts,   // conv-epilogue partials)
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ mean, float* __restrict__ invstd,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float* __restrict__ scale, float* __restrict__ shift, long rows, int C, float eps,
                                   float momentum, long long* __restrict__ num_batches_tracked,
                                   float* __restrict__ pivot_out) {
  __shared__ float red[256 / LANES][LANES][2];
  const int c = blockIdx.x * (256 / LANES) + threadIdx.x / LANES;
  float s1, s2;
  reduce_partials<LANES>(part, nparts, C, c, c < C, red, s1, s2);
  if (c >= C || (threadIdx.x & (LANES - 1)) != 0) return;
  if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;     // nn.BatchNorm's counter, without its own launch
  const float n = (float)rows;
  const float m = x[c] + s1 / n;
  float var = (s2 - s1 * (s1 / n)) / n;
  var = fmaxf(var, 0.f);
  finish_stats(c, C, m, var, n, gamma, beta, mean, invstd, running_mean, running_var, scale, shift, eps, momentum, pivot_out);
}

// ---- cross-replica BatchNorm (SyncBN) ---------------------------------------------------------------
// local statistics of this rank: stats[c] = mean_r, stats[C+c] = M2_r = sum (x - mean_r)^2, stats[2C] / [2C+1] = the row
// count split as hi*65536 + lo (both exact in fp32).  One all_gather of 2C+2 floats per BN layer.
__global__ void bn_sync_local_kernel(const float* __restrict__ x, const float* __restrict__ part, int nparts,
                                     float* __restrict__ stats, long rows, int C) {
  __shared__ float red[4][kFinLanes][2];
  const int c = blockIdx.x * 4 + threadIdx.x / kFinLanes;
  float s1, s2;
  reduce_partials(part, nparts, C, c, c < C, red, s1, s2);
  if (c >= C || (threadIdx.x & (kFinLanes - 1)) != 0) return;
  const float n = (float)rows;
  stats[c] = x[c] + s1 / n;
  stats[C + c] = fmaxf(s2 - s1 * (s1 / n), 0.f);
  if (c == 0) {
    stats[2 * C] = (float)(rows >> 16);
    stats[2 * C + 1] = (float)(rows & 0xFFFF);
  }
}

__device__ __forceinline__ float sync_count(const float* __restrict__ st, int C) { return st[2 * C] * 65536.f + st[2 * C + 1]; }

// merge the `world` ranks' (mean, M2, n) in rank order with the pairwise update of Chan et al. — every rank computes the
// bit-identical global mean / variance — then finish exactly like the single-replica finalize
__global__ void bn_sync_merge_kernel(const float* __restrict__ all_stats, int world, const float* __restrict__ gamma,
                                     const float* __restrict__ beta, float* __restrict__ mean, float* __restrict__ invstd,
                                     float* __restrict__ running_mean, float* __restrict__ running_var,
                                     float* __restrict__ scale, float* __restrict__ shift, int C, float eps, float momentum,
                                     long long* __restrict__ num_batches_tracked) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;
  const long stride = 2L * C + 2;
  float n = 0.f, m = 0.f, M2 = 0.f;
  for (int r = 0; r < world; ++r) {
    const float* st = all_stats + r * stride;
    const float nr = sync_count(st, C);
    if (nr <= 0.f) continue;
    const float nn = n + nr;
    const float d = st[c] - m;
    m += d * (nr / nn);
    M2 += st[C + c] + d * d * (n * (nr / nn));
    n = nn;
  }
  finish_stats(c, C, m, fmaxf(M2 / n, 0.f), n, gamma, beta, mean, invstd, running_mean, running_var, scale, shift, eps, momentum);
}

__global__ void bn_eval_scale_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                     const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                     float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ invstd,
                                     int C, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float is = 1.f / sqrtf(running_var[c] + eps);
  const float sc = is * (gamma ? gamma[c] : 1.f);
  scale[c] = sc;
  shift[c] = (beta ? beta[c] : 0.f) - running_mean[c] * sc;
  if (invstd) invstd[c] = is;
}

// ---- apply: y = x*scale + shift (+ residual) (ReLU) ---------------------------------------------------------------
// ReLU keep-bits: bit (i*VEC + e) of the mask says whether element e of vector i was positive after the ReLU.  The backward
// kernels read the mask (1 bit per element) instead of re-reading y (32 bits per element).
template <int VEC>
__device__ __forceinline__ unsigned keep_bits(const unsigned* __restrict__ mask, long i) {
  const long bit = i * VEC;
  return (mask[bit >> 5] >> (bit & 31)) & ((1u << VEC) - 1u);
}

template <int VEC>
__global__ __launch_bounds__(kBlock) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                           const float* __restrict__ shift,
                                                           const float* __restrict__ residual, float* __restrict__ y,
                                                           long total_v, int CG, int relu, unsigned* __restrict__ mask) {
  typedef typename VecT<VEC>::type V;
  const long stride = (long)gridDim.x * blockDim.x;          // host guarantees stride % CG == 0 (and stride % 256 == 0)
  const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const int g = (int)((unsigned long)i0 % (unsigned)CG);
  const V sc = reinterpret_cast<const V*>(scale)[g];
  const V sh = reinterpret_cast<const V*>(shift)[g];
  for (long base = i0 - lane; base < total_v; base += stride) {      // wave-uniform trip count (the mask is packed by shuffles)
    const long i = base + lane;
    const bool ok = i < total_v;
    unsigned bits = 0;
    if (ok) {
      const V xv = reinterpret_cast<const V*>(x)[i];
      V out;
      if (residual) {
        const V rv = reinterpret_cast<const V*>(residual)[i];
#pragma unroll
        for (int e = 0; e < VEC; ++e) vset<VEC>(out, e, vget<VEC>(xv, e) * vget<VEC>(sc, e) + vget<VEC>(sh, e) + vget<VEC>(rv, e));
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) vset<VEC>(out, e, vget<VEC>(xv, e) * vget<VEC>(sc, e) + vget<VEC>(sh, e));
      }
      if (relu) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          if (vget<VEC>(out, e) > 0.f) bits |= 1u << e;
          vset<VEC>(out, e, fmaxf(vget<VEC>(out, e), 0.f));
        }
      }
      reinterpret_cast<V*>(y)[i] = out;
    }
    if (mask != nullptr) {
      if (VEC == 4) {                                        // 8 lanes x 4 bits -> one word
        unsigned w = bits << ((lane & 7) * 4);
        w |= __shfl_xor(w, 1, 64);
        w |= __shfl_xor(w, 2, 64);
        w |= __shfl_xor(w, 4, 64);
        if ((lane & 7) == 0 && ok) mask[i >> 3] = w;
      } else {                                               // 64 lanes x 1 bit -> two words
        const unsigned long long b = __ballot(bits & 1u);
        if (lane == 0 && ok) mask[base >> 5] = (unsigned)b;
        if (lane == 32 && ok) mask[(base >> 5) + 1] = (unsigned)(b >> 32);
      }
    }
  }
}

// ---- backward pass 1: partial (sum g, sum g*xhat) per block, g = dy * [y > 0] ---------------------------------------
template <int VEC>
__global__ __launch_bounds__(kBlock) void bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                                const float* __restrict__ x, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, float* __restrict__ part,
                                                                long rows, int C, int L, int relu,
                                                                const unsigned* __restrict__ mask) {
  typedef typename VecT<VEC>::type V;
  __shared__ float red[kBlock * 2 * VEC];
  const int CG = C / VEC;
  const int tx = threadIdx.x & (L - 1), ty = threadIdx.x / L;
  const int rpb = kBlock / L;
  const int g = blockIdx.y * L + tx;
  const bool ok = g < CG;
  float s1[VEC], s2[VEC], mu[VEC], is[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) { s1[e] = 0.f; s2[e] = 0.f; mu[e] = 0.f; is[e] = 0.f; }
  if (ok) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) { mu[e] = mean[(long)g * VEC + e]; is[e] = invstd[(long)g * VEC + e]; }
    const long rstep = (long)gridDim.x * rpb;
    long r = (long)blockIdx.x * rpb + ty;
    for (; r + rstep < rows; r += 2 * rstep) {              // 2 x 3 independent 16-B loads in flight per lane
      V dv[2], xv[2], yv[2];
      unsigned kb[2] = {~0u, ~0u};
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const long off = (r + u * rstep) * C + (long)g * VEC;
        dv[u] = *reinterpret_cast<const V*>(dy + off);
        xv[u] = *reinterpret_cast<const V*>(x + off);
        yv[u] = dv[u];
        if (relu) {
          if (mask) kb[u] = keep_bits<VEC>(mask, off / VEC);
          else yv[u] = *reinterpret_cast<const V*>(y + off);
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          float gg = vget<VEC>(dv[u], e);
          if (relu && (mask ? !((kb[u] >> e) & 1u) : !(vget<VEC>(yv[u], e) > 0.f))) gg = 0.f;
          s1[e] += gg;
          s2[e] += gg * ((vget<VEC>(xv[u], e) - mu[e]) * is[e]);
        }
    }
    for (; r < rows; r += rstep) {
      const long off = r * C + (long)g * VEC;
      const V dv = *reinterpret_cast<const V*>(dy + off);
      const V xv = *reinterpret_cast<const V*>(x + off);
      V yv = dv;
      unsigned kb = ~0u;
      if (relu) {
        if (mask) kb = keep_bits<VEC>(mask, off / VEC);
        else yv = *reinterpret_cast<const V*>(y + off);
      }
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        float gg = vget<VEC>(dv, e);
        if (relu && (mask ? !((kb >> e) & 1u) : !(vget<VEC>(yv, e) > 0.f))) gg = 0.f;
        s1[e] += gg;
        s2[e] += gg * ((vget<VEC>(xv, e) - mu[e]) * is[e]);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    red[(threadIdx.x * VEC + e) * 2 + 0] = s1[e];
    red[(threadIdx.x * VEC + e) * 2 + 1] = s2[e];
  }
  __syncthreads();
  if (ty == 0 && ok) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float a = 0.f, b = 0.f;
      for (int j = 0; j < rpb; ++j) {
        a += red[((j * L + tx) * VEC + e) * 2 + 0];
        b += red[((j * L + tx) * VEC + e) * 2 + 1];
      }
      const long c = (long)g * VEC + e;
      part[((long)blockIdx.x * C + c) * 2 + 0] = a;
      part[((long)blockIdx.x * C + c) * 2 + 1] = b;
    }
  }
}

// finalize backward: dbeta, dgamma and the three per-channel coefficients of dx = a*g + b*x + c0
__global__ void bn_bwd_finalize_kernel(const float* __restrict__ part, int nparts, const float* __restrict__ gamma,
                                       const float* __restrict__ mean, const float* __restrict__ invstd,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ coef,
                                       long rows, int C, int training) {
  __shared__ float red[4][kFinLanes][2];
  const int c = blockIdx.x * 4 + threadIdx.x / kFinLanes;
  float s1, s2;
  reduce_partials(part, nparts, C, c, c < C, red, s1, s2);
  if (c >= C || (threadIdx.x & (kFinLanes - 1)) != 0) return;
  if (dbeta) dbeta[c] = s1;
  if (dgamma) dgamma[c] = s2;
  const float gm = gamma ? gamma[c] : 1.f;
  const float is = invstd[c], mu = mean[c];
  const float a = gm * is;
  if (training) {
    // dx = a * (g - s1/n - xhat*s2/n),  xhat = (x - mu)*is   =>  dx = a*g + bq*x + c0
    const float n = (float)rows;
    const float bq = -a * is * (s2 / n);
    coef[c] = a;
    coef[C + c] = bq;
    coef[2 * C + c] = -a * (s1 / n) - bq * mu;
  } else {
    coef[c] = a;
    coef[C + c] = 0.f;
    coef[2 * C + c] = 0.f;
  }
}

// SyncBN backward: (a) this rank's (sum g, sum g*xhat) -> sums[2C] (all-reduced by the host), (b) coefficients from the
// GLOBAL sums and row count; dgamma / dbeta stay LOCAL sums (the gradient all-reduce averages them like every parameter)
__global__ void bn_sync_bwd_sums_kernel(const float* __restrict__ part, int nparts, float* __restrict__ sums, int C) {
  __shared__ float red[4][kFinLanes][2];
  const int c = blockIdx.x * 4 + threadIdx.x / kFinLanes;
  float s1, s2;
  reduce_partials(part, nparts, C, c, c < C, red, s1, s2);
  if (c >= C || (threadIdx.x & (kFinLanes - 1)) != 0) return;
  sums[c] = s1;
  sums[C + c] = s2;
}

__global__ void bn_sync_bwd_coef_kernel(const float* __restrict__ local_sums, const float* __restrict__ global_sums,
                                        const float* __restrict__ all_stats, int world, const float* __restrict__ gamma,
                                        const float* __restrict__ mean, const float* __restrict__ invstd,
                                        float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ coef, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float n = 0.f;
  for (int r = 0; r < world; ++r) n += sync_count(all_stats + r * (2L * C + 2), C);
  if (dbeta) dbeta[c] = local_sums[c];
  if (dgamma) dgamma[c] = local_sums[C + c];
  const float s1 = global_sums[c], s2 = global_sums[C + c];
  const float is = invstd[c], mu = mean[c];
  const float a = (gamma ? gamma[c] : 1.f) * is;
  const float bq = -a * is * (s2 / n);
  coef[c] = a;
  coef[C + c] = bq;
  coef[2 * C + c] = -a * (s1 / n) - bq * mu;
}

// ---- backward pass 2: dx = a*g + b*x + c0 ; dres = g -----------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(kBlock) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                               const float* __restrict__ x, const float* __restrict__ coef,
                                                               float* __restrict__ dx, float* __restrict__ dres,
                                                               long total_v, int CG, int C, int relu,
                                                               const unsigned* __restrict__ mask) {
  typedef typename VecT<VEC>::type V;
  const long stride = (long)gridDim.x * blockDim.x;          // host guarantees stride % CG == 0
  const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int g = (int)((unsigned)i0 % (unsigned)CG);
  const V ca = reinterpret_cast<const V*>(coef)[g];
  const V cb = reinterpret_cast<const V*>(coef + C)[g];
  const V cc = reinterpret_cast<const V*>(coef + 2 * C)[g];
  for (long i = i0; i < total_v; i += stride) {
    const V dv = reinterpret_cast<const V*>(dy)[i];
    const V xv = reinterpret_cast<const V*>(x)[i];
    V gv = dv;
    if (relu) {
      if (mask) {
        const unsigned kb = keep_bits<VEC>(mask, i);
#pragma unroll
        for (int e = 0; e < VEC; ++e)
          if (!((kb >> e) & 1u)) vset<VEC>(gv, e, 0.f);
      } else {
        const V yv = reinterpret_cast<const V*>(y)[i];
#pragma unroll
        for (int e = 0; e < VEC; ++e)
          if (!(vget<VEC>(yv, e) > 0.f)) vset<VEC>(gv, e, 0.f);
      }
    }
    V out;
#pragma unroll
    for (int e = 0; e < VEC; ++e)
      vset<VEC>(out, e, vget<VEC>(ca, e) * vget<VEC>(gv, e) + vget<VEC>(cb, e) * vget<VEC>(xv, e) + vget<VEC>(cc, e));
    reinterpret_cast<V*>(dx)[i] = out;
    if (dres) reinterpret_cast<V*>(dres)[i] = gv;
  }
}

long gcd_l(long a, long b) { while (b) { long t = a % b; a = b; b = t; } return a; }

// grid such that grid*kBlock is a multiple of CG: every thread then keeps ONE channel group for its whole
// grid-stride loop (no per-element modulo, per-channel coefficients live in registers)
int ew_grid(long total_v, long CG) {
  const long unit = CG / gcd_l(CG, kBlock);                 // blocks per aligned period
  long b = nnl_cdiv(total_v, (long)kBlock * 2);             // ~2 vectors per thread
  if (b > 8192) b = 8192;
  b = nnl_cdiv(b, unit) * unit;
  if (b < unit) b = unit;
  return (int)b;
}

}  // namespace

// workspace layout (floats): [partials: kMaxRowBlocks*C*2][scale C][shift C][coef 3C]
extern "C" size_t nnl_bn_workspace_bytes(int64_t rows, int64_t C) {
  if (rows <= 0 || C <= 0) return 0;
  return (size_t)((long)kMaxRowBlocks * C * 2 + 5 * C) * sizeof(float);
}

extern "C" int nnl_bn_fwd(const float* x, const float* gamma, const float* beta, const float* residual, float* y,
                          float* save_mean, float* save_invstd, float* running_mean, float* running_var, int64_t rows,
                          int64_t C, float eps, float momentum, int training, int relu, int64_t* num_batches_tracked,
                          uint32_t* relu_mask, const float* ext_partials, int64_t ext_rows, const float* ext_pivot,
                          float* pivot_out, void* workspace, size_t workspace_bytes, void* stream) {
  NNL_CHECK_ARG(rows > 0 && C > 0 && C < (1 << 24), "bn_fwd: bad sizes rows=%ld C=%ld", (long)rows, (long)C);
  NNL_CHECK_ARG(x && y && save_mean && save_invstd, "bn_fwd: null pointer");
  NNL_CHECK_ARG(training || (running_mean && running_var), "bn_fwd: eval mode needs running statistics");
  if (workspace == nullptr || workspace_bytes < nnl_bn_workspace_bytes(rows, C))
    return nnl_set_error(NNL_ERR_WORKSPACE, "bn_fwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  float* part = (float*)workspace;
  float* scale = part + (long)kMaxRowBlocks * C * 2;
  float* shift = scale + C;
  const int VEC = (C % 4 == 0) ? 4 : 1;
  const long CG = C / VEC;
  NnlProfScope prof(NNL_PROF_ELEMENTWISE, s, (double)rows * C * (training ? 12.0 : 8.0) + (residual ? 4.0 * rows * C : 0.0));
  if (training && ext_partials != nullptr && ext_rows > 0 && ext_pivot != nullptr) {
    // the producing convolution already reduced every 64-row tile: only the finalize (pivot = the conv's pivot)
    if (ext_rows >= 1024)      // one partial per 64-row tile of the convolution: thousands on the early stages
      hipLaunchKernelGGL(bn_finalize_kernel<256>, dim3((unsigned)C), dim3(256), 0, s, ext_pivot, ext_partials, (int)ext_rows, gamma, beta,
                         save_mean, save_invstd, running_mean, running_var, scale, shift, (long)rows, (int)C, eps, momentum,
                         (long long*)num_batches_tracked, pivot_out);
    else
      hipLaunchKernelGGL(bn_finalize_kernel<kFinLanes>, dim3((unsigned)nnl_cdiv(C, 4)), dim3(256), 0, s, ext_pivot, ext_partials,
                         (int)ext_rows, gamma, beta, save_mean, save_invstd, running_mean, running_var, scale, shift, (long)rows,
                         (int)C, eps, momentum, (long long*)num_batches_tracked, pivot_out);
    NNL_CHECK_LAUNCH();
  } else if (training) {
    const Shape sh = make_shape(rows, CG);
    if (VEC == 4)
      hipLaunchKernelGGL(bn_stats_kernel<4>, dim3(sh.gx, sh.gy), dim3(kBlock), 0, s, x, part, (long)rows, (int)C, sh.L);
    else
      hipLaunchKernelGGL(bn_stats_kernel<1>, dim3(sh.gx, sh.gy), dim3(kBlock), 0, s, x, part, (long)rows, (int)C, sh.L);
    NNL_CHECK_LAUNCH();
    hipLaunchKernelGGL(bn_finalize_kernel<kFinLanes>, dim3((unsigned)nnl_cdiv(C, 4)), dim3(256), 0, s, x, part, sh.gx, gamma, beta,
                       save_mean, save_invstd, running_mean, running_var, scale, shift, (long)rows, (int)C, eps, momentum,
                       (long long*)num_batches_tracked, pivot_out);
    NNL_CHECK_LAUNCH();
  } else {
    NNL_CHECK_HIP(hipMemcpyAsync(save_mean, running_mean, sizeof(float) * C, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(bn_eval_scale_kernel, dim3((unsigned)nnl_cdiv(C, 256)), dim3(256), 0, s, gamma, beta, running_mean,
                       running_var, scale, shift, save_invstd, (int)C, eps);
    NNL_CHECK_LAUNCH();
  }
  const long total_v = rows * CG;
  if (VEC == 4)
    hipLaunchKernelGGL(bn_apply_kernel<4>, dim3(ew_grid(total_v, CG)), dim3(kBlock), 0, s, x, scale, shift, residual, y, total_v,
                       (int)CG, relu, relu ? relu_mask : nullptr);
  else
    hipLaunchKernelGGL(bn_apply_kernel<1>, dim3(ew_grid(total_v, CG)), dim3(kBlock), 0, s, x, scale, shift, residual, y, total_v,
                       (int)CG, relu, relu ? relu_mask : nullptr);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

extern "C" int nnl_bn_bwd(const float* dy, const float* y, const uint32_t* relu_mask, const float* x, const float* gamma,
                          const float* mean, const float* invstd, float* dx, float* dres, float* dgamma, float* dbeta,
                          int64_t rows, int64_t C, int training, int relu, void* workspace, size_t workspace_bytes,
                          void* stream) {
  NNL_CHECK_ARG(rows > 0 && C > 0 && C < (1 << 24), "bn_bwd: bad sizes");
  NNL_CHECK_ARG(dy && x && mean && invstd && dx && (y || relu_mask || !relu), "bn_bwd: null pointer");
  if (workspace == nullptr || workspace_bytes < nnl_bn_workspace_bytes(rows, C))
    return nnl_set_error(NNL_ERR_WORKSPACE, "bn_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  float* part = (float*)workspace;
  float* coef = part + (long)kMaxRowBlocks * C * 2 + 2 * C;
  const int VEC = (C % 4 == 0) ? 4 : 1;
  const long CG = C / VEC;
  const Shape sh = make_shape(rows, CG);
  NnlProfScope prof(NNL_PROF_ELEMENTWISE, s, (double)rows * C * (relu ? 28.0 : 20.0) + (dres ? 4.0 * rows * C : 0.0));
  if (VEC == 4)
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<4>, dim3(sh.gx, sh.gy), dim3(kBlock), 0, s, dy, y, x, mean, invstd, part, (long)rows,
                       (int)C, sh.L, relu, relu_mask);
  else
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<1>, dim3(sh.gx, sh.gy), dim3(kBlock), 0, s, dy, y, x, mean, invstd, part, (long)rows,
                       (int)C, sh.L, relu, relu_mask);
  NNL_CHECK_LAUNCH();
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)nnl_cdiv(C, 4)), dim3(256), 0, s, part, sh.gx, gamma, mean, invstd,
                     dgamma, dbeta, coef, (long)rows, (int)C, training);
  NNL_CHECK_LAUNCH();
  const long total_v = rows * CG;
  if (VEC == 4)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<4>, dim3(ew_grid(total_v, CG)), dim3(kBlock), 0, s, dy, y, x, coef, dx, dres, total_v,
                       (int)CG, (int)C, relu, relu_mask);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<1>, dim3(ew_grid(total_v, CG)), dim3(kBlock), 0, s, dy, y, x, coef, dx, dres, total_v,
                       (int)CG, (int)C, relu, relu_mask);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

// ---- BatchNorm -> ReLU -> MaxPool2d in one pass (the ResNet stem: reference retinanet.py:372-374) -----------------------------
// The normalised, rectified activation z = relu(scale*x + shift) is never written: the forward pools it on the fly (torch's tie
// rule: first maximum in (kh, kw) scan order, NaN propagates) and keeps the uint8 window index; the backward re-derives, for
// every input pixel, g = [z > 0] * sum of dpool over the windows whose arg-max it is (a gather: no atomics), recomputing z with
// the forward's arithmetic, and feeds g to the usual two BatchNorm backward passes.  Against bn_apply + maxpool_fwd and
// maxpool_bwd + bn_bwd this saves writing and re-reading the [N,H,W,C] activation (forward) and its gradient (backward):
// ~12 + 16 bytes per pre-pool element.
namespace {

__device__ __forceinline__ f32x4 bnpool_affine(const f32x4& x, const f32x4& sc, const f32x4& sh) {
  f32x4 z;
#pragma unroll
  for (int e = 0; e < 4; ++e) z[e] = x[e] * sc[e] + sh[e];       // same expression as bn_apply_kernel (-ffp-contract=off)
  return z;
}

__global__ __launch_bounds__(kBlock) void bnpool_fwd_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, float* __restrict__ y,
                                                            uint8_t* __restrict__ idx, int N, int H, int W, int C4, int P, int Q,
                                                            int ks, int stride, int pad) {
  // one thread = one OUTPUT pixel x 4 channels; the grid stride is a multiple of C4, so a thread keeps its channels
  const long total = (long)N * P * Q * C4;
  const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c4 = (int)(i0 % C4);
  const f32x4 sc = reinterpret_cast<const f32x4*>(scale)[c4], sh = reinterpret_cast<const f32x4*>(shift)[c4];
  for (long i = i0; i < total; i += (long)gridDim.x * blockDim.x) {
    long r = i / C4;
    const int q = (int)(r % Q); r /= Q;
    const int p = (int)(r % P);
    const int n = (int)(r / P);
    f32x4 best = {0.f, 0.f, 0.f, 0.f};
    int bi[4] = {-1, -1, -1, -1};
    for (int kh = 0; kh < ks; ++kh) {
      const int h = p * stride - pad + kh;
      if ((unsigned)h >= (unsigned)H) continue;
      for (int kw = 0; kw < ks; ++kw) {
        const int w = q * stride - pad + kw;
        if ((unsigned)w >= (unsigned)W) continue;
        f32x4 v = bnpool_affine(reinterpret_cast<const f32x4*>(x)[((long)(n * H + h) * W + w) * C4 + c4], sc, sh);
        const int t = kh * ks + kw;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = (v[e] != v[e]) ? v[e] : fmaxf(v[e], 0.f);            // ReLU (NaN stays NaN, as torch.relu)
          if (bi[e] < 0 || v[e] > best[e] || v[e] != v[e]) { best[e] = v[e]; bi[e] = t; }
        }
      }
    }
    reinterpret_cast<f32x4*>(y)[i] = best;
    reinterpret_cast<uchar4*>(idx)[i] = make_uchar4((uint8_t)bi[0], (uint8_t)bi[1], (uint8_t)bi[2], (uint8_t)bi[3]);
  }
}

struct PoolGeom { int H, W, C4, P, Q, ks, stride, pad; };

// g for input element (n, h, w, c4): gathered pooled gradient, gated by the recomputed ReLU
__device__ __forceinline__ f32x4 bnpool_grad(const float* __restrict__ dpool, const uint8_t* __restrict__ idx, const f32x4& xv,
                                             const f32x4& sc, const f32x4& sh, int n, int h, int w, int c4, const PoolGeom& g) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  int p_lo = h + g.pad - g.ks + 1; p_lo = p_lo > 0 ? (p_lo + g.stride - 1) / g.stride : 0;
  int p_hi = (h + g.pad) / g.stride; if (p_hi > g.P - 1) p_hi = g.P - 1;
  int q_lo = w + g.pad - g.ks + 1; q_lo = q_lo > 0 ? (q_lo + g.stride - 1) / g.stride : 0;
  int q_hi = (w + g.pad) / g.stride; if (q_hi > g.Q - 1) q_hi = g.Q - 1;
  for (int p = p_lo; p <= p_hi; ++p) {
    const int kh = h + g.pad - p * g.stride;
    for (int q = q_lo; q <= q_hi; ++q) {
      const int t = kh * g.ks + (w + g.pad - q * g.stride);
      const long o = ((long)(n * g.P + p) * g.Q + q) * g.C4 + c4;
      const uchar4 id = reinterpret_cast<const uchar4*>(idx)[o];
      const f32x4 d = reinterpret_cast<const f32x4*>(dpool)[o];
      if (id.x == t) acc[0] += d[0];
      if (id.y == t) acc[1] += d[1];
      if (id.z == t) acc[2] += d[2];
      if (id.w == t) acc[3] += d[3];
    }
  }
  const f32x4 z = bnpool_affine(xv, sc, sh);
#pragma unroll
  for (int e = 0; e < 4; ++e) if (!(z[e] > 0.f)) acc[e] = 0.f;
  return acc;
}

constexpr int kPoolRedBlock = 1024;          // 16 waves per block: the gathers are latency-bound, so 2 blocks x 16 waves fill a CU

__global__ __launch_bounds__(kPoolRedBlock) void bnpool_bwd_reduce_kernel(const float* __restrict__ dpool, const uint8_t* __restrict__ idx,
                                                                   const float* __restrict__ x, const float* __restrict__ scale,
                                                                   const float* __restrict__ shift, const float* __restrict__ mean,
                                                                   const float* __restrict__ invstd, float* __restrict__ part,
                                                                   int N, int C, PoolGeom g) {
  __shared__ float red[kPoolRedBlock][8];
  const long total = (long)N * g.H * g.W * g.C4;
  const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c4 = (int)(i0 % g.C4);
  const f32x4 sc = reinterpret_cast<const f32x4*>(scale)[c4], sh = reinterpret_cast<const f32x4*>(shift)[c4];
  const f32x4 mu = reinterpret_cast<const f32x4*>(mean)[c4], is = reinterpret_cast<const f32x4*>(invstd)[c4];
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  for (long i = i0; i < total; i += (long)gridDim.x * blockDim.x) {
    long r = i / g.C4;
    const int w = (int)(r % g.W); r /= g.W;
    const int h = (int)(r % g.H);
    const int n = (int)(r / g.H);
    const f32x4 xv = reinterpret_cast<const f32x4*>(x)[i];
    const f32x4 gg = bnpool_grad(dpool, idx, xv, sc, sh, n, h, w, c4, g);
#pragma unroll
    for (int e = 0; e < 4; ++e) { s1[e] += gg[e]; s2[e] += gg[e] * ((xv[e] - mu[e]) * is[e]); }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { red[threadIdx.x][e * 2] = s1[e]; red[threadIdx.x][e * 2 + 1] = s2[e]; }
  __syncthreads();
  if ((int)threadIdx.x < g.C4) {                                   // threads t, t + C4, t + 2*C4 ... hold the same channels
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a = 0.f, b = 0.f;
      for (int j = threadIdx.x; j < kPoolRedBlock; j += g.C4) { a += red[j][e * 2]; b += red[j][e * 2 + 1]; }
      const long c = (long)threadIdx.x * 4 + e;
      part[((long)blockIdx.x * C + c) * 2 + 0] = a;
      part[((long)blockIdx.x * C + c) * 2 + 1] = b;
    }
  }
}

// The same two sums taken over the POOLED outputs instead of the inputs: sum_in g*f = sum_out dpool * [y > 0] * f(arg-max), and at
// the arg-max the normalised value follows from the pooled output itself, xhat = (y - beta) / gamma (y = scale*x + shift there),
// so neither x nor the window gather is needed: 8 B per pooled element instead of 4 B per input element + ~2.25 gathers.
// Channels with |gamma| <= 1e-2 (the division would amplify rounding) read x at the arg-max position instead.
__global__ __launch_bounds__(kPoolRedBlock) void bnpool_bwd_reduce_out_kernel(const float* __restrict__ dpool, const float* __restrict__ y,
                                                                       const uint8_t* __restrict__ idx, const float* __restrict__ x,
                                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                       const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                       float* __restrict__ part, int N, int C, PoolGeom g) {
  __shared__ float red[kPoolRedBlock][8];
  const long total = (long)N * g.P * g.Q * g.C4;
  const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c4 = (int)(i0 % g.C4);
  float bt[4], rg[4], mu[4], is[4];
  bool fast[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = c4 * 4 + e;
    const float gm = gamma ? gamma[c] : 1.f;
    bt[e] = beta ? beta[c] : 0.f;
    fast[e] = fabsf(gm) > 1e-2f;
    rg[e] = fast[e] ? 1.f / gm : 0.f;
    mu[e] = mean[c]; is[e] = invstd[c];
  }
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  for (long i = i0; i < total; i += (long)gridDim.x * blockDim.x) {
    const f32x4 d = reinterpret_cast<const f32x4*>(dpool)[i];
    const f32x4 yv = reinterpret_cast<const f32x4*>(y)[i];
    const uchar4 id = reinterpret_cast<const uchar4*>(idx)[i];
    const unsigned char t[4] = {id.x, id.y, id.z, id.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (!(yv[e] > 0.f)) continue;                                   // ReLU gate (y = relu(z) at the arg-max)
      float xh;
      if (fast[e]) {
        xh = (yv[e] - bt[e]) * rg[e];
      } else {
        long r = i / g.C4;
        const int q = (int)(r % g.Q); r /= g.Q;
        const int p = (int)(r % g.P);
        const int n = (int)(r / g.P);
        const int h = p * g.stride - g.pad + t[e] / g.ks, w = q * g.stride - g.pad + t[e] % g.ks;
        xh = (x[((long)(n * g.H + h) * g.W + w) * C + c4 * 4 + e] - mu[e]) * is[e];
      }
      s1[e] += d[e];
      s2[e] += d[e] * xh;
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { red[threadIdx.x][e * 2] = s1[e]; red[threadIdx.x][e * 2 + 1] = s2[e]; }
  __syncthreads();
  if ((int)threadIdx.x < g.C4) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a = 0.f, b = 0.f;
      for (int j = threadIdx.x; j < kPoolRedBlock; j += g.C4) { a += red[j][e * 2]; b += red[j][e * 2 + 1]; }
      const long c = (long)threadIdx.x * 4 + e;
      part[((long)blockIdx.x * C + c) * 2 + 0] = a;
      part[((long)blockIdx.x * C + c) * 2 + 1] = b;
    }
  }
}

__global__ __launch_bounds__(kBlock) void bnpool_bwd_apply_kernel(const float* __restrict__ dpool, const uint8_t* __restrict__ idx,
                                                                  const float* __restrict__ x, const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, const float* __restrict__ coef,
                                                                  float* __restrict__ dx, int N, int C, PoolGeom g) {
  const long total = (long)N * g.H * g.W * g.C4;
  const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c4 = (int)(i0 % g.C4);
  const f32x4 sc = reinterpret_cast<const f32x4*>(scale)[c4], sh = reinterpret_cast<const f32x4*>(shift)[c4];
  const f32x4 ca = reinterpret_cast<const f32x4*>(coef)[c4], cb = *reinterpret_cast<const f32x4*>(coef + C + c4 * 4),
              cc = *reinterpret_cast<const f32x4*>(coef + 2 * C + c4 * 4);
  for (long i = i0; i < total; i += (long)gridDim.x * blockDim.x) {
    long r = i / g.C4;
    const int w = (int)(r % g.W); r /= g.W;
    const int h = (int)(r % g.H);
    const int n = (int)(r / g.H);
    const f32x4 xv = reinterpret_cast<const f32x4*>(x)[i];
    const f32x4 gg = bnpool_grad(dpool, idx, xv, sc, sh, n, h, w, c4, g);
    f32x4 out;
#pragma unroll
    for (int e = 0; e < 4; ++e) out[e] = ca[e] * gg[e] + cb[e] * xv[e] + cc[e];
    reinterpret_cast<f32x4*>(dx)[i] = out;
  }
}

bool bnpool_ok(long C) { return C % 4 == 0 && C >= 4 && C <= 1024 && (kBlock % (C / 4)) == 0; }
unsigned bnpool_grid(long total) {                                 // <= kMaxRowBlocks blocks: the partials share the BN workspace
  long g = nnl_cdiv(total, (long)kPoolRedBlock * 4);
  if (g > kMaxRowBlocks) g = kMaxRowBlocks;
  return (unsigned)(g < 1 ? 1 : g);
}

}  // namespace

extern "C" int nnl_bn_relu_maxpool_supported(int64_t C) { return bnpool_ok(C) ? 1 : 0; }

extern "C" int nnl_bn_relu_maxpool_fwd(const float* x, const float* gamma, const float* beta, float* y, uint8_t* idx,
                                       float* save_mean, float* save_invstd, float* save_scale, float* save_shift,
                                       float* running_mean, float* running_var, int64_t N, int64_t H, int64_t W, int64_t C,
                                       int64_t P, int64_t Q, int ks, int stride, int pad, float eps, float momentum, int training,
                                       int64_t* num_batches_tracked, void* workspace, size_t workspace_bytes, void* stream) {
  const long rows = (long)N * H * W;
  NNL_CHECK_ARG(N > 0 && H > 0 && W > 0 && P > 0 && Q > 0 && ks > 0 && ks * ks <= 255 && stride > 0 && pad >= 0 && 2 * pad <= ks,
                "bn_relu_maxpool_fwd: bad geometry");
  NNL_CHECK_ARG(P == (H + 2 * pad - ks) / stride + 1 && Q == (W + 2 * pad - ks) / stride + 1, "bn_relu_maxpool_fwd: P/Q do not match");
  NNL_CHECK_ARG(bnpool_ok(C), "bn_relu_maxpool_fwd: C=%ld needs C %% 4 == 0 and C/4 dividing 256", (long)C);
  NNL_CHECK_ARG(x && y && idx && save_mean && save_invstd && save_scale && save_shift, "bn_relu_maxpool_fwd: null pointer");
  NNL_CHECK_ARG(training || (running_mean && running_var), "bn_relu_maxpool_fwd: eval mode needs running statistics");
  NNL_CHECK_ARG(rows * C < (1L << 31) * 4, "bn_relu_maxpool_fwd: tensor too large");
  if (workspace == nullptr || workspace_bytes < nnl_bn_workspace_bytes(rows, C))
    return nnl_set_error(NNL_ERR_WORKSPACE, "bn_relu_maxpool_fwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  float* part = (float*)workspace;
  float* scale = save_scale;                       // kept for the backward: its ReLU gate is recomputed from exactly these values
  float* shift = save_shift;
  const long CG = C / 4;
  NnlProfScope prof(NNL_PROF_ELEMENTWISE, s, (double)rows * C * (training ? 8.0 : 4.0) + (double)N * P * Q * C * 5.0);
  if (training) {
    const Shape sh = make_shape(rows, CG);
    hipLaunchKernelGGL(bn_stats_kernel<4>, dim3(sh.gx, sh.gy), dim3(kBlock), 0, s, x, part, rows, (int)C, sh.L);
    NNL_CHECK_LAUNCH();
    hipLaunchKernelGGL(bn_finalize_kernel<kFinLanes>, dim3((unsigned)nnl_cdiv(C, 4)), dim3(256), 0, s, x, part, sh.gx, gamma, beta,
                       save_mean, save_invstd, running_mean, running_var, scale, shift, rows, (int)C, eps, momentum,
                       (long long*)num_batches_tracked, (float*)nullptr);
    NNL_CHECK_LAUNCH();
  } else {
    NNL_CHECK_HIP(hipMemcpyAsync(save_mean, running_mean, sizeof(float) * C, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(bn_eval_scale_kernel, dim3((unsigned)nnl_cdiv(C, 256)), dim3(256), 0, s, gamma, beta, running_mean,
                       running_var, scale, shift, save_invstd, (int)C, eps);
    NNL_CHECK_LAUNCH();
  }
  const long total = (long)N * P * Q * CG;
  hipLaunchKernelGGL(bnpool_fwd_kernel, dim3((unsigned)nnl_cdiv(total, kBlock)), dim3(kBlock), 0, s, x, scale, shift, y, idx, (int)N,
                     (int)H, (int)W, (int)CG, (int)P, (int)Q, ks, stride, pad);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

extern "C" int nnl_bn_relu_maxpool_bwd(const float* dpool, const float* y, const uint8_t* idx, const float* x, const float* gamma,
                                       const float* beta, const float* mean, const float* invstd, const float* scale,
                                       const float* shift, float* dx, float* dgamma, float* dbeta, int64_t N,
                                       int64_t H, int64_t W, int64_t C, int64_t P, int64_t Q, int ks, int stride, int pad,
                                       int training, void* workspace, size_t workspace_bytes, void* stream) {
  const long rows = (long)N * H * W;
  NNL_CHECK_ARG(N > 0 && H > 0 && W > 0 && P > 0 && Q > 0 && ks > 0 && stride > 0 && pad >= 0, "bn_relu_maxpool_bwd: bad geometry");
  NNL_CHECK_ARG(bnpool_ok(C), "bn_relu_maxpool_bwd: C=%ld needs C %% 4 == 0 and C/4 dividing 256", (long)C);
  NNL_CHECK_ARG(dpool && idx && x && mean && invstd && scale && shift && dx, "bn_relu_maxpool_bwd: null pointer");
  if (workspace == nullptr || workspace_bytes < nnl_bn_workspace_bytes(rows, C))
    return nnl_set_error(NNL_ERR_WORKSPACE, "bn_relu_maxpool_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  float* part = (float*)workspace;
  float* coef = part + (long)kMaxRowBlocks * C * 2 + 2 * C;
  const PoolGeom g{(int)H, (int)W, (int)(C / 4), (int)P, (int)Q, ks, stride, pad};
  const long total = rows * (C / 4);
  const unsigned grid = bnpool_grid(total);
  NnlProfScope prof(NNL_PROF_ELEMENTWISE, s, (double)rows * C * 12.0 + (double)N * P * Q * C * 10.0);
  if (y != nullptr) {                      // sums over the pooled outputs (no x, no gather)
    const unsigned go = bnpool_grid((long)N * P * Q * (C / 4));
    hipLaunchKernelGGL(bnpool_bwd_reduce_out_kernel, dim3(go), dim3(kPoolRedBlock), 0, s, dpool, y, idx, x, gamma, beta, mean, invstd,
                       part, (int)N, (int)C, g);
    NNL_CHECK_LAUNCH();
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)nnl_cdiv(C, 4)), dim3(256), 0, s, part, (int)go, gamma, mean, invstd,
                       dgamma, dbeta, coef, rows, (int)C, training);
    NNL_CHECK_LAUNCH();
  } else {
    hipLaunchKernelGGL(bnpool_bwd_reduce_kernel, dim3(grid), dim3(kPoolRedBlock), 0, s, dpool, idx, x, scale, shift, mean, invstd, part,
                       (int)N, (int)C, g);
    NNL_CHECK_LAUNCH();
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)nnl_cdiv(C, 4)), dim3(256), 0, s, part, (int)grid, gamma, mean, invstd,
                       dgamma, dbeta, coef, rows, (int)C, training);
    NNL_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(bnpool_bwd_apply_kernel, dim3((unsigned)nnl_cdiv(total, kBlock)), dim3(kBlock), 0, s, dpool, idx, x, scale, shift, coef, dx, (int)N, (int)C, g);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

// ---- SyncBN entry points (split-phase: the host runs the collective between the two halves) -------------------------
namespace {
int launch_stats(const float* x, float* part, long rows, long C, hipStream_t s, Shape& sh) {
  const int VEC = (C % 4 == 0) ? 4 : 1;
  sh = make_shape(rows, C / VEC);
  if (VEC == 4)
    hipLaunchKernelGGL(bn_stats_kernel<4>, dim3(sh.gx, sh.gy), dim3(kBlock), 0, s, x, part, rows, (int)C, sh.L);
  else
    hipLaunchKernelGGL(bn_stats_kernel<1>, dim3(sh.gx, sh.gy), dim3(kBlock), 0, s, x, part, rows, (int)C, sh.L);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}
}  // namespace

extern "C" int nnl_bn_sync_stats(const float* x, float* stats, int64_t rows, int64_t C, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  NNL_CHECK_ARG(rows > 0 && C > 0 && C < (1 << 24) && x && stats, "bn_sync_stats: bad arguments");
  if (workspace == nullptr || workspace_bytes < nnl_bn_workspace_bytes(rows, C))
    return nnl_set_error(NNL_ERR_WORKSPACE, "bn_sync_stats: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  NnlProfScope prof(NNL_PROF_ELEMENTWISE, s, (double)rows * C * 4.0);
  Shape sh;
  int rc = launch_stats(x, (float*)workspace, rows, C, s, sh);
  if (rc != NNL_OK) return rc;
  hipLaunchKernelGGL(bn_sync_local_kernel, dim3((unsigned)nnl_cdiv(C, 4)), dim3(256), 0, s, x, (const float*)workspace, sh.gx,
                     stats, (long)rows, (int)C);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

extern "C" int nnl_bn_sync_fwd(const float* x, const float* all_stats, int world, const float* gamma, const float* beta,
                               const float* residual, float* y, float* save_mean, float* save_invstd, float* running_mean,
                               float* running_var, int64_t rows, int64_t C, float eps, float momentum, int relu,
                               int64_t* num_batches_tracked, uint32_t* relu_mask, void* workspace, size_t workspace_bytes,
                               void* stream) {
  NNL_CHECK_ARG(rows > 0 && C > 0 && C < (1 << 24) && world > 0, "bn_sync_fwd: bad sizes");
  NNL_CHECK_ARG(x && y && all_stats && save_mean && save_invstd, "bn_sync_fwd: null pointer");
  if (workspace == nullptr || workspace_bytes < nnl_bn_workspace_bytes(rows, C))
    return nnl_set_error(NNL_ERR_WORKSPACE, "bn_sync_fwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  float* scale = (float*)workspace + (long)kMaxRowBlocks * C * 2;
  float* shift = scale + C;
  NnlProfScope prof(NNL_PROF_ELEMENTWISE, s, (double)rows * C * 8.0 + (residual ? 4.0 * rows * C : 0.0));
  hipLaunchKernelGGL(bn_sync_merge_kernel, dim3((unsigned)nnl_cdiv(C, 256)), dim3(256), 0, s, all_stats, world, gamma, beta,
                     save_mean, save_invstd, running_mean, running_var, scale, shift, (int)C, eps, momentum,
                     (long long*)num_batches_tracked);
  NNL_CHECK_LAUNCH();
  const int VEC = (C % 4 == 0) ? 4 : 1;
  const long CG = C / VEC, total_v = rows * CG;
  if (VEC == 4)
    hipLaunchKernelGGL(bn_apply_kernel<4>, dim3(ew_grid(total_v, CG)), dim3(kBlock), 0, s, x, scale, shift, residual, y, total_v,
                       (int)CG, relu, relu ? relu_mask : nullptr);
  else
    hipLaunchKernelGGL(bn_apply_kernel<1>, dim3(ew_grid(total_v, CG)), dim3(kBlock), 0, s, x, scale, shift, residual, y, total_v,
                       (int)CG, relu, relu ? relu_mask : nullptr);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

extern "C" int nnl_bn_sync_bwd_reduce(const float* dy, const float* y, const uint32_t* relu_mask, const float* x,
                                      const float* mean, const float* invstd, float* sums, int64_t rows, int64_t C, int relu,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  NNL_CHECK_ARG(rows > 0 && C > 0 && C < (1 << 24), "bn_sync_bwd_reduce: bad sizes");
  NNL_CHECK_ARG(dy && x && mean && invstd && sums && (y || relu_mask || !relu), "bn_sync_bwd_reduce: null pointer");
  if (workspace == nullptr || workspace_bytes < nnl_bn_workspace_bytes(rows, C))
    return nnl_set_error(NNL_ERR_WORKSPACE, "bn_sync_bwd_reduce: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  float* part = (float*)workspace;
  const int VEC = (C % 4 == 0) ? 4 : 1;
  const Shape sh = make_shape(rows, C / VEC);
  NnlProfScope prof(NNL_PROF_ELEMENTWISE, s, (double)rows * C * (relu ? 12.0 : 8.0));
  if (VEC == 4)
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<4>, dim3(sh.gx, sh.gy), dim3(kBlock), 0, s, dy, y, x, mean, invstd, part, (long)rows,
                       (int)C, sh.L, relu, relu_mask);
  else
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<1>, dim3(sh.gx, sh.gy), dim3(kBlock), 0, s, dy, y, x, mean, invstd, part, (long)rows,
                       (int)C, sh.L, relu, relu_mask);
  NNL_CHECK_LAUNCH();
  hipLaunchKernelGGL(bn_sync_bwd_sums_kernel, dim3((unsigned)nnl_cdiv(C, 4)), dim3(256), 0, s, (const float*)part, sh.gx, sums,
                     (int)C);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

extern "C" int nnl_bn_sync_bwd(const float* dy, const float* y, const uint32_t* relu_mask, const float* x, const float* gamma,
                               const float* mean,
                               const float* invstd, const float* local_sums, const float* global_sums, const float* all_stats,
                               int world, float* dx, float* dres, float* dgamma, float* dbeta, int64_t rows, int64_t C, int relu,
                               void* workspace, size_t workspace_bytes, void* stream) {
  NNL_CHECK_ARG(rows > 0 && C > 0 && C < (1 << 24) && world > 0, "bn_sync_bwd: bad sizes");
  NNL_CHECK_ARG(dy && x && mean && invstd && dx && local_sums && global_sums && all_stats && (y || relu_mask || !relu),
                "bn_sync_bwd: null pointer");
  if (workspace == nullptr || workspace_bytes < nnl_bn_workspace_bytes(rows, C))
    return nnl_set_error(NNL_ERR_WORKSPACE, "bn_sync_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  float* coef = (float*)workspace + (long)kMaxRowBlocks * C * 2 + 2 * C;
  NnlProfScope prof(NNL_PROF_ELEMENTWISE, s, (double)rows * C * (relu ? 16.0 : 12.0) + (dres ? 4.0 * rows * C : 0.0));
  hipLaunchKernelGGL(bn_sync_bwd_coef_kernel, dim3((unsigned)nnl_cdiv(C, 256)), dim3(256), 0, s, local_sums, global_sums, all_stats,
                     world, gamma, mean, invstd, dgamma, dbeta, coef, (int)C);
  NNL_CHECK_LAUNCH();
  const int VEC = (C % 4 == 0) ? 4 : 1;
  const long CG = C / VEC, total_v = rows * CG;
  if (VEC == 4)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<4>, dim3(ew_grid(total_v, CG)), dim3(kBlock), 0, s, dy, y, x, coef, dx, dres, total_v,
                       (int)CG, (int)C, relu, relu_mask);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<1>, dim3(ew_grid(total_v, CG)), dim3(kBlock), 0, s, dy, y, x, coef, dx, dres, total_v,
                       (int)CG, (int)C, relu, relu_mask);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}
