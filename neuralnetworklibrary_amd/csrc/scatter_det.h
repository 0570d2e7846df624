// Deterministic scatter-add of sample rows into table rows (dense embedding gradients): the replacement of fp32 atomicAdd in
// embdotbias_bwd / tab_scatter_bwd / embedding_rowmask_bwd.  torch's embedding_dense_backward on the CPU — what the reference's
// `nn.Embedding(sparse=False)` runs (CollabFiltering.py:196-204, General/Layers.py:63-76, Text.py:465-475) — adds the samples of
// a row in sample order; atomics add them in arrival order, so gradients differed bit-wise from run to run whenever an index
// repeated, while conv / LSTM / BN are bitwise reproducible.  Two kernels:
//   1. rank sort: order[c][rank] = sample i, rank = #{j : key_j < key_i or (key_j == key_i and j < i)} — n^2 integer compares against
//      wave-uniform keys (n = 8192: 67 M, a few microseconds on 256 CUs), exact and stable by construction;
//   2. segment sum: one wave per sorted position; the wave that sits on the first sample of a row adds that row's samples — lanes
//      across the row's elements and, for narrow rows, across sample slots that are combined by a fixed tree — and STORES the sum
//      (rows nobody touched keep the zero of the memset).  Rows hit once or a few times are summed exactly in sample order.
#pragma once
#include "nnl_common.h"

namespace nnl_det {

constexpr int kSortBlock = 1024;      // 16 waves: 64 samples x 16 slices of the keys
constexpr long kMaxSamples = 32768;      // above this the O(n^2) ranking stops paying for itself: callers keep the atomic kernels

// One wave = 64 samples i (one per lane); the block's sixteen waves split the n keys j between them and meet in LDS.  Every lane of a wave
// compares against the SAME key j, so the key stream is wave-uniform (scalar loads, four keys per iteration) — round 4's version gave
// one thread a whole pass over all keys staged through an LDS tile: n / 256 workgroups (18 for the language model's 4480 tokens) of a
// latency-bound loop, 213 us per step; this one runs n / 64 workgroups of n / 16 compares per lane.
static __global__ __launch_bounds__(kSortBlock) void rank_sort_kernel(const int64_t* __restrict__ idx, long stride, int n,
                                                                int* __restrict__ order) {
  constexpr int NW = kSortBlock / 64;
  __shared__ int part[NW][64];
  const int c = blockIdx.y, lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int i = blockIdx.x * 64 + lane;
  const int64_t ki = i < n ? idx[(long)i * stride + c] : 0;
  const int per = (n + NW - 1) / NW, j0 = min(w * per, n), j1 = min(j0 + per, n);
  const int64_t* kp = idx + c;
  int rank = 0;
  int j = j0;
  // rank = #{j : key_j < key_i or (key_j == key_i and j < i)}: ties keep sample order (stable)
  for (; j + 3 < j1; j += 4) {
    const int64_t k0 = kp[(long)j * stride], k1 = kp[(long)(j + 1) * stride], k2 = kp[(long)(j + 2) * stride], k3 = kp[(long)(j + 3) * stride];
    rank += (k0 < ki || (k0 == ki && j < i)) + (k1 < ki || (k1 == ki && j + 1 < i)) + (k2 < ki || (k2 == ki && j + 2 < i)) + (k3 < ki || (k3 == ki && j + 3 < i));
  }
  for (; j < j1; ++j) {
    const int64_t kj = kp[(long)j * stride];
    rank += kj < ki || (kj == ki && j < i);
  }
  part[w][lane] = rank;
  __syncthreads();
  if (w == 0 && i < n) {
    int r = 0;
#pragma unroll
    for (int q = 0; q < NW; ++q) r += part[q][lane];
    order[(long)c * n + r] = i;
  }
}

struct SegSumParams {
  const int64_t* idx; long idx_stride;            // row of sample i in column c: idx[i * idx_stride + c]
  const int* order;                               // [ncols][n] from rank_sort_kernel
  int n;
  const int32_t* card_arr; long card;             // rows of column c are valid in [0, card)      (array per column, or the scalar)
  const int32_t* dim_arr; int D;                  // elements per row
  const int32_t* coff_arr; int coff;              // first source column of this table's window
  const int64_t* dst_off_arr; long dst_off;       // dst + dst_off + row * D + d
  float* dst;
  const float* src; long ld;                      // source row r: src + r * ld + coff
  const int64_t* srcrow; long srcrow_stride; int srcrow_col;    // null: r = sample index; else r = srcrow[i * stride + col]
  long srcrow_card;                               // > 0: a sample whose source row is outside [0, srcrow_card) is SKIPPED (never read)
  const float* scale_i; long scale_i_stride;      // null: 1; else scale_i[c * scale_i_stride + i]
  const float* scale_row;                         // null: 1; else scale_row[row]
  long skip_row;                                  // a row that receives no gradient (padding_idx), or -1
};

static __device__ __forceinline__ void segsum_body(const SegSumParams& p, const int c) {
  const int lane = threadIdx.x & 63;
  const int pos = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pos >= p.n) return;
  const int* ord = p.order + (long)c * p.n;
  const int64_t row = p.idx[(long)ord[pos] * p.idx_stride + c];
  const long card = p.card_arr ? p.card_arr[c] : p.card;
  if (row < 0 || row >= card || row == p.skip_row) return;
  if (pos > 0 && p.idx[(long)ord[pos - 1] * p.idx_stride + c] == row) return;        // not the first sample of its row
  const int D = p.dim_arr ? p.dim_arr[c] : p.D;
  const int coff = p.coff_arr ? p.coff_arr[c] : p.coff;
  float* out = p.dst + (p.dst_off_arr ? p.dst_off_arr[c] : p.dst_off) + row * D;
  const float srow = p.scale_row ? p.scale_row[row] : 1.f;
  // segment length (wave-uniform): the samples of this row sit at sorted positions [pos, pos + len)
  int len = 1;                                        // found 64 positions at a time (ballot), not by a chain of dependent loads
  for (int base = pos + 1; base < p.n; base += 64) {
    const int q = base + lane;
    const bool same = q < p.n && p.idx[(long)ord[q] * p.idx_stride + c] == row;
    const unsigned long long m = __ballot(!same);
    if (m != 0ull) { len += __ffsll((long long)m) - 1; break; }
    len += 64;
  }
  // lanes = SL sample slots x DL element slots (DL = the power of two >= min(D, 64)): slot s adds samples pos+s, pos+s+SL, ...
  // in that order, then the SL partial sums are added by a fixed xor tree — one order per (D, len), so the result is bitwise
  // reproducible; a short row (len <= SL... in particular every row without a repeated index) is a plain in-order sum.  Narrow
  // tables (the tabular embeddings: 2 - 8 elements, hundreds of samples per row at cardinality 4) get 8 - 32 samples in flight
  // per wave instead of one.
  int DL = 1;
  while (DL < D && DL < 64) DL <<= 1;
  const int SL = 64 / DL, s = lane / DL, dl = lane - s * DL;
  for (int d0 = 0; d0 < D; d0 += DL) {
    const int d = d0 + dl;
    float acc = 0.f;
    for (int q = s; q < len; q += SL) {
      const int i = ord[pos + q];
      const long r = p.srcrow ? p.srcrow[(long)i * p.srcrow_stride + p.srcrow_col] : i;
      if (p.srcrow_card > 0 && (r < 0 || r >= p.srcrow_card)) continue;     // bad partner index: skip, do not multiply garbage by 0
      float sc = srow;
      if (p.scale_i) sc *= p.scale_i[(long)c * p.scale_i_stride + i];
      if (d < D) acc += p.src[r * p.ld + coff + d] * sc;
    }
    for (int o = DL; o < 64; o <<= 1) acc += __shfl_xor(acc, o, 64);
    if (s == 0 && d < D) out[d] = acc;
  }
}

static __global__ __launch_bounds__(256) void segsum_kernel(SegSumParams p) { segsum_body(p, blockIdx.y); }

// up to four INDEPENDENT segment sums (own destination, source, row indirection, width) in one launch: blockIdx.y picks the set
// (EmbeddingDotBias: dU, dM, dbu, dbi — four launches of a 64-sample step were four graph nodes of ~3 us each)
struct SegSumParams4 { SegSumParams q[4]; };
static __global__ __launch_bounds__(256) void segsum4_kernel(SegSumParams4 ps) { segsum_body(ps.q[blockIdx.y], 0); }

static inline size_t order_bytes(long n, int ncols) { return (size_t)n * ncols * sizeof(int); }

static inline int sort_rows(const int64_t* idx, long stride, long n, int ncols, int* order, hipStream_t s) {
  hipLaunchKernelGGL(rank_sort_kernel, dim3((unsigned)nnl_cdiv(n, 64L), ncols), dim3(kSortBlock), 0, s, idx, stride, (int)n, order);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

static inline int segsum(const SegSumParams& p, int ncols, hipStream_t s) {
  hipLaunchKernelGGL(segsum_kernel, dim3((unsigned)nnl_cdiv(p.n, 4), ncols), dim3(256), 0, s, p);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

static inline int segsum4(const SegSumParams4& ps, int nsets, hipStream_t s) {
  hipLaunchKernelGGL(segsum4_kernel, dim3((unsigned)nnl_cdiv(ps.q[0].n, 4), nsets), dim3(256), 0, s, ps);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

// deterministic by default; NNL_SCATTER_ATOMIC=1 selects the fp32-atomicAdd kernels (A/B and very large minibatches)
static inline bool use_det(long n, const void* workspace) { return workspace != nullptr && n <= kMaxSamples && NNL_ENV_INT("NNL_SCATTER_ATOMIC", 0) == 0; }

}  // namespace nnl_det
