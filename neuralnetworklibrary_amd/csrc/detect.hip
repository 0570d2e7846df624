// Detection INFERENCE post-processing: the device half of BBoxPredictor.__call__ / nms
// (reference Applications/VisionModels/retinanet.py:523-812).  The reference thresholds and decodes with ~25 eager torch ops
// per image, copies everything to Python lists and runs a `while` loop with a numpy IoU per kept box.  Here:
//   1. bbox_decode_kernel   one thread per (image, anchor): best class (first maximum), threshold, decode with mean / std,
//                           clip to the image, drop empty boxes, append to the image's candidate list;
//   2. rank_sort_kernel     exact top_k + descending sort by RANKING: every candidate counts the candidates that precede it
//                           under the total order (score desc, anchor index asc) — O(n^2) compares on LDS tiles, but n <= 49104
//                           is ~2.4e9 compares = tens of microseconds on 256 CUs, needs no multi-pass radix machinery and is
//                           deterministic (the atomic append order of step 1 does not matter);
//   3. nms_mask_kernel      the [m x m] suppression bit matrix (IoU > max_overlap and same class) in 64x64 tiles;
//   4. nms_scan_kernel      greedy scan, one wave per image, 64 boxes per step: the 64x64 diagonal tile is resolved serially
//                           in registers, then the kept boxes' rows are OR-ed into the removed set by all lanes.
// The IoU is evaluated with exactly the reference's fp32 operation order (numpy: inter / ((a1 + a2) - inter), no FMA
// contraction: the library is built with -ffp-contract=off), so keep / suppress decisions are identical; box coordinates
// can differ from a CPU run by the ulp of expf.  The rest of nms (relative thresholds, inclusion / duplicate filters on the
// few surviving boxes) is list logic and stays on the host (Applications/VisionModels/retinanet.py).
#include "nnl_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

__device__ __forceinline__ unsigned desc_key(float s) {        // larger score -> smaller key
  unsigned u = __float_as_uint(s);
  u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;                   // ascending-sortable
  return ~u;
}

__global__ __launch_bounds__(256) void bbox_decode_kernel(const float* __restrict__ anchors, const float* __restrict__ reg,
                                                          const float* __restrict__ clas, int A, int K, f32x4 mean, f32x4 stdv,
                                                          float thresh, float width, float height, float* __restrict__ cbox,
                                                          int* __restrict__ ccls, float* __restrict__ cscore,
                                                          int* __restrict__ corder, int* __restrict__ ccount) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x, img = blockIdx.y;
  if (a >= A) return;
  const float* cl = clas + ((long)img * A + a) * K;
  float best = cl[0];
  int bk = 0;
  for (int k = 1; k < K; ++k) {
    const float v = cl[k];
    if (v > best || (v != v && best == best)) { best = v; bk = k; }      // first maximum (torch.max); NaN wins
  }
  if (!(best > thresh)) return;
  const f32x4 an = reinterpret_cast<const f32x4*>(anchors)[a];
  const f32x4 r = reinterpret_cast<const f32x4*>(reg)[(long)img * A + a];
  const float w = an[2] - an[0], h = an[3] - an[1];
  const float cx = an[0] + 0.5f * w, cy = an[1] + 0.5f * h;
  const float dx = r[0] * stdv[0] + mean[0], dy = r[1] * stdv[1] + mean[1];
  const float dw = r[2] * stdv[2] + mean[2], dh = r[3] * stdv[3] + mean[3];
  const float pcx = cx + w * dx, pcy = cy + h * dy;
  const float pw = w * expf(dw), ph = h * expf(dh);
  float x0 = pcx - 0.5f * pw, y0 = pcy - 0.5f * ph, x1 = pcx + 0.5f * pw, y1 = pcy + 0.5f * ph;
  x0 = fmaxf(x0, 0.f); y0 = fmaxf(y0, 0.f);
  x1 = fminf(x1, width); y1 = fminf(y1, height);
  if (!((x1 - x0) > 0.f) || !((y1 - y0) > 0.f)) return;
  const int pos = atomicAdd(&ccount[img], 1);
  const long o = (long)img * A + pos;
  const f32x4 bx = {x0, y0, x1, y1};
  reinterpret_cast<f32x4*>(cbox)[o] = bx;
  ccls[o] = bk; cscore[o] = best; corder[o] = a;
}

// rank of every candidate under (score desc, order asc); the first top_k go to their sorted position
constexpr int kRankTile = 1024;
__global__ __launch_bounds__(256) void rank_sort_kernel(const float* __restrict__ cbox, const int* __restrict__ ccls,
                                                        const float* __restrict__ cscore, const int* __restrict__ corder,
                                                        const int* __restrict__ ccount, int cap, int top_k,
                                                        float* __restrict__ sbox, int* __restrict__ scls,
                                                        float* __restrict__ sscore, int* __restrict__ scount) {
  __shared__ u64 tile[kRankTile];
  const int img = blockIdx.y;
  int n = ccount[img];
  if (n > cap) n = cap;
  if (blockIdx.x == 0 && threadIdx.x == 0) scount[img] = n < top_k ? n : top_k;
  if ((int)blockIdx.x * 256 >= n) return;                       // uniform per block
  const long base = (long)img * cap;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool live = i < n;
  const u64 mine = live ? (((u64)desc_key(cscore[base + i]) << 32) | (unsigned)(corder ? corder[base + i] : i)) : ~0ull;
  int rank = 0;
  for (int t0 = 0; t0 < n; t0 += kRankTile) {
    const int cnt = min(kRankTile, n - t0);
    __syncthreads();
    for (int j = threadIdx.x; j < cnt; j += 256)
      tile[j] = ((u64)desc_key(cscore[base + t0 + j]) << 32) | (unsigned)(corder ? corder[base + t0 + j] : t0 + j);
    __syncthreads();
    int r = 0;
#pragma unroll 8
    for (int j = 0; j < cnt; ++j) r += tile[j] < mine;           // LDS broadcast reads
    rank += r;
  }
  if (live && rank < top_k) {
    const long o = (long)img * top_k + rank;
    reinterpret_cast<f32x4*>(sbox)[o] = reinterpret_cast<const f32x4*>(cbox)[base + i];
    scls[o] = ccls[base + i];
    sscore[o] = cscore[base + i];
  }
}

__device__ __forceinline__ bool overlaps(const f32x4 a, const f32x4 b, float thr) {
  // numpy order (retinanet.py:500-521): inter / ((area_a + area_b) - inter)
  const float iw = fmaxf(fminf(a[2], b[2]) - fmaxf(a[0], b[0]), 0.f);
  const float ih = fmaxf(fminf(a[3], b[3]) - fmaxf(a[1], b[1]), 0.f);
  const float inter = iw * ih;
  const float aa = (a[2] - a[0]) * (a[3] - a[1]);
  const float ab = (b[2] - b[0]) * (b[3] - b[1]);
  const float uni = (aa + ab) - inter;
  return (inter / uni) > thr;
}

// mask[img][i][w] bit b: box j = w*64 + b (j > i) would be deleted when box i is kept
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ sbox, const int* __restrict__ scls,
                                                      const int* __restrict__ scount, int top_k, int words, float thr,
                                                      u64* __restrict__ mask) {
  __shared__ f32x4 cb[64];
  __shared__ int cc[64];
  const int img = blockIdx.z, m = scount[img];
  const int row0 = blockIdx.y * 64, col0 = blockIdx.x * 64;
  if (row0 >= m || col0 >= m || col0 + 63 < row0) return;        // tiles below the diagonal are never read
  const long base = (long)img * top_k;
  const int j = col0 + threadIdx.x;
  if (j < m) { cb[threadIdx.x] = reinterpret_cast<const f32x4*>(sbox)[base + j]; cc[threadIdx.x] = scls[base + j]; }
  __syncthreads();
  const int i = row0 + threadIdx.x;
  if (i >= m) return;
  const f32x4 bi = reinterpret_cast<const f32x4*>(sbox)[base + i];
  const int ci = scls[base + i];
  u64 bits = 0;
  const int lim = min(64, m - col0);
  for (int b = 0; b < lim; ++b)
    if (col0 + b > i && cc[b] == ci && overlaps(bi, cb[b], thr)) bits |= 1ull << b;
  mask[((long)img * top_k + i) * words + blockIdx.x] = bits;
}

// one wave per image
__global__ __launch_bounds__(64) void nms_scan_kernel(const float* __restrict__ sbox, const int* __restrict__ scls,
                                                      const float* __restrict__ sscore, const int* __restrict__ scount,
                                                      int top_k, int words, const u64* __restrict__ mask,
                                                      float* __restrict__ kbox, int* __restrict__ kcls, float* __restrict__ kscore,
                                                      int* __restrict__ kcount) {
  extern __shared__ u64 removed[];                                // [words]
  const int img = blockIdx.x, m = scount[img], lane = threadIdx.x;
  const long base = (long)img * top_k;
  const u64* mk = mask + base * words;
  for (int w = lane; w < words; w += 64) removed[w] = 0;
  __syncthreads();
  int nkept = 0;
  const int chunks = (m + 63) / 64;
  for (int c = 0; c < chunks; ++c) {
    const int i = c * 64 + lane;
    // diagonal tile: lane l holds row (c*64 + l)'s word c; resolve the 64 boxes of the chunk in order
    const u64 diag = i < m ? mk[(long)i * words + c] : 0;
    u64 rem = removed[c];
    u64 kept = 0;
    const int lim = min(64, m - c * 64);
    for (int b = 0; b < lim; ++b) {
      const u64 row = __shfl(diag, b, 64);                        // every lane follows the same scalar recurrence
      if (!((rem >> b) & 1ull)) { kept |= 1ull << b; rem |= row; }
    }
    // rows of the kept boxes delete later boxes: lane-strided words, loads are independent of each other
    for (int w = c + 1 + lane; w < chunks; w += 64) {            // (words >= chunks hold no boxes and were never written)
      u64 acc = removed[w];
      u64 k = kept;
      while (k) {
        const int b = __ffsll((long long)k) - 1;
        k &= k - 1;
        acc |= mk[(long)(c * 64 + b) * words + w];
      }
      removed[w] = acc;
    }
    __syncthreads();
    // emit the kept boxes of this chunk in order
    const u64 kw = kept;                                          // identical in every lane
    if ((kw >> lane) & 1ull) {
      const int pos = nkept + __popcll(kw & ((1ull << lane) - 1ull));
      const long o = (long)img * top_k + pos;
      reinterpret_cast<f32x4*>(kbox)[o] = reinterpret_cast<const f32x4*>(sbox)[base + i];
      kcls[o] = scls[base + i];
      kscore[o] = sscore[base + i];
    }
    nkept += __popcll(kw);
    __syncthreads();
  }
  if (lane == 0) kcount[img] = nkept;
}

}  // namespace

extern "C" int nnl_bbox_decode(const float* anchors, const float* reg, const float* clas, int64_t bs, int64_t A, int64_t K,
                               const float* mean4, const float* std4, float thresh, float width, float height,
                               float* cand_boxes, int32_t* cand_classes, float* cand_scores, int32_t* cand_order,
                               int32_t* cand_count, void* stream) {
  NNL_CHECK_ARG(anchors && reg && clas && mean4 && std4 && cand_boxes && cand_classes && cand_scores && cand_order && cand_count,
                "bbox_decode: null pointer");
  NNL_CHECK_ARG(bs > 0 && bs < 65536 && A > 0 && A < (1L << 30) && K > 0, "bbox_decode: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  NnlProfScope prof(NNL_PROF_ELEMENTWISE, s, 4.0 * bs * A * (K + 8));
  NNL_CHECK_HIP(hipMemsetAsync(cand_count, 0, sizeof(int32_t) * bs, s));
  const f32x4 mean = {mean4[0], mean4[1], mean4[2], mean4[3]}, stdv = {std4[0], std4[1], std4[2], std4[3]};
  hipLaunchKernelGGL(bbox_decode_kernel, dim3((unsigned)nnl_cdiv(A, 256), (unsigned)bs), dim3(256), 0, s, anchors, reg, clas, (int)A,
                     (int)K, mean, stdv, thresh, width, height, cand_boxes, cand_classes, cand_scores, cand_order, cand_count);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

static inline int64_t nms_words(int64_t top_k) { return nnl_cdiv(top_k, 64); }

extern "C" size_t nnl_nms_workspace_bytes(int64_t bs, int64_t top_k) {
  if (bs <= 0 || top_k <= 0) return 0;
  // sorted boxes / classes / scores / counts + the suppression bit matrix
  return (size_t)bs * top_k * (16 + 4 + 4) + (size_t)bs * 16 + (size_t)bs * top_k * nms_words(top_k) * 8;
}

extern "C" int nnl_nms(const float* cand_boxes, const int32_t* cand_classes, const float* cand_scores, const int32_t* cand_order,
                       const int32_t* cand_count, int64_t bs, int64_t cap, int64_t top_k, float max_overlap, float* kept_boxes,
                       int32_t* kept_classes, float* kept_scores, int32_t* kept_count, void* workspace, size_t workspace_bytes,
                       void* stream) {
  NNL_CHECK_ARG(cand_boxes && cand_classes && cand_scores && cand_count && kept_boxes && kept_classes && kept_scores && kept_count,
                "nms: null pointer");
  NNL_CHECK_ARG(bs > 0 && bs < 65536 && cap > 0 && cap < (1L << 30) && top_k > 0 && top_k <= 16384,
                "nms: bad sizes (top_k <= 16384)");
  if (workspace == nullptr || workspace_bytes < nnl_nms_workspace_bytes(bs, top_k))
    return nnl_set_error(NNL_ERR_WORKSPACE, "nms: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int words = (int)nms_words(top_k);
  char* w = (char*)workspace;
  float* sbox = (float*)w;            w += (size_t)bs * top_k * 16;
  int* scls = (int*)w;                w += (size_t)bs * top_k * 4;
  float* sscore = (float*)w;          w += (size_t)bs * top_k * 4;
  int* scount = (int*)w;              w += (size_t)bs * 16;
  u64* mask = (u64*)w;
  NnlProfScope prof(NNL_PROF_ELEMENTWISE, s, (double)bs * cap * 8.0);
  hipLaunchKernelGGL(rank_sort_kernel, dim3((unsigned)nnl_cdiv(cap, 256), (unsigned)bs), dim3(256), 0, s, cand_boxes, cand_classes,
                     cand_scores, cand_order, cand_count, (int)cap, (int)top_k, sbox, scls, sscore, scount);
  NNL_CHECK_LAUNCH();
  hipLaunchKernelGGL(nms_mask_kernel, dim3(words, words, (unsigned)bs), dim3(64), 0, s, (const float*)sbox, (const int*)scls,
                     (const int*)scount, (int)top_k, words, max_overlap, mask);
  NNL_CHECK_LAUNCH();
  hipLaunchKernelGGL(nms_scan_kernel, dim3((unsigned)bs), dim3(64), words * sizeof(u64), s, (const float*)sbox, (const int*)scls,
                     (const float*)sscore, (const int*)scount, (int)top_k, words, (const u64*)mask, kept_boxes, kept_classes,
                     kept_scores, kept_count);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}
