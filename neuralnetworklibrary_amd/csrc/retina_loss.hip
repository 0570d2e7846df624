// K6 — fused RetinaNet detection loss: anchor/object matching + focal loss + smooth-L1 box loss for a whole batch,
// forward in one launch plus a one-block fixed-order finalize, backward in one launch, no host synchronisation.  (Folding the
// finalize into the forward with ticket hand-overs was measured SLOWER: the chain "last chunk -> image sum -> batch sum" is
// three dependent device-scope round trips at the very end of the kernel, ~15 us with the chip idle; a second launch costs ~4.)
// Replaces SSD_loss.__call__ -> ssd1 -> match_anchors_objects / focal_loss_retina / smoothL1_loss_retina
// (Applications/Vision.py:1474-1511, 1513-1530, 1532-1566, 1568-1605, 1620-1644) whose reference form is a Python
// loop over images (:1636), a Python loop over positive anchors (:1593) and two .nonzero() syncs per image.
//
// HBM-bound: per (image, anchor) 16 B anchor + 16 B reg + 4K B clas read, 4 B state written (fwd); the same read
// again + 16 + 4K B gradient written (bwd): 224 B per anchor at K = 20.  Layout of the work: a block owns a
// chunk of 256 consecutive anchors of one image.  Phase 1, one lane per anchor: 16-B anchor / state / reg / dreg accesses,
// consecutive lanes on consecutive addresses; the IoU loop runs over the image's objects in LDS; the anchor's target class goes
// to LDS.  Phase 2, the chunk's clas / dclas rows as ONE flat stream of 16-B pieces (K = 20: five per anchor), consecutive
// lanes on consecutive pieces — the class index and the anchor of an element follow from its position, the target class
// comes from LDS.  (The first version walked each anchor's K classes from one lane: an 80-B lane stride, every wave
// instruction touched 64 cache lines.)  reg is read for positive anchors only.
//
// Per-image semantics restated exactly (same operation order, single IEEE ops, no FMA contraction => the match
// thresholds see bit-identical IoU values):
//   iou(o,a) = inter / (area_o + area_a - inter); best = max_o iou (first maximum wins); pos: best > 0.5; neg: best < 0.4
//   focal   = sum over pos+neg anchors, all classes, of -w*(t*log p + (1-t)*log(1-p)), p = clamp(clas,1e-4,1-1e-4),
//             w = (alpha t + (1-alpha)(1-t)) (1-pt)^gamma, divided by max(#pos, 1)
//   smoothL1= mean over pos anchors x 4 coords of huber_{1/9}(|target_delta - reg|); 0 when there is no positive
//   loss    = mean over images of (1-beta)*smoothL1 + beta*focal
#include "nnl_common.h"

namespace {

constexpr int kBlock = 256;    // threads per block = anchors per chunk
constexpr int kMaxObj = 128;   // objects per image held in LDS
constexpr int kPre = 8;        // 16-B clas pieces per lane fetched ahead of the matching (K <= 32: the whole chunk)

// block-wide sums of three values with one LDS exchange (red: 12 floats); fixed tree => deterministic
__device__ __forceinline__ void block_sum3(float& a, float& b, float& c, float* red) {
  a = nnl_wave_sum(a); b = nnl_wave_sum(b); c = nnl_wave_sum(c);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[wave] = a; red[4 + wave] = b; red[8 + wave] = c; }
  __syncthreads();
  a = (red[0] + red[1]) + (red[2] + red[3]);
  b = (red[4] + red[5]) + (red[6] + red[7]);
  c = (red[8] + red[9]) + (red[10] + red[11]);
}

struct Encoded { float t[4]; };

__device__ __forceinline__ Encoded encode_box(const float4 a, const float4 o) {
  // Vision.py:1543-1562
  const float aw = a.z - a.x, ah = a.w - a.y;
  const float acx = a.x + 0.5f * aw, acy = a.y + 0.5f * ah;
  float tw = o.z - o.x, th = o.w - o.y;
  const float tcx = o.x + 0.5f * tw, tcy = o.y + 0.5f * th;
  tw = fmaxf(tw, 1.f);
  th = fmaxf(th, 1.f);
  Encoded e;
  e.t[0] = ((tcx - acx) / aw) / 0.1f;
  e.t[1] = ((tcy - acy) / ah) / 0.1f;
  e.t[2] = logf(tw / aw) / 0.2f;
  e.t[3] = logf(th / ah) / 0.2f;
  return e;
}

template <bool G2>
__device__ __forceinline__ float powg(float x, float gamma) { return G2 ? x * x : powf(x, gamma); }
template <bool G2>
__device__ __forceinline__ float powg1(float x, float gamma) { return G2 ? x : powf(x, gamma - 1.f); }   // x^(gamma-1)

// The valid objects of image `img` (Vision.py:1637-1638: rows padded with -1 are dropped), compacted in order into LDS by the
// first two waves: one load per lane, ballot + prefix count instead of a serial loop of dependent loads.  Returns their count.
__device__ __forceinline__ int load_objects(const float* __restrict__ boxes, const int64_t* __restrict__ cats, int img, int M,
                                            float4* s_box, float* s_area, int* s_cat, int* s_cnt) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int64_t c = -1;
  float4 b = {0.f, 0.f, 0.f, 0.f};
  if (tid < M) {                                    // M <= kMaxObj = 128: waves 0 and 1
    c = cats[(long)img * M + tid];
    b = reinterpret_cast<const float4*>(boxes)[(long)img * M + tid];
  }
  const bool keep = c >= 0;
  const unsigned long long mask = __ballot(keep);
  const int before = __popcll(mask & ((1ull << lane) - 1ull));
  if (lane == 0 && wave < 2) s_cnt[wave] = __popcll(mask);
  __syncthreads();
  if (keep) {
    const int pos = before + (wave == 1 ? s_cnt[0] : 0);
    s_box[pos] = b;
    if (s_area) s_area[pos] = (b.z - b.x) * (b.w - b.y);
    s_cat[pos] = (int)c;
  }
  __syncthreads();
  return s_cnt[0] + s_cnt[1];
}

// forward focal term of one class probability (Vision.py:1525-1528 with t in {0,1}): pt = p (t=1) or 1-p (t=0);
// w = wa*(1-pt)^gamma — for t=0 the reference evaluates 1-(1-p) in fp32, which is not bitwise p: keep that form
template <bool G2>
__device__ __forceinline__ float focal_term(float x, bool is_target, float alpha, float gamma) {
  const float p = fminf(fmaxf(x, 1e-4f), 1.0f - 1e-4f);
  const float q = 1.f - p;
  // target: -(alpha (1-p)^g) log p ; other: -((1-alpha) (1-q)^g) log q — ONE logf: select its argument, not its result
  const float w = is_target ? alpha * powg<G2>(q, gamma) : (1.f - alpha) * powg<G2>(1.f - q, gamma);
  return -w * logf(is_target ? p : q);
}

template <bool G2>
__device__ __forceinline__ float focal_grad(float raw, bool is_target, float alpha, float gamma) {
  // target: L = -alpha (1-p)^g log p      => dL/dp =  alpha     (g (1-p)^(g-1) log p     - (1-p)^g / p)
  // other : L = -(1-alpha) p^g log(1-p)   => dL/dp = -(1-alpha) (g p^(g-1)     log(1-p) - p^g / (1-p))
  // one logf and one division: u = the log's argument, v = the power's base
  const float u = is_target ? raw : 1.f - raw, v = is_target ? 1.f - raw : raw;
  const float c = is_target ? alpha : -(1.f - alpha);
  const float d = c * (gamma * powg1<G2>(v, gamma) * logf(u) - powg<G2>(v, gamma) / u);
  return (raw >= 1e-4f && raw <= 1.0f - 1e-4f) ? d : 0.f;      // clamp passes gradient only inside its range
}

// one 16-B piece f of a chunk's flat clas stream: floats 4f .. 4f+3 belong to anchor (4f) / K (K % 4 == 0: never two anchors)
template <bool G2>
__device__ __forceinline__ float focal_piece(const float4 v, int f, int K, float inv_k, const int* s_t, float alpha, float gamma) {
  const int e = 4 * f;
  const int an = (int)(((float)e + 0.5f) * inv_k);              // e / K (exact: (e + .5) / K is >= .5 / K away from an integer)
  const int k0 = e - an * K;
  const int t = s_t[an];
  if (t == -2) return 0.f;
  return ((focal_term<G2>(v.x, k0 == t, alpha, gamma) + focal_term<G2>(v.y, k0 + 1 == t, alpha, gamma)) +
          focal_term<G2>(v.z, k0 + 2 == t, alpha, gamma)) + focal_term<G2>(v.w, k0 + 3 == t, alpha, gamma);
}

template <bool G2>
__device__ __forceinline__ float4 focal_grad_piece(const float4 v, int f, int K, float inv_k, const int* s_t, float alpha, float gamma,
                                                   float gc) {
  const int e = 4 * f;
  const int an = (int)(((float)e + 0.5f) * inv_k);
  const int k0 = e - an * K;
  const int t = s_t[an];
  float4 d = {0.f, 0.f, 0.f, 0.f};
  if (t != -2) {
    d.x = focal_grad<G2>(v.x, k0 == t, alpha, gamma) * gc;
    d.y = focal_grad<G2>(v.y, k0 + 1 == t, alpha, gamma) * gc;
    d.z = focal_grad<G2>(v.z, k0 + 2 == t, alpha, gamma) * gc;
    d.w = focal_grad<G2>(v.w, k0 + 3 == t, alpha, gamma) * gc;
  }
  return d;
}

// grid: (chunks per image, bs).  part[(img*gridDim.x + chunk)*3 + {0,1,2}] = {focal sum, smoothL1 sum, #pos}
template <bool G2>
__global__ __launch_bounds__(kBlock) void retina_fwd_kernel(
    const float* __restrict__ anchors, const float* __restrict__ reg, const float* __restrict__ clas,
    const float* __restrict__ boxes, const int64_t* __restrict__ cats, int32_t* __restrict__ state,
    float* __restrict__ part, int A, int K, int M, float alpha, float gamma) {
  __shared__ float4 s_box[kMaxObj];
  __shared__ float s_area[kMaxObj];
  __shared__ int s_cat[kMaxObj];
  __shared__ int s_cnt[2];
  __shared__ int s_t[kBlock];           // per anchor of the chunk: target class (>= 0), -1 negative, -2 ignored / out of range
  __shared__ float red[12];
  const int img = blockIdx.y, tid = threadIdx.x;
  const int a0 = blockIdx.x * kBlock, a = a0 + tid;
  // every global load that does not depend on the matching is issued FIRST — the chunk's class probabilities (up to kPre 16-B
  // pieces per lane: all of them at K <= 32) and this lane's anchor — so the object compaction, the IoU loop and the two
  // barriers run under their latency instead of in front of it (a block is otherwise a chain of six dependent round trips)
  const int n_anch = min(kBlock, A - a0);
  const float* __restrict__ cp = clas + ((long)img * A + a0) * K;
  const bool vec = (K & 3) == 0;
  const int nv = vec ? n_anch * K / 4 : 0;
  float4 pre[kPre];
#pragma unroll
  for (int i = 0; i < kPre; ++i) {
    const int f = tid + i * kBlock;
    if (f < nv) pre[i] = reinterpret_cast<const float4*>(cp)[f];
  }
  float4 an = {0.f, 0.f, 0.f, 0.f};
  if (a < A) an = reinterpret_cast<const float4*>(anchors)[a];
  const int m = load_objects(boxes, cats, img, M, s_box, s_area, s_cat, s_cnt);
  float focal = 0.f, sl1 = 0.f, np = 0.f;
  int tcls = -2;
  if (a < A) {
    const float area_a = (an.z - an.x) * (an.w - an.y);
    float best = -1.f;
    int arg = 0;
    for (int j = 0; j < m; ++j) {
      const float4 o = s_box[j];
      const float iw = fmaxf(fminf(o.z, an.z) - fmaxf(o.x, an.x), 0.f);
      const float ih = fmaxf(fminf(o.w, an.w) - fmaxf(o.y, an.y), 0.f);
      const float inter = iw * ih;
      const float iou = inter / ((s_area[j] + area_a) - inter);
      if (iou > best) { best = iou; arg = j; }
    }
    int st;                                   // >= 0: positive, matched object; -1: negative; -2: ignored
    if (m == 0) st = -1;                      // Vision.py:1498-1501: no objects => every anchor is a negative
    else if (best > 0.5f) st = arg;
    else if (best < 0.4f) st = -1;
    else st = -2;
    state[(long)img * A + a] = st;
    tcls = st >= 0 ? s_cat[st] : st;
    if (st >= 0) {
      np = 1.f;
      const Encoded e = encode_box(an, s_box[st]);
      const float4 r = reinterpret_cast<const float4*>(reg)[(long)img * A + a];
      const float rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float d = fabsf(e.t[c] - rr[c]);
        sl1 += (d < (1.f / 9.f)) ? 4.5f * (d * d) : d - (0.5f / 9.f);
      }
    }
  }
  s_t[tid] = tcls;
  __syncthreads();
  // the chunk's class probabilities as one flat stream
  const float inv_k = 1.f / (float)K;
  if (vec) {
#pragma unroll
    for (int i = 0; i < kPre; ++i) {
      const int f = tid + i * kBlock;
      if (f < nv) focal += focal_piece<G2>(pre[i], f, K, inv_k, s_t, alpha, gamma);
    }
    for (int f = tid + kPre * kBlock; f < nv; f += kBlock)
      focal += focal_piece<G2>(reinterpret_cast<const float4*>(cp)[f], f, K, inv_k, s_t, alpha, gamma);
  } else {
    const int ne = n_anch * K;
    for (int e = tid; e < ne; e += kBlock) {
      const int an_i = (int)(((float)e + 0.5f) * inv_k);
      const int t = s_t[an_i];
      if (t != -2) focal += focal_term<G2>(cp[e], e - an_i * K == t, alpha, gamma);
    }
  }
  block_sum3(focal, sl1, np, red);
  if (tid == 0) {
    float* o = part + ((long)img * gridDim.x + blockIdx.x) * 3;
    o[0] = focal; o[1] = sl1; o[2] = np;
  }
}

// out[0] = total loss, out[1] = reg loss, out[2] = clas loss (batch means); npos[img] kept for backward.  ONE block of 16 waves:
// wave w reduces images w, w+16, ...; lane l adds that image's partials l, l+64, ... then a fixed shuffle tree; thread 0 adds the
// 16 wave results in order.  (The first version was one thread walking bs x nblk x 3 dependent loads: 232 us.)
__global__ __launch_bounds__(1024) void retina_finalize_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                               float* __restrict__ npos, int bs, int nblk, float beta) {
  __shared__ float fin[32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float clas_sum = 0.f, reg_sum = 0.f;
  for (int i = wave; i < bs; i += 16) {
    float f = 0.f, s = 0.f, n = 0.f;
    for (int b = lane; b < nblk; b += 64) {
      const float* p = part + ((long)i * nblk + b) * 3;
      f += p[0]; s += p[1]; n += p[2];
    }
    f = nnl_wave_sum(f); s = nnl_wave_sum(s); n = nnl_wave_sum(n);
    if (lane == 0) npos[i] = n;
    clas_sum += f / fmaxf(n, 1.f);
    reg_sum += n > 0.f ? s / (n * 4.f) : 0.f;
  }
  if (lane == 0) { fin[wave] = clas_sum; fin[16 + wave] = reg_sum; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float c = 0.f, r = 0.f;
    for (int w = 0; w < 16; ++w) { c += fin[w]; r += fin[16 + w]; }
    c /= bs; r /= bs;
    out[1] = r;
    out[2] = c;
    out[0] = (1.f - beta) * r + beta * c;
  }
}

// gradients wrt reg [bs,A,4] and clas [bs,A,K]; gup = upstream d(loss) (device scalar pointer)
template <bool G2>
__global__ __launch_bounds__(kBlock) void retina_bwd_kernel(
    const float* __restrict__ anchors, const float* __restrict__ reg, const float* __restrict__ clas,
    const float* __restrict__ boxes, const int64_t* __restrict__ cats, const int32_t* __restrict__ state,
    const float* __restrict__ npos, const float* __restrict__ gup, float* __restrict__ dreg, float* __restrict__ dclas,
    int A, int K, int M, int bs, float alpha, float gamma, float beta) {
  __shared__ float4 s_box[kMaxObj];
  __shared__ int s_cat[kMaxObj];
  __shared__ int s_cnt[2];
  __shared__ int s_t[kBlock];
  const int img = blockIdx.y, tid = threadIdx.x;
  const int a0 = blockIdx.x * kBlock, a = a0 + tid;
  const int n_anch = min(kBlock, A - a0);
  const float* __restrict__ cp = clas + ((long)img * A + a0) * K;
  float* __restrict__ dp = dclas + ((long)img * A + a0) * K;
  const bool vec = (K & 3) == 0;
  const int nv = vec ? n_anch * K / 4 : 0;
  float4 pre[kPre];                                  // as in the forward: the chunk's clas pieces and the state are fetched first
#pragma unroll
  for (int i = 0; i < kPre; ++i) {
    const int f = tid + i * kBlock;
    if (f < nv) pre[i] = reinterpret_cast<const float4*>(cp)[f];
  }
  int st = -2;
  if (a < A) st = state[(long)img * A + a];
  load_objects(boxes, cats, img, M, s_box, nullptr, s_cat, s_cnt);
  const float g = gup[0];
  const float n = npos[img];
  const float gc = g * beta / (bs * fmaxf(n, 1.f));
  const float gr = n > 0.f ? g * (1.f - beta) / (bs * n * 4.f) : 0.f;
  int tcls = -2;
  if (a < A) {
    const long ia = (long)img * A + a;
    tcls = st >= 0 ? s_cat[st] : st;
    float4 dr = {0.f, 0.f, 0.f, 0.f};
    if (st >= 0) {
      const float4 an = reinterpret_cast<const float4*>(anchors)[a];
      const Encoded e = encode_box(an, s_box[st]);
      const float4 r = reinterpret_cast<const float4*>(reg)[ia];
      const float rr[4] = {r.x, r.y, r.z, r.w};
      float o[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float diff = e.t[c] - rr[c];
        const float d = fabsf(diff);
        const float dl = (d < (1.f / 9.f)) ? 9.f * d : 1.f;               // d huber / d |diff|
        const float sgn = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);   // d|diff|/d diff ; d diff / d reg = -1
        o[c] = -sgn * dl * gr;
      }
      dr.x = o[0]; dr.y = o[1]; dr.z = o[2]; dr.w = o[3];
    }
    reinterpret_cast<float4*>(dreg)[ia] = dr;
  }
  s_t[tid] = tcls;
  __syncthreads();
  const float inv_k = 1.f / (float)K;
  if (vec) {
#pragma unroll
    for (int i = 0; i < kPre; ++i) {
      const int f = tid + i * kBlock;
      if (f < nv) reinterpret_cast<float4*>(dp)[f] = focal_grad_piece<G2>(pre[i], f, K, inv_k, s_t, alpha, gamma, gc);
    }
    for (int f = tid + kPre * kBlock; f < nv; f += kBlock)
      reinterpret_cast<float4*>(dp)[f] = focal_grad_piece<G2>(reinterpret_cast<const float4*>(cp)[f], f, K, inv_k, s_t, alpha, gamma, gc);
  } else {
    const int ne = n_anch * K;
    for (int e = tid; e < ne; e += kBlock) {
      const int an = (int)(((float)e + 0.5f) * inv_k);
      const int t = s_t[an];
      dp[e] = t != -2 ? focal_grad<G2>(cp[e], e - an * K == t, alpha, gamma) * gc : 0.f;
    }
  }
}

int chunks_per_image(long A) { return (int)nnl_cdiv(A, kBlock); }

}  // namespace

extern "C" size_t nnl_retina_loss_workspace_bytes(int64_t bs, int64_t A) {
  if (bs <= 0 || A <= 0) return 0;
  return (size_t)(bs * chunks_per_image(A) * 3) * sizeof(float);
}

extern "C" int nnl_retina_loss_fwd(const float* anchors, const float* reg, const float* clas, const float* boxes,
                                   const int64_t* cats, int32_t* state, float* npos, float* out, int64_t bs, int64_t A,
                                   int64_t K, int64_t M, float beta, float alpha, float gamma, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  NNL_CHECK_ARG(bs > 0 && A > 0 && K > 0 && M >= 0 && A < (1L << 30) && bs < 65536 && K <= 4096, "retina_loss_fwd: bad sizes");
  NNL_CHECK_ARG(M <= kMaxObj, "retina_loss_fwd: at most %d objects per image (got %ld)", kMaxObj, (long)M);
  NNL_CHECK_ARG(anchors && reg && clas && state && npos && out && (M == 0 || (boxes && cats)), "retina_loss_fwd: null pointer");
  if (workspace == nullptr || workspace_bytes < nnl_retina_loss_workspace_bytes(bs, A))
    return nnl_set_error(NNL_ERR_WORKSPACE, "retina_loss_fwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(chunks_per_image(A), (unsigned)bs);
  NnlProfScope prof(NNL_PROF_RETINA_LOSS, s, (double)bs * A * (16 + 16 + 4.0 * K + 4));
  if (gamma == 2.f)
    hipLaunchKernelGGL(retina_fwd_kernel<true>, grid, dim3(kBlock), 0, s, anchors, reg, clas, boxes, cats, state, (float*)workspace,
                       (int)A, (int)K, (int)M, alpha, gamma);
  else
    hipLaunchKernelGGL(retina_fwd_kernel<false>, grid, dim3(kBlock), 0, s, anchors, reg, clas, boxes, cats, state, (float*)workspace,
                       (int)A, (int)K, (int)M, alpha, gamma);
  NNL_CHECK_LAUNCH();
  hipLaunchKernelGGL(retina_finalize_kernel, dim3(1), dim3(1024), 0, s, (const float*)workspace, out, npos, (int)bs, (int)grid.x, beta);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

extern "C" int nnl_retina_loss_bwd(const float* anchors, const float* reg, const float* clas, const float* boxes,
                                   const int64_t* cats, const int32_t* state, const float* npos, const float* grad_out,
                                   float* dreg, float* dclas, int64_t bs, int64_t A, int64_t K, int64_t M, float beta,
                                   float alpha, float gamma, void* stream) {
  NNL_CHECK_ARG(bs > 0 && A > 0 && K > 0 && M >= 0 && M <= kMaxObj && bs < 65536 && K <= 4096, "retina_loss_bwd: bad sizes");
  NNL_CHECK_ARG(anchors && reg && clas && state && npos && grad_out && dreg && dclas, "retina_loss_bwd: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(chunks_per_image(A), (unsigned)bs);
  NnlProfScope prof(NNL_PROF_RETINA_LOSS, s, (double)bs * A * (16 + 32 + 8.0 * K + 4));
  if (gamma == 2.f)
    hipLaunchKernelGGL(retina_bwd_kernel<true>, grid, dim3(kBlock), 0, s, anchors, reg, clas, boxes, cats, state, npos, grad_out, dreg,
                       dclas, (int)A, (int)K, (int)M, (int)bs, alpha, gamma, beta);
  else
    hipLaunchKernelGGL(retina_bwd_kernel<false>, grid, dim3(kBlock), 0, s, anchors, reg, clas, boxes, cats, state, npos, grad_out, dreg,
                       dclas, (int)A, (int)K, (int)M, (int)bs, alpha, gamma, beta);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}
