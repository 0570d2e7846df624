// K6 — fused RetinaNet detection loss: anchor/object matching + focal loss + smooth-L1 box loss for a whole batch,
// forward in one launch (+ a tiny fixed-order finalize), backward in one launch, no host synchronisation.
// Replaces SSD_loss.__call__ -> ssd1 -> match_anchors_objects / focal_loss_retina / smoothL1_loss_retina
// (Applications/Vision.py:1474-1511, 1513-1530, 1532-1566, 1568-1605, 1620-1644) whose reference form is a Python
// loop over images (:1636), a Python loop over positive anchors (:1593) and two .nonzero() syncs per image.
//
// HBM-bound: per (image, anchor) 16 B anchor + 16 B reg + 4K B clas read, 4 B state written (fwd); the same read
// again + 16 + 4K B gradient written (bwd): 224 B per anchor at K = 20.
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

constexpr int kBlock = 256;
constexpr int kMaxObj = 128;   // objects per image held in LDS

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = nnl_wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
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

__device__ __forceinline__ float powg(float x, float gamma) { return gamma == 2.f ? x * x : powf(x, gamma); }

// grid: (blocks_per_image, bs).  part[(img*gridDim.x + blk)*3 + {0,1,2}] = {focal sum, smoothL1 sum, #pos}
__global__ __launch_bounds__(kBlock) void retina_fwd_kernel(
    const float* __restrict__ anchors, const float* __restrict__ reg, const float* __restrict__ clas,
    const float* __restrict__ boxes, const int64_t* __restrict__ cats, int32_t* __restrict__ state,
    float* __restrict__ part, int A, int K, int M, float alpha, float gamma) {
  __shared__ float4 s_box[kMaxObj];
  __shared__ float s_area[kMaxObj];
  __shared__ int s_cat[kMaxObj];
  __shared__ int s_m;
  __shared__ float red[4];
  const int img = blockIdx.y;
  // compact the valid objects of this image (Vision.py:1637-1638: rows padded with -1 are dropped), keeping order
  if (threadIdx.x == 0) {
    int m = 0;
    for (int j = 0; j < M && m < kMaxObj; ++j) {
      const int64_t c = cats[(long)img * M + j];
      if (c >= 0) {
        const float4 b = reinterpret_cast<const float4*>(boxes)[(long)img * M + j];
        s_box[m] = b;
        s_area[m] = (b.z - b.x) * (b.w - b.y);
        s_cat[m] = (int)c;
        ++m;
      }
    }
    s_m = m;
  }
  __syncthreads();
  const int m = s_m;
  float focal = 0.f, sl1 = 0.f, npos = 0.f;
  for (int a = blockIdx.x * kBlock + threadIdx.x; a < A; a += gridDim.x * kBlock) {
    const float4 an = reinterpret_cast<const float4*>(anchors)[a];
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
    if (st == -2) continue;
    const int tcls = st >= 0 ? s_cat[st] : -1;
    const float* __restrict__ cp = clas + ((long)img * A + a) * K;
    for (int k = 0; k < K; ++k) {
      const float p = fminf(fmaxf(cp[k], 1e-4f), 1.0f - 1e-4f);
      // Vision.py:1525-1528 with t in {0,1}: pt = p (t=1) or 1-p (t=0); w = wa*(1-pt)^gamma — for t=0 the reference
      // evaluates 1-(1-p) in fp32, which is not bitwise p: keep that form
      const float q = 1.f - p;
      if (k == tcls) focal += -(alpha * powg(q, gamma)) * logf(p);
      else focal += -((1.f - alpha) * powg(1.f - q, gamma)) * logf(q);
    }
    if (st >= 0) {
      npos += 1.f;
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
  focal = block_sum(focal, red);
  sl1 = block_sum(sl1, red);
  npos = block_sum(npos, red);
  if (threadIdx.x == 0) {
    float* o = part + ((long)img * gridDim.x + blockIdx.x) * 3;
    o[0] = focal; o[1] = sl1; o[2] = npos;
  }
}

// out[0] = total loss, out[1] = reg loss, out[2] = clas loss (batch means); npos[img] kept for backward
__global__ void retina_finalize_kernel(const float* __restrict__ part, float* __restrict__ out, float* __restrict__ npos,
                                       int bs, int nblk, float beta) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float reg_sum = 0.f, clas_sum = 0.f;
  for (int i = 0; i < bs; ++i) {
    float f = 0.f, s = 0.f, n = 0.f;
    for (int b = 0; b < nblk; ++b) {
      const float* p = part + ((long)i * nblk + b) * 3;
      f += p[0]; s += p[1]; n += p[2];
    }
    npos[i] = n;
    clas_sum += f / fmaxf(n, 1.f);
    reg_sum += n > 0.f ? s / (n * 4.f) : 0.f;
  }
  const float r = reg_sum / bs, c = clas_sum / bs;
  out[1] = r;
  out[2] = c;
  out[0] = (1.f - beta) * r + beta * c;
}

// gradients wrt reg [bs,A,4] and clas [bs,A,K]; gscale = upstream d(loss) (device scalar pointer)
__global__ __launch_bounds__(kBlock) void retina_bwd_kernel(
    const float* __restrict__ anchors, const float* __restrict__ reg, const float* __restrict__ clas,
    const float* __restrict__ boxes, const int64_t* __restrict__ cats, const int32_t* __restrict__ state,
    const float* __restrict__ npos, const float* __restrict__ gup, float* __restrict__ dreg, float* __restrict__ dclas,
    int A, int K, int M, int bs, float alpha, float gamma, float beta) {
  __shared__ float4 s_box[kMaxObj];
  __shared__ int s_cat[kMaxObj];
  const int img = blockIdx.y;
  if (threadIdx.x == 0) {
    int m = 0;
    for (int j = 0; j < M && m < kMaxObj; ++j) {
      const int64_t c = cats[(long)img * M + j];
      if (c >= 0) { s_box[m] = reinterpret_cast<const float4*>(boxes)[(long)img * M + j]; s_cat[m] = (int)c; ++m; }
    }
  }
  __syncthreads();
  const float g = gup[0];
  const float n = npos[img];
  const float gc = g * beta / (bs * fmaxf(n, 1.f));
  const float gr = n > 0.f ? g * (1.f - beta) / (bs * n * 4.f) : 0.f;
  for (int a = blockIdx.x * kBlock + threadIdx.x; a < A; a += gridDim.x * kBlock) {
    const long ia = (long)img * A + a;
    const int st = state[ia];
    const float* __restrict__ cp = clas + ia * K;
    float* __restrict__ dp = dclas + ia * K;
    if (st == -2) {
      for (int k = 0; k < K; ++k) dp[k] = 0.f;
    } else {
      const int tcls = st >= 0 ? s_cat[st] : -1;
      for (int k = 0; k < K; ++k) {
        const float raw = cp[k];
        float d = 0.f;
        if (raw >= 1e-4f && raw <= 1.0f - 1e-4f) {       // clamp passes gradient only inside its range
          const float p = raw;
          if (k == tcls) {
            // L = -alpha (1-p)^g log p
            const float q = 1.f - p;
            d = alpha * (gamma * powg(q, gamma - 1.f) * logf(p) - powg(q, gamma) / p);
          } else {
            // L = -(1-alpha) p^g log(1-p)
            d = (1.f - alpha) * (-gamma * powg(p, gamma - 1.f) * logf(1.f - p) + powg(p, gamma) / (1.f - p));
          }
        }
        dp[k] = d * gc;
      }
    }
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
}

int blocks_per_image(int A) {
  long b = nnl_cdiv(A, kBlock);
  if (b > 256) b = 256;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" size_t nnl_retina_loss_workspace_bytes(int64_t bs, int64_t A) {
  if (bs <= 0 || A <= 0) return 0;
  return (size_t)(bs * blocks_per_image((int)A) * 3) * sizeof(float);
}

extern "C" int nnl_retina_loss_fwd(const float* anchors, const float* reg, const float* clas, const float* boxes,
                                   const int64_t* cats, int32_t* state, float* npos, float* out, int64_t bs, int64_t A,
                                   int64_t K, int64_t M, float beta, float alpha, float gamma, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  NNL_CHECK_ARG(bs > 0 && A > 0 && K > 0 && M >= 0 && A < (1L << 30) && bs < 65536, "retina_loss_fwd: bad sizes");
  NNL_CHECK_ARG(M <= kMaxObj, "retina_loss_fwd: at most %d objects per image (got %ld)", kMaxObj, (long)M);
  NNL_CHECK_ARG(anchors && reg && clas && state && npos && out && (M == 0 || (boxes && cats)), "retina_loss_fwd: null pointer");
  if (workspace == nullptr || workspace_bytes < nnl_retina_loss_workspace_bytes(bs, A))
    return nnl_set_error(NNL_ERR_WORKSPACE, "retina_loss_fwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int nblk = blocks_per_image((int)A);
  NnlProfScope prof(NNL_PROF_RETINA_LOSS, s, (double)bs * A * (16 + 16 + 4.0 * K + 4));
  hipLaunchKernelGGL(retina_fwd_kernel, dim3(nblk, (unsigned)bs), dim3(kBlock), 0, s, anchors, reg, clas, boxes, cats, state,
                     (float*)workspace, (int)A, (int)K, (int)M, alpha, gamma);
  NNL_CHECK_LAUNCH();
  hipLaunchKernelGGL(retina_finalize_kernel, dim3(1), dim3(64), 0, s, (const float*)workspace, out, npos, (int)bs, nblk, beta);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

extern "C" int nnl_retina_loss_bwd(const float* anchors, const float* reg, const float* clas, const float* boxes,
                                   const int64_t* cats, const int32_t* state, const float* npos, const float* grad_out,
                                   float* dreg, float* dclas, int64_t bs, int64_t A, int64_t K, int64_t M, float beta,
                                   float alpha, float gamma, void* stream) {
  NNL_CHECK_ARG(bs > 0 && A > 0 && K > 0 && M >= 0 && M <= kMaxObj, "retina_loss_bwd: bad sizes");
  NNL_CHECK_ARG(anchors && reg && clas && state && npos && grad_out && dreg && dclas, "retina_loss_bwd: null pointer");
  hipStream_t s = (hipStream_t)stream;
  NnlProfScope prof(NNL_PROF_RETINA_LOSS, s, (double)bs * A * (16 + 32 + 8.0 * K + 4));
  hipLaunchKernelGGL(retina_bwd_kernel, dim3(blocks_per_image((int)A), (unsigned)bs), dim3(kBlock), 0, s, anchors, reg, clas,
                     boxes, cats, state, npos, grad_out, dreg, dclas, (int)A, (int)K, (int)M, (int)bs, alpha, gamma, beta);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}
