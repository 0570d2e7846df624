// Pooling kernels of the vision path, NHWC, HBM-bound:
//   * MaxPool2d (the ResNet stem's 3x3 / stride 2 / pad 1: reference retinanet.py:307,374; torchvision resnet.maxpool) with
//     torch's tie rule (the FIRST maximum in (kh, kw) scan order wins; NaN propagates) and a gather-style backward — every
//     input pixel looks at the <= ceil(k/s)^2 windows that contain it, so there are no atomics and the result is bitwise
//     reproducible;
//   * AdaptiveConcatPool2d (reference General/Layers.py:78-87): cat([AdaptiveMaxPool2d(1), AdaptiveAvgPool2d(1)], 1) as one
//     pass over the [N, HW, C] activation; the max gradient goes to the first arg-max pixel (torch's adaptive_max_pool2d
//     backward), the mean gradient to every pixel.
// Algorithmic bytes: maxpool fwd 4*(in + out) + out (uint8 window index), bwd 4*(in + ~2.25*out) ; concat-pool 4*in each way.
#include "nnl_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          uint8_t* __restrict__ idx, int N, int H, int W, int C4, int P,
                                                          int Q, int ks, int stride, int pad) {
  // one thread = one output pixel x 4 channels
  const long total = (long)N * P * Q * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    long r = i / C4;
    const int q = (int)(r % Q); r /= Q;
    const int p = (int)(r % P);
    const int n = (int)(r / P);
    const float ninf = -__builtin_inff();
    f32x4 best = {ninf, ninf, ninf, ninf};
    int bi[4] = {-1, -1, -1, -1};
    for (int kh = 0; kh < ks; ++kh) {
      const int h = p * stride - pad + kh;
      if ((unsigned)h >= (unsigned)H) continue;
      for (int kw = 0; kw < ks; ++kw) {
        const int w = q * stride - pad + kw;
        if ((unsigned)w >= (unsigned)W) continue;
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[((long)(n * H + h) * W + w) * C4 + c4];
        const int t = kh * ks + kw;
#pragma unroll
        for (int e = 0; e < 4; ++e)                                   // torch: `if ((val > maxval) || isnan(val))` in scan order,
          if (bi[e] < 0 || v[e] > best[e] || v[e] != v[e]) {           // starting from the first in-bounds tap
            best[e] = v[e];
            bi[e] = t;
          }
      }
    }
    reinterpret_cast<f32x4*>(y)[i] = best;
    reinterpret_cast<uchar4*>(idx)[i] = make_uchar4((uint8_t)bi[0], (uint8_t)bi[1], (uint8_t)bi[2], (uint8_t)bi[3]);
  }
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                          float* __restrict__ dx, int N, int H, int W, int C4, int P, int Q,
                                                          int ks, int stride, int pad) {
  // one thread = one INPUT pixel x 4 channels: sum dy over the windows whose arg-max is this pixel
  const long total = (long)N * H * W * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    long r = i / C4;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H);
    const int n = (int)(r / H);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // windows p with p*stride - pad <= h <= p*stride - pad + ks - 1
    int p_lo = h + pad - ks + 1; p_lo = p_lo > 0 ? (p_lo + stride - 1) / stride : 0;
    int p_hi = (h + pad) / stride; if (p_hi > P - 1) p_hi = P - 1;
    int q_lo = w + pad - ks + 1; q_lo = q_lo > 0 ? (q_lo + stride - 1) / stride : 0;
    int q_hi = (w + pad) / stride; if (q_hi > Q - 1) q_hi = Q - 1;
    for (int p = p_lo; p <= p_hi; ++p) {
      const int kh = h + pad - p * stride;
      for (int q = q_lo; q <= q_hi; ++q) {
        const int t = kh * ks + (w + pad - q * stride);
        const long o = ((long)(n * P + p) * Q + q) * C4 + c4;
        const uchar4 id = reinterpret_cast<const uchar4*>(idx)[o];
        const f32x4 g = reinterpret_cast<const f32x4*>(dy)[o];
        if (id.x == t) acc[0] += g[0];
        if (id.y == t) acc[1] += g[1];
        if (id.z == t) acc[2] += g[2];
        if (id.w == t) acc[3] += g[3];
      }
    }
    reinterpret_cast<f32x4*>(dx)[i] = acc;
  }
}

// block = 256 threads = 64 channel lanes x 4 pixel lanes; grid = (ceil(C/64), N)
__global__ __launch_bounds__(256) void concat_pool_fwd_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                              int32_t* __restrict__ argmax, int HW, int C) {
  __shared__ float smax[4][64], ssum[4][64];
  __shared__ int sidx[4][64], snan[4][64];
  const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl, n = blockIdx.y;
  // torch scans the pixels in order with `if ((val > max) || isnan(val))`: without NaNs the FIRST maximum wins, with NaNs the
  // LAST NaN does.  Each of the 4 pixel lanes keeps (max of its non-NaN values, first index of it, last NaN index).
  float m = -__builtin_inff(), s = 0.f;
  int mi = -1, ni = -1;
  if (c < C)
    for (int p = pl; p < HW; p += 4) {
      const float v = x[((long)n * HW + p) * C + c];
      s += v;
      if (v != v) ni = p;
      else if (mi < 0 || v > m) { m = v; mi = p; }
    }
  smax[pl][cl] = m; ssum[pl][cl] = s; sidx[pl][cl] = mi; snan[pl][cl] = ni;
  __syncthreads();
  if (pl == 0 && c < C) {
    float bm = smax[0][cl]; int bi = sidx[0][cl], bn = snan[0][cl]; float tot = ssum[0][cl];
#pragma unroll
    for (int l = 1; l < 4; ++l) {
      const float v = smax[l][cl]; const int vi = sidx[l][cl];
      tot += ssum[l][cl];
      bn = max(bn, snan[l][cl]);
      if (vi >= 0 && (bi < 0 || v > bm || (v == bm && vi < bi))) { bm = v; bi = vi; }
    }
    if (bi < 0) bi = 0;
    // -inf-only channels: torch keeps its initial index (pixel 0) because nothing compares greater than -inf
    if (bm == -__builtin_inff()) bi = 0;
    out[(long)n * 2 * C + c] = bn >= 0 ? __builtin_nanf("") : bm;
    out[(long)n * 2 * C + C + c] = tot / (float)HW;
    argmax[(long)n * C + c] = bn >= 0 ? bn : bi;
  }
}

__global__ __launch_bounds__(256) void concat_pool_bwd_kernel(const float* __restrict__ dout, const int32_t* __restrict__ argmax,
                                                              float* __restrict__ dx, int HW, int C) {
  const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl, n = blockIdx.y;
  if (c >= C) return;
  const float gmax = dout[(long)n * 2 * C + c];
  const float gavg = dout[(long)n * 2 * C + C + c] / (float)HW;
  const int am = argmax[(long)n * C + c];
  for (int p = pl; p < HW; p += 4) dx[((long)n * HW + p) * C + c] = gavg + (p == am ? gmax : 0.f);
}

int ew_blocks(long total) {
  long b = nnl_cdiv(total, 256);
  if (b > 65536) b = 65536;
  return (int)(b < 1 ? 1 : b);
}

}  // namespace

extern "C" int nnl_maxpool2d_fwd(const float* x, float* y, uint8_t* idx, int64_t N, int64_t H, int64_t W, int64_t C, int64_t P,
                                 int64_t Q, int ksize, int stride, int pad, void* stream) {
  NNL_CHECK_ARG(x && y && idx && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "maxpool2d_fwd: bad argument (C %% 4 == 0)");
  NNL_CHECK_ARG(ksize >= 1 && ksize <= 15 && stride >= 1 && pad >= 0 && 2 * pad <= ksize, "maxpool2d_fwd: bad window");
  NNL_CHECK_ARG(P == (H + 2 * pad - ksize) / stride + 1 && Q == (W + 2 * pad - ksize) / stride + 1, "maxpool2d_fwd: bad P/Q");
  NNL_CHECK_ARG(N * H * W * C < (1L << 40), "maxpool2d_fwd: tensor too large");
  hipStream_t s = (hipStream_t)stream;
  const long total = N * P * Q * (C / 4);
  NnlProfScope prof(NNL_PROF_ELEMENTWISE, s, 4.0 * N * H * W * C + 5.0 * N * P * Q * C);
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, s, x, y, idx, (int)N, (int)H, (int)W, (int)(C / 4),
                     (int)P, (int)Q, ksize, stride, pad);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

extern "C" int nnl_maxpool2d_bwd(const float* dy, const uint8_t* idx, float* dx, int64_t N, int64_t H, int64_t W, int64_t C,
                                 int64_t P, int64_t Q, int ksize, int stride, int pad, void* stream) {
  NNL_CHECK_ARG(dy && dx && idx && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "maxpool2d_bwd: bad argument (C %% 4 == 0)");
  NNL_CHECK_ARG(ksize >= 1 && ksize <= 15 && stride >= 1 && pad >= 0 && 2 * pad <= ksize, "maxpool2d_bwd: bad window");
  NNL_CHECK_ARG(P == (H + 2 * pad - ksize) / stride + 1 && Q == (W + 2 * pad - ksize) / stride + 1, "maxpool2d_bwd: bad P/Q");
  hipStream_t s = (hipStream_t)stream;
  const long total = N * H * W * (C / 4);
  NnlProfScope prof(NNL_PROF_ELEMENTWISE, s, 4.0 * N * H * W * C + 5.0 * N * P * Q * C);
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, s, dy, idx, dx, (int)N, (int)H, (int)W, (int)(C / 4),
                     (int)P, (int)Q, ksize, stride, pad);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

extern "C" int nnl_concat_pool_fwd(const float* x, float* out, int32_t* argmax, int64_t N, int64_t HW, int64_t C, void* stream) {
  NNL_CHECK_ARG(x && out && argmax && N > 0 && HW > 0 && C > 0 && N < 65536 && HW < (1L << 30), "concat_pool_fwd: bad argument");
  hipStream_t s = (hipStream_t)stream;
  NnlProfScope prof(NNL_PROF_ELEMENTWISE, s, 4.0 * N * HW * C);
  hipLaunchKernelGGL(concat_pool_fwd_kernel, dim3((unsigned)nnl_cdiv(C, 64), (unsigned)N), dim3(256), 0, s, x, out, argmax, (int)HW,
                     (int)C);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

extern "C" int nnl_concat_pool_bwd(const float* dout, const int32_t* argmax, float* dx, int64_t N, int64_t HW, int64_t C,
                                   void* stream) {
  NNL_CHECK_ARG(dout && dx && argmax && N > 0 && HW > 0 && C > 0 && N < 65536 && HW < (1L << 30), "concat_pool_bwd: bad argument");
  hipStream_t s = (hipStream_t)stream;
  NnlProfScope prof(NNL_PROF_ELEMENTWISE, s, 4.0 * N * HW * C);
  hipLaunchKernelGGL(concat_pool_bwd_kernel, dim3((unsigned)nnl_cdiv(C, 64), (unsigned)N), dim3(256), 0, s, dout, argmax, dx,
                     (int)HW, (int)C);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}


// ---- gradient of a nearest-neighbour x2 upsampling (FPN top-down path, reference retinanet.py:131-141: nn.Upsample(scale_factor=2)) ----
// dsmall[n][h][w][c] = dy[n][2h][2w][c] + dy[n][2h][2w+1][c] + dy[n][2h+1][2w][c] + dy[n][2h+1][2w+1][c]   (torch's order)
// HBM-bound: 4 B read per dy element + 1 B written; 16-B accesses, one float4 of channels per thread.
namespace {
__global__ __launch_bounds__(256) void upsample2_bwd_kernel(const float* __restrict__ dy, float* __restrict__ ds, long total4, int h, int w, int C4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    long r = i / C4;
    const int x = (int)(r % w); r /= w;
    const int y = (int)(r % h);
    const long n = r / h;
    const f32x4* src = reinterpret_cast<const f32x4*>(dy) + ((n * 2 * h + 2 * y) * 2 * w + 2 * x) * C4 + c;
    const f32x4 a = src[0], b = src[C4], cc = src[(long)2 * w * C4], d = src[(long)2 * w * C4 + C4];
    reinterpret_cast<f32x4*>(ds)[i] = ((a + b) + cc) + d;
  }
}
}  // namespace

extern "C" int nnl_upsample2_bwd(const float* dy, float* dsmall, int64_t N, int64_t h, int64_t w, int64_t C, void* stream) {
  NNL_CHECK_ARG(dy && dsmall && N > 0 && h > 0 && w > 0 && C > 0 && C % 4 == 0, "upsample2_bwd: bad argument (C %% 4 == 0)");
  NNL_CHECK_ARG(N * h * w * C * 4 < (1L << 40), "upsample2_bwd: tensor too large");
  hipStream_t s = (hipStream_t)stream;
  const long total4 = N * h * w * (C / 4);
  NnlProfScope prof(NNL_PROF_ELEMENTWISE, s, 20.0 * N * h * w * C);
  long blocks = nnl_cdiv(total4, 256);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(upsample2_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dy, dsmall, total4, (int)h, (int)w, (int)(C / 4));
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}
