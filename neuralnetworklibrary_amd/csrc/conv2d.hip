// K1 — conv2d forward / dgrad / wgrad as implicit GEMMs on the exact-fp32 MFMA (gfx950).
// Replaces the cuDNN convolutions behind the reference's ResNet blocks, FPN and RetinaNet heads
// (Applications/VisionModels/retinanet.py:26-28,43-59,77-97,126-148,187-217,260-295,304,344-348).
// Layout: activations NHWC, filters KRSC (see include/nnl.h).  MFMA-bound: 2*N*P*Q*K*R*S*C flops per pass.
#include "igemm_kernels.h"

namespace {

int check_geom(const nnl_conv_geom_t* g, const char* who) {
  if (!g) return nnl_set_error(NNL_ERR_INVALID_ARG, "%s: null geometry", who);
  if (g->N <= 0 || g->H <= 0 || g->W <= 0 || g->C <= 0 || g->K <= 0 || g->R <= 0 || g->S <= 0 || g->stride <= 0 ||
      g->pad < 0)
    return nnl_set_error(NNL_ERR_INVALID_ARG, "%s: non-positive dimension", who);
  const int P = (g->H + 2 * g->pad - g->R) / g->stride + 1, Q = (g->W + 2 * g->pad - g->S) / g->stride + 1;
  if (P != g->P || Q != g->Q || P <= 0 || Q <= 0)
    return nnl_set_error(NNL_ERR_INVALID_ARG, "%s: P,Q=(%d,%d) do not match the geometry (%d,%d)", who, g->P, g->Q, P, Q);
  if (g->C % 4 != 0)
    return nnl_set_error(NNL_ERR_UNSUPPORTED, "%s: C=%d must be a multiple of 4 (pad the channels; ops.py does)", who, g->C);
  if ((long)g->N * g->H * g->W * g->C >= (1L << 31) || (long)g->N * g->P * g->Q * g->K >= (1L << 31) ||
      (long)g->N * g->P * g->Q >= (1L << 30))
    return nnl_set_error(NNL_ERR_UNSUPPORTED, "%s: tensor too large for 32-bit row indexing", who);
  return NNL_OK;
}

template <int BM, int BN, int WGM, int WGN, int MODE>
int launch_rowk(IgemmRowkParams p, hipStream_t s) {
  p.grid_m = (int)nnl_cdiv(p.M, BM);
  p.grid_n = (int)nnl_cdiv(p.Nc, BN);
  hipLaunchKernelGGL((igemm_rowk_kernel<BM, BN, 16, WGM, WGN, MODE>), dim3(p.grid_m * p.grid_n), dim3(256), 0, s, p);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

template <int MODE>
int dispatch_rowk(const IgemmRowkParams& p, hipStream_t s) {
  const long b128 = nnl_cdiv(p.M, 128);
  if (p.Nc > 64) {
    if (b128 * nnl_cdiv(p.Nc, 128) >= 400) return launch_rowk<128, 128, 2, 2, MODE>(p, s);
    if (nnl_cdiv(p.M, 64) * nnl_cdiv(p.Nc, 128) >= 400) return launch_rowk<64, 128, 2, 2, MODE>(p, s);
    return launch_rowk<64, 64, 2, 2, MODE>(p, s);
  }
  if (b128 >= 400) return launch_rowk<128, 64, 2, 2, MODE>(p, s);
  return launch_rowk<64, 64, 2, 2, MODE>(p, s);
}

__global__ void weight_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int K, int RS, int C) {
  // w [K][RS][C] -> wt [C][RS][K]; 32x32 LDS tile transpose over (K, C) for each tap
  __shared__ float tile[32][33];
  const int tap = blockIdx.z;
  const int k0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int k = k0 + i, c = c0 + tx;
    tile[i][tx] = (k < K && c < C) ? w[((long)k * RS + tap) * C + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, k = k0 + tx;
    if (k < K && c < C) wt[((long)c * RS + tap) * K + k] = tile[tx][i];
  }
}

__global__ void splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, long n4, int splits) {
  // out[i] = sum_s part[s][i], fixed order => bitwise reproducible; float4 per thread
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 acc = reinterpret_cast<const f32x4*>(part)[i];
    for (int s = 1; s < splits; ++s) acc += reinterpret_cast<const f32x4*>(part)[(long)s * n4 + i];
    reinterpret_cast<f32x4*>(out)[i] = acc;
  }
}

__global__ void colsum_kernel(const float* __restrict__ a, float* __restrict__ out, long rows, int cols) {
  // out[c] = sum_r a[r][c]; one block per 64-column strip, 256 threads = 4 row-lanes x 64 columns
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  float acc = 0.f;
  if (c < cols)
    for (long r = rl; r < rows; r += 4) acc += a[r * cols + c];
  red[rl][threadIdx.x & 63] = acc;
  __syncthreads();
  if (rl == 0 && c < cols) out[c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

struct WgradPlan { int bm, bn, grid_m, grid_n, splits, k_per_split; };

WgradPlan plan_wgrad(int Mc, int Nc, long Kp) {
  WgradPlan pl;
  pl.bm = (Mc >= 128) ? 128 : 64;
  pl.bn = (Nc >= 128 && pl.bm == 128) ? 128 : 64;
  pl.grid_m = (int)nnl_cdiv(Mc, pl.bm);
  pl.grid_n = (int)nnl_cdiv(Nc, pl.bn);
  const long tiles = (long)pl.grid_m * pl.grid_n;
  long splits = nnl_cdiv(1024, tiles);
  const long max_splits = Kp / 256 > 0 ? Kp / 256 : 1;    // at least 256 pixels (16 k-steps) per split
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  long kps = nnl_cdiv(Kp, splits);
  kps = nnl_cdiv(kps, 16) * 16;
  splits = nnl_cdiv(Kp, kps);
  pl.splits = (int)splits;
  pl.k_per_split = (int)kps;
  return pl;
}

}  // namespace

int nnl_internal_gemm_nt(const float* a, const float* b, float* y, const float* bias, const float* add, int M, int N,
                         int K, int relu, hipStream_t s) {
  if (M <= 0 || N <= 0 || K <= 0 || K % 4 != 0) return nnl_set_error(NNL_ERR_INVALID_ARG, "gemm_nt: bad sizes");
  IgemmRowkParams p{};
  p.a = a; p.b = b; p.y = y; p.bias = bias; p.add = add;
  p.N = M; p.H = 1; p.W = 1; p.C = K; p.P = 1; p.Q = 1; p.R = 1; p.S = 1; p.stride = 1; p.pad = 0;
  p.M = M; p.Nc = N; p.Kg = K; p.relu = relu;
  return dispatch_rowk<IGEMM_MODE_FWD>(p, s);
}

size_t nnl_internal_gemm_tn_workspace_bytes(int Mc, int Nc, long Kp) {
  const WgradPlan pl = plan_wgrad(Mc, Nc, Kp);
  return pl.splits > 1 ? (size_t)pl.splits * Mc * Nc * sizeof(float) : 0;
}

int nnl_internal_gemm_tn(const float* a, const float* b, float* y, int Mc, int Nc, long Kp, void* ws, size_t ws_bytes,
                         hipStream_t s) {
  if (Mc <= 0 || Nc <= 0 || Kp <= 0 || Mc % 4 != 0 || Nc % 4 != 0) return nnl_set_error(NNL_ERR_INVALID_ARG, "gemm_tn: bad sizes");
  IgemmKmajorParams p{};
  p.a = a; p.b = b;
  p.N = (int)Kp; p.H = 1; p.W = 1; p.C = Nc; p.P = 1; p.Q = 1; p.R = 1; p.S = 1; p.stride = 1; p.pad = 0;
  p.Mc = Mc; p.Nc = Nc; p.Kp = Kp;
  const WgradPlan pl = plan_wgrad(Mc, Nc, Kp);
  p.grid_m = pl.grid_m; p.grid_n = pl.grid_n; p.splits = pl.splits; p.k_per_split = pl.k_per_split;
  const size_t need = nnl_internal_gemm_tn_workspace_bytes(Mc, Nc, Kp);
  if (need > 0 && (ws == nullptr || ws_bytes < need)) return nnl_set_error(NNL_ERR_WORKSPACE, "gemm_tn: workspace too small");
  p.y = pl.splits > 1 ? (float*)ws : y;
  const dim3 grid(pl.grid_m * pl.grid_n * pl.splits), block(256);
  if (pl.bm == 128 && pl.bn == 128)
    hipLaunchKernelGGL((igemm_kmajor_kernel<128, 128, 16, 2, 2>), grid, block, 0, s, p);
  else if (pl.bm == 128)
    hipLaunchKernelGGL((igemm_kmajor_kernel<128, 64, 16, 2, 2>), grid, block, 0, s, p);
  else
    hipLaunchKernelGGL((igemm_kmajor_kernel<64, 64, 16, 2, 2>), grid, block, 0, s, p);
  NNL_CHECK_LAUNCH();
  if (pl.splits > 1) {
    const long n4 = (long)Mc * Nc / 4;
    int blocks = (int)nnl_cdiv(n4, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, (const float*)ws, y, n4, pl.splits);
    NNL_CHECK_LAUNCH();
  }
  return NNL_OK;
}

extern "C" int nnl_conv2d_fwd(const float* x, const float* w, const float* bias, float* y, const nnl_conv_geom_t* g,
                              int relu, void* stream) {
  int st = check_geom(g, "conv2d_fwd");
  if (st) return st;
  NNL_CHECK_ARG(x && w && y, "conv2d_fwd: null pointer");
  hipStream_t s = (hipStream_t)stream;
  IgemmRowkParams p{};
  p.a = x; p.b = w; p.y = y; p.bias = bias;
  p.N = g->N; p.H = g->H; p.W = g->W; p.C = g->C; p.P = g->P; p.Q = g->Q;
  p.R = g->R; p.S = g->S; p.stride = g->stride; p.pad = g->pad;
  p.M = g->N * g->P * g->Q; p.Nc = g->K; p.Kg = g->R * g->S * g->C; p.relu = relu;
  NnlProfScope prof(NNL_PROF_CONV_FWD, s, 2.0 * p.M * (double)p.Nc * p.Kg);
  return dispatch_rowk<IGEMM_MODE_FWD>(p, s);
}

extern "C" int nnl_conv2d_weight_transpose(const float* w, float* wt, int K, int R, int S, int C, void* stream) {
  NNL_CHECK_ARG(w && wt && K > 0 && R > 0 && S > 0 && C > 0, "conv2d_weight_transpose: bad argument");
  hipStream_t s = (hipStream_t)stream;
  NnlProfScope prof(NNL_PROF_ELEMENTWISE, s, 8.0 * K * R * S * C);
  hipLaunchKernelGGL(weight_transpose_kernel, dim3((unsigned)nnl_cdiv(C, 32), (unsigned)nnl_cdiv(K, 32), R * S), dim3(256), 0, s, w,
                     wt, K, R * S, C);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

extern "C" int nnl_conv2d_dgrad(const float* dy, const float* wt, float* dx, const nnl_conv_geom_t* g, void* stream) {
  int st = check_geom(g, "conv2d_dgrad");
  if (st) return st;
  NNL_CHECK_ARG(dy && wt && dx, "conv2d_dgrad: null pointer");
  NNL_CHECK_ARG(g->K % 4 == 0, "conv2d_dgrad: K=%d must be a multiple of 4", g->K);
  hipStream_t s = (hipStream_t)stream;
  IgemmRowkParams p{};
  p.a = dy; p.b = wt; p.y = dx; p.bias = nullptr;
  p.N = g->N; p.H = g->P; p.W = g->Q; p.C = g->K;          // A source = dy [N][P][Q][K]
  p.P = g->H; p.Q = g->W;                                  // GEMM rows enumerate dx pixels
  p.R = g->R; p.S = g->S; p.stride = g->stride; p.pad = g->pad;
  p.M = g->N * g->H * g->W; p.Nc = g->C; p.Kg = g->R * g->S * g->K; p.relu = 0;
  NnlProfScope prof(NNL_PROF_CONV_DGRAD, s, 2.0 * g->N * (double)g->P * g->Q * g->K * g->R * g->S * g->C);
  return dispatch_rowk<IGEMM_MODE_DGRAD>(p, s);
}

extern "C" size_t nnl_conv2d_wgrad_workspace_bytes(const nnl_conv_geom_t* g) {
  if (!g || g->K <= 0 || g->C <= 0) return 0;
  const WgradPlan pl = plan_wgrad(g->K, g->R * g->S * g->C, (long)g->N * g->P * g->Q);
  return pl.splits > 1 ? (size_t)pl.splits * g->K * g->R * g->S * g->C * sizeof(float) : 0;
}

extern "C" int nnl_conv2d_wgrad(const float* x, const float* dy, float* dw, const nnl_conv_geom_t* g, void* workspace,
                                size_t workspace_bytes, void* stream) {
  int st = check_geom(g, "conv2d_wgrad");
  if (st) return st;
  NNL_CHECK_ARG(x && dy && dw, "conv2d_wgrad: null pointer");
  NNL_CHECK_ARG(g->K % 4 == 0, "conv2d_wgrad: K=%d must be a multiple of 4", g->K);
  hipStream_t s = (hipStream_t)stream;
  IgemmKmajorParams p{};
  p.a = dy; p.b = x;
  p.N = g->N; p.H = g->H; p.W = g->W; p.C = g->C; p.P = g->P; p.Q = g->Q;
  p.R = g->R; p.S = g->S; p.stride = g->stride; p.pad = g->pad;
  p.Mc = g->K; p.Nc = g->R * g->S * g->C; p.Kp = (long)g->N * g->P * g->Q;
  const WgradPlan pl = plan_wgrad(p.Mc, p.Nc, p.Kp);
  p.grid_m = pl.grid_m; p.grid_n = pl.grid_n; p.splits = pl.splits; p.k_per_split = pl.k_per_split;
  const size_t need = nnl_conv2d_wgrad_workspace_bytes(g);
  if (need > 0 && (workspace == nullptr || workspace_bytes < need))
    return nnl_set_error(NNL_ERR_WORKSPACE, "conv2d_wgrad: workspace %zu B < required %zu B", workspace_bytes, need);
  p.y = pl.splits > 1 ? (float*)workspace : dw;
  NnlProfScope prof(NNL_PROF_CONV_WGRAD, s, 2.0 * p.Kp * (double)p.Mc * p.Nc);
  const dim3 grid(pl.grid_m * pl.grid_n * pl.splits), block(256);
  if (pl.bm == 128 && pl.bn == 128)
    hipLaunchKernelGGL((igemm_kmajor_kernel<128, 128, 16, 2, 2>), grid, block, 0, s, p);
  else if (pl.bm == 128)
    hipLaunchKernelGGL((igemm_kmajor_kernel<128, 64, 16, 2, 2>), grid, block, 0, s, p);
  else
    hipLaunchKernelGGL((igemm_kmajor_kernel<64, 64, 16, 2, 2>), grid, block, 0, s, p);
  NNL_CHECK_LAUNCH();
  if (pl.splits > 1) {
    const long n4 = (long)p.Mc * p.Nc / 4;                 // Nc = R*S*C with C % 4 == 0
    int blocks = (int)nnl_cdiv(n4, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, (const float*)workspace, dw, n4, pl.splits);
    NNL_CHECK_LAUNCH();
  }
  return NNL_OK;
}

extern "C" int nnl_colsum(const float* a, float* out, int64_t rows, int64_t cols, void* stream) {
  NNL_CHECK_ARG(a && out && rows >= 0 && cols > 0 && cols < (1L << 30), "colsum: bad argument");
  hipStream_t s = (hipStream_t)stream;
  NnlProfScope prof(NNL_PROF_ELEMENTWISE, s, 4.0 * rows * cols);
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)nnl_cdiv(cols, 64)), dim3(256), 0, s, a, out, (long)rows, (int)cols);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}
