// K5 — the recurrence of one (weight-dropped) LSTM layer: WeightDropLSTM1.forward -> nn.LSTM (cuDNN RNN in the
// reference; Applications/Text.py:495-513, :535-551), forward and backward through time.
//
// Round-1 structure: the time loop lives HERE (one C call per layer per direction, no Python per-step overhead);
// each timestep is one fp32-MFMA GEMM launch (igemm_rowk: gates_t = gx_t + h_{t-1} W_hh^T, the `add` epilogue fuses
// the input projection) + one fused pointwise cell kernel.  The input projections for all timesteps (gx) and the
// weight gradients are single large GEMMs outside the loop (ops.py).  Gate order i,f,g,o and the cell equations
// are torch's:  c_t = s(f)*c_{t-1} + s(i)*tanh(g);  h_t = s(o)*tanh(c_t).
// Recurrent GEMM per step: 2*B*4H*H flop (677 MFLOP at B=64,H=1150) against 4H*H*4 B of W_hh (21 MB, L2/MALL
// resident across steps) — latency-bound at this size; a persistent W_hh-resident kernel is the planned upgrade.
#include "igemm.h"

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// gates [B,4H] pre-activation in, activated out; c_prev [B,H]; writes c [B,H], h [B,H] and hpad [B,Hp] (zero pad)
__global__ void lstm_cell_fwd_kernel(float* __restrict__ gates, const float* __restrict__ c_prev, float* __restrict__ c,
                                     float* __restrict__ h, float* __restrict__ hpad, int B, int H, int Hp) {
  const int total = B * Hp;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int b = i / Hp, u = i - b * Hp;
    if (u >= H) { hpad[i] = 0.f; continue; }
    float* g = gates + (long)b * 4 * H;
    const float gi = sigmoidf_(g[u]);
    const float gf = sigmoidf_(g[H + u]);
    const float gg = tanhf(g[2 * H + u]);
    const float go = sigmoidf_(g[3 * H + u]);
    const float cn = gf * c_prev[b * H + u] + gi * gg;
    const float hn = go * tanhf(cn);
    g[u] = gi; g[H + u] = gf; g[2 * H + u] = gg; g[3 * H + u] = go;
    c[b * H + u] = cn;
    h[b * H + u] = hn;
    hpad[i] = hn;
  }
}

__global__ void pad_copy_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int H, int Hp) {
  const int total = B * Hp;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int b = i / Hp, u = i - b * Hp;
    dst[i] = u < H ? src[b * H + u] : 0.f;
  }
}

// dh = dy_t + dh_rec ; in: gates (activated), c_t, c_prev, dc (in/out: dc_next -> dc_prev); out: dgates [B,4H]
__global__ void lstm_cell_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ dh_rec,
                                     const float* __restrict__ gates, const float* __restrict__ c,
                                     const float* __restrict__ c_prev, float* __restrict__ dc,
                                     float* __restrict__ dgates, int B, int H) {
  const int total = B * H;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int b = i / H, u = i - b * H;
    const float* g = gates + (long)b * 4 * H;
    const float gi = g[u], gf = g[H + u], gg = g[2 * H + u], go = g[3 * H + u];
    const float dh = (dy ? dy[i] : 0.f) + (dh_rec ? dh_rec[i] : 0.f);
    const float tc = tanhf(c[i]);
    const float dcn = dc[i] + dh * go * (1.f - tc * tc);
    float* dg = dgates + (long)b * 4 * H;
    dg[u] = dcn * gg * (gi * (1.f - gi));
    dg[H + u] = dcn * c_prev[i] * (gf * (1.f - gf));
    dg[2 * H + u] = dcn * gi * (1.f - gg * gg);
    dg[3 * H + u] = dh * tc * (go * (1.f - go));
    dc[i] = dcn * gf;
  }
}

int ew_grid(long n) {
  long b = nnl_cdiv(n, kBlock);
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" size_t nnl_lstm_workspace_bytes(int64_t B, int64_t H) {
  if (B <= 0 || H <= 0) return 0;
  const long Hp = nnl_cdiv(H, 4) * 4;
  return (size_t)(2 * B * Hp + 2 * B * H) * sizeof(float);      // fwd: 2 padded h buffers; bwd: dh_rec + dc
}

extern "C" int nnl_lstm_fwd(const float* gx, const float* w_hh_pad, const float* h0, const float* c0, float* y, float* cy,
                            float* gates, int64_t T, int64_t B, int64_t H, void* workspace, size_t workspace_bytes,
                            void* stream) {
  NNL_CHECK_ARG(T > 0 && B > 0 && H > 0 && 4 * H < (1 << 24), "lstm_fwd: bad sizes");
  NNL_CHECK_ARG(gx && w_hh_pad && h0 && c0 && y && cy && gates, "lstm_fwd: null pointer");
  if (workspace == nullptr || workspace_bytes < nnl_lstm_workspace_bytes(B, H))
    return nnl_set_error(NNL_ERR_WORKSPACE, "lstm_fwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int Hp = (int)(nnl_cdiv(H, 4) * 4);
  float* hbuf[2] = {(float*)workspace, (float*)workspace + B * Hp};
  const long BH = B * H, BG = B * 4 * H;
  NnlProfScope prof(NNL_PROF_LSTM, s, 2.0 * T * B * 4.0 * H * H);
  hipLaunchKernelGGL(pad_copy_kernel, dim3(ew_grid(B * Hp)), dim3(kBlock), 0, s, h0, hbuf[0], (int)B, (int)H, Hp);
  NNL_CHECK_LAUNCH();
  for (long t = 0; t < T; ++t) {
    float* g_t = gates + t * BG;
    int st = nnl_internal_gemm_nt(hbuf[t & 1], w_hh_pad, g_t, nullptr, gx + t * BG, (int)B, (int)(4 * H), Hp, 0, s);
    if (st) return st;
    const float* c_prev = t == 0 ? c0 : cy + (t - 1) * BH;
    hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3(ew_grid(B * Hp)), dim3(kBlock), 0, s, g_t, c_prev, cy + t * BH, y + t * BH,
                       hbuf[(t + 1) & 1], (int)B, (int)H, Hp);
    NNL_CHECK_LAUNCH();
  }
  return NNL_OK;
}

extern "C" int nnl_lstm_bwd(const float* dy, const float* dhT, const float* dcT, const float* gates, const float* cy,
                            const float* c0, const float* w_hh_t, float* dgates, float* dh0, float* dc0, int64_t T,
                            int64_t B, int64_t H, void* workspace, size_t workspace_bytes, void* stream) {
  NNL_CHECK_ARG(T > 0 && B > 0 && H > 0, "lstm_bwd: bad sizes");
  NNL_CHECK_ARG(gates && cy && c0 && w_hh_t && dgates && dh0 && dc0, "lstm_bwd: null pointer");
  if (workspace == nullptr || workspace_bytes < nnl_lstm_workspace_bytes(B, H))
    return nnl_set_error(NNL_ERR_WORKSPACE, "lstm_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const long BH = B * H, BG = B * 4 * H;
  float* dh_rec = dh0;            // running d loss / d h_{t-1} from the recurrent path, ends as dh0
  float* dc = dc0;                // running d loss / d c_{t-1}, ends as dc0
  NnlProfScope prof(NNL_PROF_LSTM, s, 2.0 * T * B * 4.0 * H * H);
  if (dhT) NNL_CHECK_HIP(hipMemcpyAsync(dh_rec, dhT, sizeof(float) * BH, hipMemcpyDeviceToDevice, s));
  else NNL_CHECK_HIP(hipMemsetAsync(dh_rec, 0, sizeof(float) * BH, s));
  if (dcT) NNL_CHECK_HIP(hipMemcpyAsync(dc, dcT, sizeof(float) * BH, hipMemcpyDeviceToDevice, s));
  else NNL_CHECK_HIP(hipMemsetAsync(dc, 0, sizeof(float) * BH, s));
  for (long t = T - 1; t >= 0; --t) {
    const float* c_prev = t == 0 ? c0 : cy + (t - 1) * BH;
    hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3(ew_grid(BH)), dim3(kBlock), 0, s, dy ? dy + t * BH : nullptr, dh_rec,
                       gates + t * BG, cy + t * BH, c_prev, dc, dgates + t * BG, (int)B, (int)H);
    NNL_CHECK_LAUNCH();
    // dh_{t-1} = dgates_t [B,4H] * W_hh [4H,H]  ==  gemm_nt(a = dgates_t, b = W_hh^T [H,4H])
    int st = nnl_internal_gemm_nt(dgates + t * BG, w_hh_t, dh_rec, nullptr, nullptr, (int)B, (int)H, (int)(4 * H), 0, s);
    if (st) return st;
  }
  return NNL_OK;
}
