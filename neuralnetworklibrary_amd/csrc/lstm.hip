// K5 — the recurrence of one (weight-dropped) LSTM layer: WeightDropLSTM1.forward -> nn.LSTM (cuDNN RNN in the
// reference; Applications/Text.py:495-513, :535-551), forward and backward through time.
//
// The time loop lives HERE (one C call per layer per direction, no Python per-step overhead).  Each timestep needs a skinny
// fp32-MFMA GEMM  h_{t-1} [B,H] x W_hh^T  (fwd)  /  dgates_t [B,4H] x W_hh  (bwd), launched with SPLIT-K so that the 64x64
// output tiles x k-ranges give <= 256 workgroups (the plain tile grid would occupy 18-72 of the 256 CUs), and the pointwise
// cell, which needs the complete sums:
//   forward : ONE launch per timestep (igemm_taps_kernel EPI 1) — tiles hold the 4 gates of 16 hidden units (gathered B
//             rows), slices write fp32 slabs, and the workgroup that takes the last ticket of a tile sums the slabs in slice
//             order (bitwise reproducible) and applies the cell.  Nobody waits for anybody: no deadlock is possible.
//             19.8 us per step against ~30 us for GEMM + separate cell kernel.
//   backward: two launches per timestep (GEMM, then a cell kernel that also sums the slabs).  The fused variant exists
//             (EPI 2, NNL_LSTM_FUSED_BWD=1) but is slower: dh has only 18 output tiles, so the whole pointwise backward
//             would run on 18 CUs (52 us vs ~25 us).
// The input projections for all timesteps (gx) and the weight gradients are single large GEMMs outside the loop
// (ops_text.py).  Gate order i,f,g,o and the cell equations are torch's:
//   c_t = s(f)*c_{t-1} + s(i)*tanh(g);  h_t = s(o)*tanh(c_t).
// Per step: 2*B*4H*H flop (677 MFLOP at B=64, H=1150) against 4H*H*4 B of W_hh (21 MB, L2 / Infinity-Cache resident
// across steps): latency-bound; a persistent W_hh-resident kernel is the planned upgrade.
#include "igemm_taps.h"

// the persistent (one cooperative launch per layer and direction) path: lstm_persist.hip
bool nnl_lstm_persist_ok(long B, long H, long Kp, long Gp);
size_t nnl_lstm_persist_ws_floats(long T, long Kp, long Gp);
hipError_t nnl_lstm_persist_fwd(const float* gx, const float* w_hh_pad, const float* h0, const float* c0, float* y, float* cy,
                                float* gates, long T, long B, long H, long Kp, long Gp, float* ws, int* err, hipStream_t s);
hipError_t nnl_lstm_persist_bwd(const float* dy, const float* dhT, const float* dcT, const float* gates, const float* cy,
                                const float* c0, const float* w_hh_t_pad, float* dgates_pad, float* dh0, float* dc0, long T, long B,
                                long H, long Kp, long Gp, float* ws, int* err, hipStream_t s);

// the 2-D partitioned persistent BPTT (round 4): lstm_bptt2.hip
bool nnl_lstm_bptt2_ok(long B, long H, long Gp);
size_t nnl_lstm_bptt2_ws_floats(long T, long B, long H, long Gp);
hipError_t nnl_lstm_bptt2(const float* dy, const float* dhT, const float* dcT, const float* gates, const float* cy, const float* c0,
                          const float* w_hh_t_pad, float* dgates_pad, float* dh0, float* dc0, long T, long B, long H, long Gp,
                          float* ws, int* err, hipStream_t s);

namespace {

constexpr int kBlock = 256;

// NNL_LSTM_PERSIST: bit 0 = persistent forward, bit 1 = the first persistent backward, bit 2 = the 2-D partitioned persistent
// backward (lstm_bptt2.hip).  Default 5.  The first persistent BPTT kernel is correct (tests/test_text.py runs it) but measured
// slower than the per-timestep pair at H = 1150 (42 vs 20-25 us per step: every workgroup must stream all of dgates_{t+1}, 1.2 MB,
// per step, and its U = 5 output columns leave the 4x4 MFMA chains latency-bound); the 2-D partition takes 15-16 us.
bool persist_fwd_enabled() { return (NNL_ENV_INT("NNL_LSTM_PERSIST", 5) & 1) != 0; }
bool persist_bwd_enabled() { return (NNL_ENV_INT("NNL_LSTM_PERSIST", 5) & 2) != 0; }
// bit 2 = the 2-D partitioned persistent BPTT (lstm_bptt2.hip; takes precedence over bit 1)
bool bptt2_enabled() { return (NNL_ENV_INT("NNL_LSTM_PERSIST", 5) & 4) != 0; }
// (a 2-D partitioned persistent FORWARD — the twin of lstm_bptt2.hip, bit 3 in round 4 — only tied the first persistent forward
// (18.3 vs 18.8 us per step at H = 1150, 14.4 vs 9.5 at H = 400: two hand-overs per step against one) and was removed in round 5)

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__global__ void pad_copy_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int H, int Hp) {
  const int total = B * Hp;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int b = i / Hp, u = i - b * Hp;
    dst[i] = u < H ? src[b * H + u] : 0.f;
  }
}

// acc + p[0] + p[stride] + ... + p[(n-1)*stride], added IN SLAB ORDER (bitwise the same sum as the plain loop) but with up to 16
// loads in flight: the plain loop compiled to load -> s_waitcnt vmcnt(0) -> add per slab, i.e. 14 serialized L2 round trips per
// element — 7 of the 8 us of the BPTT cell kernel (round 3, found in the ISA)
__device__ __forceinline__ float add_slabs(float acc, const float* __restrict__ p, int n, long stride) {
  int s = 0;
  for (; s + 16 <= n; s += 16) {
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = p[(long)(s + j) * stride];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc += v[j];
  }
  if (s + 8 <= n) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = p[(long)(s + j) * stride];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += v[j];
    s += 8;
  }
  if (s + 4 <= n) {
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = p[(long)(s + j) * stride];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc += v[j];
    s += 4;
  }
  for (; s < n; ++s) acc += p[(long)s * stride];
  return acc;
}

// dh = dy_t + sum_s dh_slab[s]; gates (activated), c_t, c_prev; dc in/out (dc_next -> dc_prev); dgates [B,Gp] (Gp >= 4H,
// pad columns are never written and stay zero)
__global__ void lstm_cell_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ dh_slabs, int nslab,
                                     long slab_stride, const float* __restrict__ gates, const float* __restrict__ c,
                                     const float* __restrict__ c_prev, float* __restrict__ dc, float* __restrict__ dgates,
                                     int B, int H, int Gp) {
  const int total = B * H;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int b = i / H, u = i - b * H;
    const float* g = gates + (long)b * 4 * H;
    const float gi = g[u], gf = g[H + u], gg = g[2 * H + u], go = g[3 * H + u];
    float dh = dy ? dy[i] : 0.f;
    dh = add_slabs(dh, dh_slabs + i, nslab, slab_stride);
    const float tc = tanhf(c[i]);
    const float dcn = dc[i] + dh * go * (1.f - tc * tc);
    float* dg = dgates + (long)b * Gp;
    dg[u] = dcn * gg * (gi * (1.f - gi));
    dg[H + u] = dcn * c_prev[i] * (gf * (1.f - gf));
    dg[2 * H + u] = dcn * gi * (1.f - gg * gg);
    dg[3 * H + u] = dh * tc * (go * (1.f - go));
    dc[i] = dcn * gf;
  }
}

__global__ void sum_slabs_kernel(const float* __restrict__ slabs, int nslab, long slab_stride, float* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    out[i] = add_slabs(0.f, slabs + i, nslab, slab_stride);
  }
}

int ew_grid(long n) {
  long b = nnl_cdiv(n, kBlock);
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

long ceil32(long x) { return nnl_cdiv(x, 32) * 32; }

// split count that brings the workgroup count of an [M x N] output (64x64 tiles) with nk k-tiles close to, but not above,
// the 256 CUs: one workgroup per CU is MFMA-bound inside its 4 waves, so a 257th workgroup doubles the step time
// (measured: 288 workgroups 10.8 ms/step of LSTM time, 216-250: 9.7 ms, 576: 11.4 ms)
int pick_splits(long M, long N, long nk) {
  const long tiles = nnl_cdiv(M, 64) * nnl_cdiv(N, 64);
  const long target = NNL_AB_INT("NNL_LSTM_WG", 256);           // tuning hook: workgroup budget
  long s = target / tiles;
  if (s < 1) s = 1;
  if (s > nk) s = nk;
  if (s > 32) s = 32;
  return (int)s;
}

struct Plan { int Hp, Gp, sf, sb, tf, tb; long fwd_slab, bwd_slab; };
Plan make_plan(long B, long H) {
  Plan p;
  p.Hp = (int)ceil32(H);
  p.Gp = (int)ceil32(4 * H);
  p.tf = (int)(nnl_cdiv(B, 64) * nnl_cdiv(H, 16));          // forward tiles: 16 hidden units x 4 gates each
  p.tb = (int)(nnl_cdiv(B, 64) * nnl_cdiv(H, 64));          // backward tiles: 64 hidden units
  p.sf = pick_splits(B, 64 * nnl_cdiv(H, 16), p.Hp / 32);
  p.sb = pick_splits(B, H, p.Gp / 32);
  p.fwd_slab = B * 64 * nnl_cdiv(H, 16);                    // [B][logical columns]
  p.bwd_slab = B * 64 * nnl_cdiv(H, 64);
  return p;
}

// fills the GEMM part of a fused step: y_slabs[ks][M][grid_n*64] = a[M][K] * b[rows][K]^T
IgemmTapsParams step_params(const float* a, const float* b, float* slabs, int M, int Nc, int K, long b_rows, int ks,
                            long slab_stride) {
  IgemmTapsParams q{};
  q.a = a; q.b = b; q.y = slabs; q.bias = nullptr; q.add = nullptr;
  q.a_bytes = (unsigned)((long)M * K * 4); q.b_bytes = (unsigned)(b_rows * K * 4);
  q.H = 1; q.W = 1; q.C = K; q.P = 1; q.Q = 1; q.in_stride = 1; q.ih0 = 0; q.iw0 = 0;
  q.OH = 1; q.OW = 1; q.out_stride = 1; q.oh0 = 0; q.ow0 = 0;
  q.M = M; q.Nc = Nc; q.b_row_stride = K; q.relu = 0; q.ntaps = 1;
  q.tap_dh[0] = 0; q.tap_dw[0] = 0; q.tap_aoff[0] = 0; q.tap_woff[0] = 0;
  q.tap_affine = 1; q.tap_R = 1; q.tap_S = 1; q.tap_dstep = 1;
  q.ksplit = ks; q.slab_stride = slab_stride;
  return q;
}

}  // namespace

extern "C" int64_t nnl_lstm_padded_hidden(int64_t H) { return ceil32(H); }
extern "C" int64_t nnl_lstm_padded_gates(int64_t H) { return ceil32(4 * H); }

// workspace (floats): forward  [2 x B*Hp h ping-pong][sf slabs][T*tf int counters]
//                     backward [sb slabs][B*H][T*tb int counters]
static size_t lstm_ws_floats(const Plan& p, long T, long B, long H) {
  const long fwd = 2 * B * p.Hp + (long)p.sf * p.fwd_slab + T * p.tf;
  const long bwd = (long)p.sb * p.bwd_slab + B * H + T * p.tb;
  size_t n = (size_t)(fwd > bwd ? fwd : bwd);
  if (nnl_lstm_persist_ok(B, H, p.Hp, p.Gp)) {               // the persistent path's per-timestep exchange slots
    const size_t pn = nnl_lstm_persist_ws_floats(T, p.Hp, p.Gp);
    if (pn > n) n = pn;
  }
  const size_t p2 = nnl_lstm_bptt2_ws_floats(T, B, H, p.Gp);     // 0 when the shape does not fit that kernel
  if (p2 > n) n = p2;
  return n;
}

extern "C" size_t nnl_lstm_workspace_bytes(int64_t T, int64_t B, int64_t H) {
  if (T <= 0 || B <= 0 || H <= 0) return 0;
  return lstm_ws_floats(make_plan(B, H), T, B, H) * sizeof(float);
}

extern "C" int nnl_lstm_fwd(const float* gx, const float* w_hh_pad, const float* h0, const float* c0, float* y, float* cy,
                            float* gates, int64_t T, int64_t B, int64_t H, void* workspace, size_t workspace_bytes,
                            int32_t* err_flag, void* stream) {
  NNL_CHECK_ARG(T > 0 && B > 0 && H > 0 && 4 * H < (1 << 24), "lstm_fwd: bad sizes");
  NNL_CHECK_ARG(gx && w_hh_pad && h0 && c0 && y && cy && gates, "lstm_fwd: null pointer");
  if (workspace == nullptr || workspace_bytes < nnl_lstm_workspace_bytes(T, B, H))
    return nnl_set_error(NNL_ERR_WORKSPACE, "lstm_fwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const Plan p = make_plan(B, H);
  float* hbuf[2] = {(float*)workspace, (float*)workspace + B * p.Hp};
  float* slabs = (float*)workspace + 2 * B * p.Hp;
  int* counters = (int*)(slabs + (long)p.sf * p.fwd_slab);
  const long BH = B * H, BG = B * 4 * H;
  NnlProfScope prof(NNL_PROF_LSTM, s, 2.0 * T * B * 4.0 * H * H);
  if (persist_fwd_enabled() && err_flag != nullptr && nnl_lstm_persist_ok(B, H, p.Hp, p.Gp)) {
    // one cooperative launch for the whole sequence; if the runtime refuses it (not co-resident, LDS attribute) nothing has
    // run and the per-timestep path below takes over
    if (nnl_lstm_persist_fwd(gx, w_hh_pad, h0, c0, y, cy, gates, T, B, H, p.Hp, p.Gp, (float*)workspace, err_flag, s) == hipSuccess)
      return NNL_OK;
    (void)hipGetLastError();
  }
  hipLaunchKernelGGL(pad_copy_kernel, dim3(ew_grid(B * p.Hp)), dim3(kBlock), 0, s, h0, hbuf[0], (int)B, (int)H, p.Hp);
  NNL_CHECK_LAUNCH();
  NNL_CHECK_HIP(hipMemsetAsync(hbuf[1], 0, sizeof(float) * B * p.Hp, s));          // pad columns stay zero
  NNL_CHECK_HIP(hipMemsetAsync(counters, 0, sizeof(int) * T * p.tf, s));
  for (long t = 0; t < T; ++t) {
    // one launch per timestep: gates = gx_t + h_{t-1} W_hh^T in sf k slices; the last slice of a tile applies the cell
    IgemmTapsParams q = step_params(hbuf[t & 1], w_hh_pad, slabs, (int)B, (int)(4 * H), p.Hp, 4 * H, p.sf, p.fwd_slab);
    q.lstm.H = (int)H; q.lstm.Hp = p.Hp; q.lstm.Gp = p.Gp;
    q.lstm.counters = counters + t * p.tf;
    q.lstm.gx = gx + t * BG;
    q.lstm.c_prev = t == 0 ? c0 : cy + (t - 1) * BH;
    q.lstm.gates = gates + t * BG;
    q.lstm.c = cy + t * BH;
    q.lstm.h = y + t * BH;
    q.lstm.hpad = hbuf[(t + 1) & 1];
    int st = nnl_internal_lstm_step(q, 1, s);
    if (st) return st;
  }
  return NNL_OK;
}

extern "C" int nnl_lstm_bwd(const float* dy, const float* dhT, const float* dcT, const float* gates, const float* cy,
                            const float* c0, const float* w_hh_t_pad, float* dgates_pad, float* dh0, float* dc0, int64_t T,
                            int64_t B, int64_t H, void* workspace, size_t workspace_bytes, int32_t* err_flag, void* stream) {
  NNL_CHECK_ARG(T > 0 && B > 0 && H > 0, "lstm_bwd: bad sizes");
  NNL_CHECK_ARG(gates && cy && c0 && w_hh_t_pad && dgates_pad && dh0 && dc0, "lstm_bwd: null pointer");
  if (workspace == nullptr || workspace_bytes < nnl_lstm_workspace_bytes(T, B, H))
    return nnl_set_error(NNL_ERR_WORKSPACE, "lstm_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const Plan p = make_plan(B, H);
  const long BH = B * H, BG = B * 4 * H, BGp = B * (long)p.Gp;
  float* slabs = (float*)workspace;              // [sb][B][tiles*64]: split-K partials of dh_{t}
  int* counters = (int*)(slabs + (long)p.sb * p.bwd_slab + BH);
  float* dc = dc0;                               // running d loss / d c_{t-1}, ends as dc0
  NnlProfScope prof(NNL_PROF_LSTM, s, 2.0 * T * B * 4.0 * H * H);
  if (bptt2_enabled() && err_flag != nullptr && nnl_lstm_bptt2_ok(B, H, p.Gp)) {
    if (nnl_lstm_bptt2(dy, dhT, dcT, gates, cy, c0, w_hh_t_pad, dgates_pad, dh0, dc0, T, B, H, p.Gp, (float*)workspace, err_flag, s) ==
        hipSuccess)
      return NNL_OK;
    (void)hipGetLastError();                     // refused (not co-resident / LDS attribute): nothing ran, the paths below take over
  }
  if (persist_bwd_enabled() && err_flag != nullptr && nnl_lstm_persist_ok(B, H, p.Hp, p.Gp)) {
    if (nnl_lstm_persist_bwd(dy, dhT, dcT, gates, cy, c0, w_hh_t_pad, dgates_pad, dh0, dc0, T, B, H, p.Hp, p.Gp, (float*)workspace,
                             err_flag, s) == hipSuccess)
      return NNL_OK;
    (void)hipGetLastError();                     // refused: the per-timestep path below reads the same gates / cy layout
  }
  if (dcT) NNL_CHECK_HIP(hipMemcpyAsync(dc, dcT, sizeof(float) * BH, hipMemcpyDeviceToDevice, s));
  else NNL_CHECK_HIP(hipMemsetAsync(dc, 0, sizeof(float) * BH, s));
  NNL_CHECK_HIP(hipMemsetAsync(counters, 0, sizeof(int) * T * p.tb, s));
  const bool fused_bwd = NNL_ENV_INT("NNL_LSTM_FUSED_BWD", 0) == 1;   // tuning hook; default: two launches per backward timestep
  if (fused_bwd) {
    // t = T-1: no recurrent term from a later step (only dhT, if any): the stand-alone cell kernel
    hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3(ew_grid(BH)), dim3(kBlock), 0, s, dy ? dy + (T - 1) * BH : nullptr, dhT,
                       dhT ? 1 : 0, 0L, gates + (T - 1) * BG, cy + (T - 1) * BH, T == 1 ? c0 : cy + (T - 2) * BH, dc,
                       dgates_pad + (T - 1) * BGp, (int)B, (int)H, p.Gp);
    NNL_CHECK_LAUNCH();
    for (long t = T - 2; t >= 0; --t) {
      // one launch: dh_t = dy_t + dgates_{t+1} [B,Gp] * W_hh [Gp,H] in sb k slices; the last slice of a tile runs the cell
      // backward of its 64 hidden units.  Measured SLOWER than two launches (52 vs ~25 us): only 18 output tiles exist, so
      // the whole pointwise backward lands on 18 CUs.
      IgemmTapsParams q = step_params(dgates_pad + (t + 1) * BGp, w_hh_t_pad, slabs, (int)B, (int)H, p.Gp, H, p.sb, p.bwd_slab);
      q.lstm.H = (int)H; q.lstm.Hp = p.Hp; q.lstm.Gp = p.Gp;
      q.lstm.counters = counters + t * p.tb;
      q.lstm.dy = dy ? dy + t * BH : nullptr;
      q.lstm.gates = const_cast<float*>(gates) + t * BG;
      q.lstm.c = const_cast<float*>(cy) + t * BH;
      q.lstm.c_prev = t == 0 ? c0 : cy + (t - 1) * BH;
      q.lstm.dc = dc;
      q.lstm.dgates = dgates_pad + t * BGp;
      int st = nnl_internal_lstm_step(q, 2, s);
      if (st) return st;
    }
    int st = nnl_internal_gemm_nt_splitk(dgates_pad, w_hh_t_pad, slabs, (int)B, (int)H, p.Gp, p.sb, s);
    if (st) return st;
  } else {
    for (long t = T - 1; t >= 0; --t) {
      const float* c_prev = t == 0 ? c0 : cy + (t - 1) * BH;
      const bool last = t == T - 1;              // recurrent term at the last step = dhT (or nothing)
      hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3(ew_grid(BH)), dim3(kBlock), 0, s, dy ? dy + t * BH : nullptr,
                         last ? dhT : (const float*)slabs, last ? (dhT ? 1 : 0) : p.sb, last ? 0L : BH, gates + t * BG,
                         cy + t * BH, c_prev, dc, dgates_pad + t * BGp, (int)B, (int)H, p.Gp);
      NNL_CHECK_LAUNCH();
      // dh_{t-1} = dgates_t [B,Gp] * W_hh [Gp(4H),H]  ==  gemm_nt(a = dgates_t, b = W_hh^T padded [H,Gp]) in sb k-ranges
      int st = nnl_internal_gemm_nt_splitk(dgates_pad + t * BGp, w_hh_t_pad, slabs, (int)B, (int)H, p.Gp, p.sb, s);
      if (st) return st;
    }
  }
  hipLaunchKernelGGL(sum_slabs_kernel, dim3(ew_grid(BH)), dim3(kBlock), 0, s, (const float*)slabs, p.sb, BH, dh0, BH);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}
