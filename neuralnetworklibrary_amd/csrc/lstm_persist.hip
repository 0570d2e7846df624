// K5 — PERSISTENT recurrence of one LSTM layer (forward and BPTT): one cooperative launch runs all T timesteps.
// Replaces the per-timestep launches of lstm.hip when the layer fits (B <= 64, the workgroup's W_hh slice fits the LDS):
// WeightDropLSTM1.forward -> nn.LSTM (cuDNN's persistent RNN in the reference; Applications/Text.py:495-513, :535-551).
//
// Partition.  NWG = ceil(H / U) <= 256 workgroups, one per CU (co-resident: cooperative launch), U = ceil(H / 256) hidden units
// each (H = 1150: 230 workgroups x 5 units).  A workgroup keeps, for ALL timesteps, in its LDS:
//   forward : the 4U rows of W_hh that produce the i, f, g, o gates of its units   (4U x H floats: 92 KB at H = 1150)
//   backward: its U columns of W_hh (rows of W_hh^T)                               (U x 4H floats: 92 KB)
// and in registers the cell state c (forward) / the running dc (backward) of its units.  W_hh is read from HBM once per layer
// instead of once per timestep (21 MB x 70).
//
// Per timestep a workgroup needs the WHOLE previous h_{t-1} [B, H] (forward) / dgates_{t+1} [B, 4H] (backward).  These live in
// a k-major exchange buffer with ONE SLOT PER TIMESTEP (xT [T+1][K][64], 256 B per k): every address is written exactly once in
// the launch — by the owner of that k, with agent-scope (write-through) stores — and read only after that step's grid barrier,
// so plain cached loads can never meet a stale line, and the ~29 workgroups of an XCD share the step's slot through their L2.
// Grid barrier per step: drain stores (vmcnt 0), one agent-scope atomic add on arrive[t], spin on an agent-scope load (bounded:
// a workgroup that waits longer than ~seconds raises *err = 2 and every workgroup leaves — the grid always drains).
//
// MFMA.  The step is a skinny GEMM out[b][col] = sum_k x[b][k] W[col][k] with 64 batch rows and only 4U = 20 (forward) or U = 5
// (backward) columns per workgroup: a 32x32 or 16x16 tile would waste 37 % - 69 % of the matrix core on padding.
// v_mfma_f32_4x4x1_16b_f32 with A-broadcast fits exactly: block = lane / 4 (16 blocks = 16 x 4 batch rows), B operand = x[lane][k]
// (one coalesced 256-B row of the k-major slot per k), A operand = one register holding W[k][0..63] whose lanes 4g..4g+3 are
// broadcast to all blocks by CBSZ = 4 / ABID = g: five MFMAs per k cover 20 columns from ONE LDS read, no padding FLOPs.
// (Lane mapping measured with tools/mfma4x4_probe.hip: D[vgpr v][lane l] += A[lane 4g + v] * B[lane l].)  The four waves split k;
// their partial sums meet in LDS; the cell runs on (unit, batch) pairs, one or two per thread, in fixed order => bitwise
// reproducible, no atomics on data.
#include "nnl_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef NNL_TAPS_TIMING
// timing builds only (tools/build_timing_lib.sh): per timestep, workgroups 0 / 77 / 153 / 229 record six 100 MHz timestamps
// (step start, k loop done, h published, arrived, tape stores issued, barrier passed) — tools/lstm_timing.py reads them back
__device__ unsigned long long g_lstm_stamps[4 * 128 * 6];
extern "C" int nnl_debug_lstm_stamps(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_lstm_stamps), sizeof(unsigned long long) * n, 0, hipMemcpyDeviceToHost);
}
#define NNL_LSTAMP(i) do { if (dbg_slot >= 0 && threadIdx.x == 0 && t < 128) g_lstm_stamps[(dbg_slot * 128 + t) * 6 + (i)] = wall_clock64(); } while (0)
#else
#define NNL_LSTAMP(i) do { } while (0)
#endif

namespace {

constexpr int kBlock = 256;
constexpr int kLanes = 64;         // batch rows per workgroup = lanes of a wave
constexpr int kMaxU = 8;           // units per workgroup: 4U <= 32 gate columns (8 MFMA column groups)
constexpr int kSpinLimit = 1 << 22;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

struct PersistFwd {
  const float* gx;      // [T][B][4H]
  const float* w;       // [4H][Kp]  W_hh, k padded with zeros
  const float* c0;      // [B][H]
  float* y;             // [T][B][H]
  float* cy;            // [T][B][H]
  float* gates;         // [T][B][4H] activated i, f, g, o (saved for backward)
  float* xT;            // [T+1][Kp][64] k-major h: slot 0 = h0 (written by the prologue kernel), slot t+1 = h_t
  int* arrive;          // [T] zero on entry
  int* err;
  int T, B, H, Kp, U, NWG;
  int dbg;              // timing experiments only (NNL_LSTM_DBG): 1 = no k loop, 2 = no grid wait
  int pd;               // k groups per chunk of the forward k loop (NNL_LSTM_PD: 6 / 9 / 12 / 18; default 12)
  int single;           // 1: the single-chunk k loop where it is instantiated (NNL_LSTM_SINGLE, default 1)
};

struct PersistBwd {
  const float* dy;      // [T][B][H] or null
  const float* dhT;     // [B][H] or null
  const float* dcT;     // [B][H] or null
  const float* gates;   // [T][B][4H] activated gates of the forward
  const float* cy;      // [T][B][H]
  const float* c0;      // [B][H]
  const float* wt;      // [H][Gp]  W_hh^T, k (= gate column) padded with zeros
  float* dgates;        // [T][B][Gp] row-major (what the weight-gradient GEMMs read)
  float* dgT;           // [T][Gp][64] k-major exchange
  float* dh0;           // [B][H]
  float* dc0;           // [B][H]
  int* arrive;          // [T] zero on entry
  int* err;
  int T, B, H, Gp, U, NWG;
};

// acc[g][v] (lane l) += sum over the wave's nkg k-groups (4 k each, starting at group kg0) of W[col 4g+v][k] * x[l][k].
// xs = one k-major slot [K][64] (rows >= the real K are zero: written by the prologue kernel), wl = LDS [K/4][NCOL][4].
// kg0 / nkg / rot are WAVE-UNIFORM (scalar registers): every address is "scalar base + lane offset + immediate", there is no
// per-lane predicate and no 64-bit vector address arithmetic in the loop (the first version computed both per lane: 730 cycles
// per k-group against ~200 for its 20 MFMAs).
// The k range is walked in chunks of PD groups starting at chunk `rot` (a per-workgroup constant): a slot was written with
// write-through stores by workgroups on every XCD, so the FIRST reader of a line in an XCD misses its L2; rotated starts spread
// those misses over the whole slot at once and everybody else's chunks are L2 hits.  The order is fixed per workgroup => results
// stay bitwise reproducible.
template <int NG, int NCOL, int PD = 12>                  // PD k groups per chunk: 4 PD rows of 256 B in flight per wave, twice
__device__ __forceinline__ void panel(const float* __restrict__ xs, const float* wl, int kg0, int nkg, int ln, int wcol, int rot,
                                      f32x4 (&acc)[NG]) {
  float cur[PD][4], nxt[PD][4];
  const int nfull = nkg / PD;
  const float* xb = xs + (long)kg0 * 4 * kLanes + ln;
  const float* wb = wl + ((long)kg0 * NCOL + wcol) * 4;
  auto fetch = [&](float (&dst)[PD][4], int c) {
    const float* q = xb + (long)c * PD * 4 * kLanes;
#pragma unroll
    for (int j = 0; j < PD; ++j)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) dst[j][kk] = q[(j * 4 + kk) * kLanes];
  };
  auto mfma4 = [&](const f32x4 a4, const float (&b4)[4]) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const float a = a4[kk], b = b4[kk];
#define NNL_MFMA_G(G) if constexpr (NG > G) acc[G] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[G], 4, G, 0);
      NNL_MFMA_G(0) NNL_MFMA_G(1) NNL_MFMA_G(2) NNL_MFMA_G(3) NNL_MFMA_G(4) NNL_MFMA_G(5) NNL_MFMA_G(6) NNL_MFMA_G(7)
#undef NNL_MFMA_G
    }
  };
  if (nfull > 0) {
    int c = rot % nfull;
    fetch(cur, c);
    for (int i = 0; i < nfull; ++i) {
      const int cn = c + 1 == nfull ? 0 : c + 1;
      if (i + 1 < nfull) fetch(nxt, cn);
      const float* wq = wb + (long)c * PD * NCOL * 4;
      f32x4 aw[PD];                                       // the chunk's W pieces: all LDS reads issued ahead of the MFMAs
#pragma unroll
      for (int j = 0; j < PD; ++j) aw[j] = *reinterpret_cast<const f32x4*>(wq + j * NCOL * 4);
#pragma unroll
      for (int j = 0; j < PD; ++j) mfma4(aw[j], cur[j]);
#pragma unroll
      for (int j = 0; j < PD; ++j)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) cur[j][kk] = nxt[j][kk];
      c = cn;
    }
  }
  for (int kg = nfull * PD; kg < nkg; ++kg) {             // the ragged rest of the wave's range
    float b4[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) b4[kk] = xb[((long)kg * 4 + kk) * kLanes];
    mfma4(*reinterpret_cast<const f32x4*>(wb + (long)kg * NCOL * 4), b4);
  }
}

// Single-chunk variant (round 3): ALL of the wave's k rows are requested at once — PDS groups of 4 rows, one dword per lane and row
// (PDS = 36 at H = 1150: 144 loads, 144 VGPRs; the kernel runs one 512-thread workgroup per CU, i.e. 256 VGPRs per wave) — and the
// MFMAs consume them in issue order while the rest is still in flight.  The chunked loop above exposes one L2 / Infinity-Cache
// round trip per 48 rows: the timestamp build (tools/lstm_timing.py) put the k loop at 11.9 of the 16.9 us of a timestep.  The W
// pieces come from LDS through a small register ring (RW ahead).  Same k order as the chunked loop with rot = 0 => bitwise
// reproducible.
template <int NG, int NCOL, int PDS>
__device__ __forceinline__ void panel_single(const float* __restrict__ xs, const float* wl, int kg0, int ln, int wcol, f32x4 (&acc)[NG]) {
  constexpr int RW = 6;
  float xb_[PDS][4];
  const float* xb = xs + (long)kg0 * 4 * kLanes + ln;
  const float* wb = wl + ((long)kg0 * NCOL + wcol) * 4;
#pragma unroll
  for (int j = 0; j < PDS; ++j)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) xb_[j][kk] = xb[(j * 4 + kk) * kLanes];
  f32x4 aw[RW];
#pragma unroll
  for (int j = 0; j < RW && j < PDS; ++j) aw[j] = *reinterpret_cast<const f32x4*>(wb + j * NCOL * 4);
#pragma unroll
  for (int j = 0; j < PDS; ++j) {
    const f32x4 a4 = aw[j % RW];
    if (j + RW < PDS) aw[j % RW] = *reinterpret_cast<const f32x4*>(wb + (j + RW) * NCOL * 4);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const float a = a4[kk], b = xb_[j][kk];
#define NNL_MFMA_G(G) if constexpr (NG > G) acc[G] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[G], 4, G, 0);
      NNL_MFMA_G(0) NNL_MFMA_G(1) NNL_MFMA_G(2) NNL_MFMA_G(3) NNL_MFMA_G(4) NNL_MFMA_G(5) NNL_MFMA_G(6) NNL_MFMA_G(7)
#undef NNL_MFMA_G
    }
  }
}

// the workgroup's slice of a [rows][Kp] matrix -> LDS [Kp/4][ncol][4]; row_of(col) = source row (or -1: zeros)
template <typename RowOf>
__device__ __forceinline__ void load_slice(const float* __restrict__ w, int Kp, int ncol, float* wl, RowOf row_of) {
  const int kg_n = Kp / 4;
  for (int i = threadIdx.x; i < ncol * kg_n; i += blockDim.x) {
    const int col = i / kg_n, kg = i - col * kg_n;        // consecutive threads: consecutive 16-B pieces of one source row
    const long r = row_of(col);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r >= 0) v = *reinterpret_cast<const f32x4*>(w + r * Kp + 4 * kg);
    *reinterpret_cast<f32x4*>(wl + ((long)kg * ncol + col) * 4) = v;
  }
}

// Grid barrier for step `slot`, in two halves so that the stores nobody waits for (y, cy, gates) overlap the wait:
//   grid_arrive: called right after the exchange stores — drains them (vmcnt 0; later stores are issued after it), one arrival;
//   grid_wait  : spins until every workgroup has arrived; returns false when the launch aborts (bounded spin).
__device__ __forceinline__ void grid_arrive(int* arrive, int slot) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(arrive + slot, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ bool grid_wait(int* arrive, int slot, int nwg, int* err, int* s_flag) {
  if (threadIdx.x == 0) {
    int ok = 1;
    for (int it = 0; __hip_atomic_load(arrive + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nwg; ++it) {
      __builtin_amdgcn_s_sleep(1);
      if (it > kSpinLimit) { __hip_atomic_store(err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = 0; break; }
      if ((it & 255) == 255 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2) { ok = 0; break; }
    }
    *s_flag = ok;
  }
  __syncthreads();
  return *s_flag != 0;
}

// Forward: kFwdWaves = 8 waves (two per SIMD: one wave per SIMD leaves the 4x4x1 MFMA issue latency- and LDS-bound, measured 10 us
// for the k loop at H = 1150 against ~5 at full rate), each taking 1/8 of k.
constexpr int kFwdWaves = 8, kFwdBlock = kFwdWaves * 64;

template <int NG>   // NG = U column groups of 4 gate columns: column = gate * U + unit
__global__ __launch_bounds__(kFwdBlock) void lstm_persist_fwd_kernel(PersistFwd p) {
  extern __shared__ float lds[];
  __shared__ int s_flag;
  const int U = p.U, ncol = 4 * U, H = p.H, B = p.B;
  float* wl = lds;                                        // [Kp/4][ncol][4]
  float* red = lds + (long)p.Kp * ncol;                   // [kFwdWaves][ncol][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave-uniform by construction: keep it in a scalar register
  const int u0 = blockIdx.x * U;
  load_slice(p.w, p.Kp, ncol, wl, [&](int col) -> long {
    const int gate = col / U, uu = col - gate * U;
    return u0 + uu < H ? (long)gate * H + u0 + uu : -1L;
  });
  // the (unit, batch) pairs this thread owns for the whole sequence: pair index = uu * 64 + b
  int pu[2], pb[2];
  bool pok[2];
  float c_state[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int idx = tid + q * kFwdBlock;
    pu[q] = idx >> 6; pb[q] = idx & 63;
    pok[q] = idx < U * kLanes && u0 + pu[q] < H && pb[q] < B;
    c_state[q] = pok[q] ? p.c0[(long)pb[q] * H + u0 + pu[q]] : 0.f;
  }
  __syncthreads();
  const int kg_per = p.Kp / (4 * kFwdWaves);              // Kp % 32 == 0: the eight waves get equal k ranges
  const int kg0 = wave * kg_per;
  const int rot = blockIdx.x;                             // any two workgroups of an XCD start at different chunks, however the XCDs are assigned
  // all k rows of a wave in ONE request batch: a win only for short ranges (H = 400, 13 groups: 4.4 -> 3.9 us per step).  At H = 1150
  // (36 groups, 144 loads per wave) it is 3x SLOWER than the chunked loop (k loop 11.9 -> 34 us, tools/lstm_timing.py): without the
  // rotated chunk starts every workgroup of an XCD misses the same lines of the fresh slot at the same moment.  NNL_LSTM_SINGLE=2
  // forces it for A/B runs.
  const int single = ((p.single >= 1 && kg_per == 13) || (p.single == 2 && kg_per == 36)) ? kg_per : 0;
  const int ln = lane < B ? lane : 0, wcol = lane < 4 * NG ? lane : 4 * NG - 1;
  const long BH = (long)B * H, BG = (long)B * 4 * H, slot = (long)p.Kp * kLanes;
#ifdef NNL_TAPS_TIMING
  const int dbg_slot = blockIdx.x == 0 ? 0 : ((int)blockIdx.x == p.NWG / 3 ? 1 : ((int)blockIdx.x == 2 * p.NWG / 3 ? 2 : ((int)blockIdx.x == p.NWG - 1 ? 3 : -1)));
#endif
  for (int t = 0; t < p.T; ++t) {
    NNL_LSTAMP(0);
    // this step's input projections: issued before the k loop, consumed after it
    float gxv[2][4];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int g = 0; g < 4; ++g) gxv[q][g] = pok[q] ? p.gx[t * BG + (long)pb[q] * 4 * H + (long)g * H + u0 + pu[q]] : 0.f;
    f32x4 acc[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!(p.dbg & 1)) {
      if (single == 36) panel_single<NG, 4 * NG, 36>(p.xT + t * slot, wl, kg0, ln, wcol, acc);
      else if (single == 13) panel_single<NG, 4 * NG, 13>(p.xT + t * slot, wl, kg0, ln, wcol, acc);
      else if (p.pd == 6) panel<NG, 4 * NG, 6>(p.xT + t * slot, wl, kg0, kg_per, ln, wcol, rot, acc);
      else if (p.pd == 9) panel<NG, 4 * NG, 9>(p.xT + t * slot, wl, kg0, kg_per, ln, wcol, rot, acc);
      else if (p.pd == 18) panel<NG, 4 * NG, 18>(p.xT + t * slot, wl, kg0, kg_per, ln, wcol, rot, acc);
      else panel<NG, 4 * NG>(p.xT + t * slot, wl, kg0, kg_per, ln, wcol, rot, acc);
    }
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int v = 0; v < 4; ++v) red[((long)wave * ncol + 4 * g + v) * kLanes + lane] = acc[g][v];
    __syncthreads();
    NNL_LSTAMP(1);
    float* hs = p.xT + (t + 1) * slot;
    float hv[2], gact[2][4];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      float pre[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = g * U + (pok[q] ? pu[q] : 0);
        const int b = pok[q] ? pb[q] : 0;
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < kFwdWaves; w += 2)               // fixed order over the waves' partial sums
          sum += red[((long)w * ncol + col) * kLanes + b] + red[((long)(w + 1) * ncol + col) * kLanes + b];
        pre[g] = gxv[q][g] + sum;
      }
      gact[q][0] = sigmoidf_(pre[0]); gact[q][1] = sigmoidf_(pre[1]); gact[q][2] = tanhf(pre[2]); gact[q][3] = sigmoidf_(pre[3]);
      const float c = gact[q][1] * c_state[q] + gact[q][0] * gact[q][2];
      hv[q] = gact[q][3] * tanhf(c);
      c_state[q] = c;
      // the one store the other workgroups wait for goes first
      if (pok[q]) __hip_atomic_store(hs + (long)(u0 + pu[q]) * kLanes + pb[q], hv[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const bool more = t + 1 < p.T;
    NNL_LSTAMP(2);
    if (more) grid_arrive(p.arrive, t);
    NNL_LSTAMP(3);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (pok[q]) {
        const long o = t * BH + (long)pb[q] * H + u0 + pu[q];
        p.y[o] = hv[q];
        p.cy[o] = c_state[q];
        float* gt = p.gates + t * BG + (long)pb[q] * 4 * H + u0 + pu[q];
#pragma unroll
        for (int g = 0; g < 4; ++g) gt[(long)g * H] = gact[q][g];
      }
    }
    NNL_LSTAMP(4);
    if (more && !(p.dbg & 2) && !grid_wait(p.arrive, t, p.NWG, p.err, &s_flag)) return;
    NNL_LSTAMP(5);
  }
}

template <int UC>    // UC = U units (columns) per workgroup; NG = ceil(U / 4) MFMA column groups
__global__ __launch_bounds__(kBlock) void lstm_persist_bwd_kernel(PersistBwd p) {
  constexpr int NG = (UC + 3) / 4;
  extern __shared__ float lds[];
  __shared__ int s_flag;
  const int U = UC, H = p.H, B = p.B;
  float* wl = lds;                                        // [Gp/4][U][4]
  float* red = lds + (long)p.Gp * U;                      // [4 waves][4 NG][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int u0 = blockIdx.x * U;
  load_slice(p.wt, p.Gp, U, wl, [&](int col) -> long { return u0 + col < H ? (long)(u0 + col) : -1L; });
  int pu[2], pb[2];
  bool pok[2];
  float dc_state[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int idx = tid + q * kBlock;
    pu[q] = idx >> 6; pb[q] = idx & 63;
    pok[q] = idx < U * kLanes && u0 + pu[q] < H && pb[q] < B;
    dc_state[q] = (pok[q] && p.dcT) ? p.dcT[(long)pb[q] * H + u0 + pu[q]] : 0.f;
  }
  __syncthreads();
  const int kg_per = p.Gp / 16;
  const int kg0 = wave * kg_per;
  const int rot = blockIdx.x;
  const int ln = lane < B ? lane : 0, wcol = lane < UC ? lane : UC - 1;
  const long BH = (long)B * H, BG = (long)B * 4 * H, slot = (long)p.Gp * kLanes;
  for (int t = p.T - 1; t >= -1; --t) {
    // dh_t = dy_t + dgates_{t+1} W_hh  (t = T-1: + dhT;  t = -1: dh0, no cell)
    float dyv[2], cv[2], cpv[2], gv[2][4];
    if (t >= 0) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const long o = (long)pb[q] * H + u0 + pu[q];
        dyv[q] = (pok[q] && p.dy) ? p.dy[t * BH + o] : 0.f;
        cv[q] = pok[q] ? p.cy[t * BH + o] : 0.f;
        cpv[q] = pok[q] ? (t == 0 ? p.c0[o] : p.cy[(t - 1) * BH + o]) : 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) gv[q][g] = pok[q] ? p.gates[t * BG + (long)pb[q] * 4 * H + (long)g * H + u0 + pu[q]] : 0.f;
      }
    }
    f32x4 acc[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (t < p.T - 1) panel<NG, UC>(p.dgT + (long)(t + 1) * slot, wl, kg0, kg_per, ln, wcol, rot, acc);
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int v = 0; v < 4; ++v) red[((long)wave * 4 * NG + 4 * g + v) * kLanes + lane] = acc[g][v];
    __syncthreads();
    float* row = t >= 0 ? p.dgates + (long)t * B * p.Gp : nullptr;
    float* ks = t >= 0 ? p.dgT + (long)t * slot : nullptr;
    float d[2][4];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (pok[q]) {
        const int col = pu[q];
        const long o = (long)pb[q] * H + u0 + pu[q];
        float dh = (red[(0L * 4 * NG + col) * kLanes + pb[q]] + red[(1L * 4 * NG + col) * kLanes + pb[q]]) +
                   (red[(2L * 4 * NG + col) * kLanes + pb[q]] + red[(3L * 4 * NG + col) * kLanes + pb[q]]);
        if (t == p.T - 1 && p.dhT) dh += p.dhT[o];
        if (t < 0) {
          p.dh0[o] = dh;
          p.dc0[o] = dc_state[q];
        } else {
          dh += dyv[q];
          const float gi = gv[q][0], gf = gv[q][1], gg = gv[q][2], go = gv[q][3];
          const float tc = tanhf(cv[q]);
          const float dcn = dc_state[q] + dh * go * (1.f - tc * tc);
          d[q][0] = dcn * gg * (gi * (1.f - gi));
          d[q][1] = dcn * cpv[q] * (gf * (1.f - gf));
          d[q][2] = dcn * gi * (1.f - gg * gg);
          d[q][3] = dh * tc * (go * (1.f - go));
          dc_state[q] = dcn * gf;
#pragma unroll
          for (int g = 0; g < 4; ++g)             // what the other workgroups wait for goes first
            __hip_atomic_store(ks + ((long)g * H + u0 + pu[q]) * kLanes + pb[q], d[q][g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    if (t < 0) break;
    grid_arrive(p.arrive, t);                      // (t = 0 too: the dh0 pass reads slot 0)
#pragma unroll
    for (int q = 0; q < 2; ++q)
      if (pok[q])
#pragma unroll
        for (int g = 0; g < 4; ++g) row[(long)pb[q] * p.Gp + (long)g * H + u0 + pu[q]] = d[q][g];
    if (!grid_wait(p.arrive, t, p.NWG, p.err, &s_flag)) return;
  }
}

// prologue: slot 0 of the forward exchange buffer = h0, k-major (h0 == nullptr: skipped), and the pad rows k in [K, Kp) of ALL
// nslot slots = 0 (the k loop runs over Kp without predicates; W's pad columns are zero, but 0 x garbage could be NaN)
__global__ void exchange_prologue_kernel(const float* __restrict__ h0, float* __restrict__ xs, int B, int K, int Kp, int nslot) {
  const int stride = gridDim.x * blockDim.x, i0 = blockIdx.x * blockDim.x + threadIdx.x;
  if (h0)
    for (int i = i0; i < K * kLanes; i += stride) {
      const int k = i >> 6, b = i & 63;
      xs[i] = b < B ? h0[(long)b * K + k] : 0.f;
    }
  const int pad = (Kp - K) * kLanes;
  for (int i = i0; i < pad * nslot; i += stride) {
    const int sl = i / pad, r = i - sl * pad;
    xs[(long)sl * Kp * kLanes + (long)K * kLanes + r] = 0.f;
  }
}

struct Shape { int U, NWG; size_t lds_fwd, lds_bwd; bool ok; };

Shape shape_of(long B, long H, long Kp, long Gp) {
  Shape s{};
  s.U = (int)nnl_cdiv(H, 256);
  if (s.U < 1) s.U = 1;
  s.NWG = (int)nnl_cdiv(H, s.U);
  const int ngb = (s.U + 3) / 4;
  s.lds_fwd = ((size_t)Kp * 4 * s.U + (size_t)kFwdWaves * 4 * s.U * kLanes) * sizeof(float);
  s.lds_bwd = ((size_t)Gp * s.U + 4u * 4 * ngb * kLanes) * sizeof(float);
  s.ok = B >= 1 && B <= kLanes && s.U <= kMaxU && s.NWG <= 256 && s.lds_fwd <= 150 * 1024 && s.lds_bwd <= 150 * 1024 &&
         Kp % 32 == 0 && Gp % 16 == 0;
  return s;
}

template <typename K, typename P>
hipError_t coop_launch(K kernel, int nwg, size_t lds, P& p, hipStream_t s, int block = kBlock) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  void* args[] = {&p};
  return hipLaunchCooperativeKernel(reinterpret_cast<const void*>(kernel), dim3(nwg), dim3(block), args, (unsigned)lds, s);
}

}  // namespace

// ---- entry points used by lstm.hip ----------------------------------------------------------------------------------
bool nnl_lstm_persist_ok(long B, long H, long Kp, long Gp) { return shape_of(B, H, Kp, Gp).ok; }

// extra workspace of the persistent path (floats): forward [T+1][Kp][64] + T ints; backward [T][Gp][64] + T ints
size_t nnl_lstm_persist_ws_floats(long T, long Kp, long Gp) {
  const size_t f = (size_t)(T + 1) * Kp * kLanes + (size_t)T + 16;
  const size_t b = (size_t)T * Gp * kLanes + (size_t)T + 16;
  return f > b ? f : b;
}

// returns hipSuccess when the cooperative launch was issued; any other value: nothing was launched, take the per-step path
hipError_t nnl_lstm_persist_fwd(const float* gx, const float* w_hh_pad, const float* h0, const float* c0, float* y, float* cy,
                                float* gates, long T, long B, long H, long Kp, long Gp, float* ws, int* err, hipStream_t s) {
  const Shape sh = shape_of(B, H, Kp, Gp);
  if (!sh.ok) return hipErrorInvalidValue;
  PersistFwd p{};
  p.gx = gx; p.w = w_hh_pad; p.c0 = c0; p.y = y; p.cy = cy; p.gates = gates;
  p.xT = ws;
  p.arrive = reinterpret_cast<int*>(ws + (size_t)(T + 1) * Kp * kLanes);
  p.err = err;
  p.T = (int)T; p.B = (int)B; p.H = (int)H; p.Kp = (int)Kp; p.U = sh.U; p.NWG = sh.NWG;
  p.dbg = NNL_AB_INT("NNL_LSTM_DBG", 0);
  p.single = NNL_AB_INT("NNL_LSTM_SINGLE", 1);
  p.pd = NNL_AB_INT("NNL_LSTM_PD", 12);
  hipError_t e = hipMemsetAsync(p.arrive, 0, sizeof(int) * T, s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(exchange_prologue_kernel, dim3((unsigned)nnl_cdiv(H * kLanes, 256)), dim3(256), 0, s, h0, p.xT, (int)B, (int)H,
                     (int)Kp, (int)(T + 1));
  switch (sh.U) {
    case 1: return coop_launch(lstm_persist_fwd_kernel<1>, sh.NWG, sh.lds_fwd, p, s, kFwdBlock);
    case 2: return coop_launch(lstm_persist_fwd_kernel<2>, sh.NWG, sh.lds_fwd, p, s, kFwdBlock);
    case 3: return coop_launch(lstm_persist_fwd_kernel<3>, sh.NWG, sh.lds_fwd, p, s, kFwdBlock);
    case 4: return coop_launch(lstm_persist_fwd_kernel<4>, sh.NWG, sh.lds_fwd, p, s, kFwdBlock);
    case 5: return coop_launch(lstm_persist_fwd_kernel<5>, sh.NWG, sh.lds_fwd, p, s, kFwdBlock);
    case 6: return coop_launch(lstm_persist_fwd_kernel<6>, sh.NWG, sh.lds_fwd, p, s, kFwdBlock);
    case 7: return coop_launch(lstm_persist_fwd_kernel<7>, sh.NWG, sh.lds_fwd, p, s, kFwdBlock);
    default: return coop_launch(lstm_persist_fwd_kernel<8>, sh.NWG, sh.lds_fwd, p, s, kFwdBlock);
  }
}

hipError_t nnl_lstm_persist_bwd(const float* dy, const float* dhT, const float* dcT, const float* gates, const float* cy,
                                const float* c0, const float* w_hh_t_pad, float* dgates_pad, float* dh0, float* dc0, long T, long B,
                                long H, long Kp, long Gp, float* ws, int* err, hipStream_t s) {
  const Shape sh = shape_of(B, H, Kp, Gp);
  if (!sh.ok) return hipErrorInvalidValue;
  PersistBwd p{};
  p.dy = dy; p.dhT = dhT; p.dcT = dcT; p.gates = gates; p.cy = cy; p.c0 = c0; p.wt = w_hh_t_pad;
  p.dgates = dgates_pad; p.dgT = ws; p.dh0 = dh0; p.dc0 = dc0;
  p.arrive = reinterpret_cast<int*>(ws + (size_t)T * Gp * kLanes);
  p.err = err;
  p.T = (int)T; p.B = (int)B; p.H = (int)H; p.Gp = (int)Gp; p.U = sh.U; p.NWG = sh.NWG;
  hipError_t e = hipMemsetAsync(p.arrive, 0, sizeof(int) * T, s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(exchange_prologue_kernel, dim3(64), dim3(256), 0, s, (const float*)nullptr, p.dgT, (int)B, (int)(4 * H), (int)Gp,
                     (int)T);
  switch (sh.U) {
    case 1: return coop_launch(lstm_persist_bwd_kernel<1>, sh.NWG, sh.lds_bwd, p, s);
    case 2: return coop_launch(lstm_persist_bwd_kernel<2>, sh.NWG, sh.lds_bwd, p, s);
    case 3: return coop_launch(lstm_persist_bwd_kernel<3>, sh.NWG, sh.lds_bwd, p, s);
    case 4: return coop_launch(lstm_persist_bwd_kernel<4>, sh.NWG, sh.lds_bwd, p, s);
    case 5: return coop_launch(lstm_persist_bwd_kernel<5>, sh.NWG, sh.lds_bwd, p, s);
    case 6: return coop_launch(lstm_persist_bwd_kernel<6>, sh.NWG, sh.lds_bwd, p, s);
    case 7: return coop_launch(lstm_persist_bwd_kernel<7>, sh.NWG, sh.lds_bwd, p, s);
    default: return coop_launch(lstm_persist_bwd_kernel<8>, sh.NWG, sh.lds_bwd, p, s);
  }
}
