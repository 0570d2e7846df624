// igemm_wgrad_kernel — second-generation weight-gradient GEMM on the exact-fp32 MFMA:
//   dW[co][(r,s,ci)] = sum over pixels k=(n,p,q) of dy[k][co] * x[n][p*stride-pad+r][q*stride-pad+s][ci]
// Both operands are k-major (rows = pixels, channels contiguous): LDS tiles [BK][BM] / [BK][BN], ds_read_b32 fragments.
// Against the first-generation kernel (igemm_kernels.h: igemm_kmajor_kernel): operands come through BUFFER loads with
// per-lane 32-bit offsets (invalid pixels / ragged edges / rows past the split get an out-of-range offset and read 0:
// no branches), the per-row pixel decode is incremental (BK pixels per k tile: adds and selects, no division), the loads of tile t+1 are pinned ahead of the MFMAs of tile t, and __launch_bounds__ keeps 4-5 waves/SIMD.
// grid = grid_m * grid_n * splits; split s reduces pixels [s*k_per_split, (s+1)*k_per_split) into its own slab.
#pragma once
#include "igemm_taps.h"

struct IgemmWgradParams {
  unsigned long long* dbg_t;   // timing builds only (NNL_TAPS_TIMING): 5 x u64 per workgroup, see igemm_taps.h
  const float* a;      // dy [Kp][Mc]
  const float* b;      // x  [N][H][W][C]
  float* y;            // [Mc][Nc] or [splits][Mc][Nc]
  unsigned a_bytes, b_bytes;
  int H, W, C, P, Q;
  int R, S, stride, pad;
  int Mc, Nc, Kp;
  int splits, k_per_split, grid_m, grid_n;
  int n_fast;          // tile order inside a split: 1 = column tiles fastest (neighbours share the dy columns), 0 = row tiles fastest
  // WINO instantiations (3x3 / stride 1 / pad 1, even Q): the reduction index runs over output PAIRS (n, p, j) — Kp = N*P*Q/2,
  // Q below = pairs per line — and the columns over (position xi, filter row r, ci): Nc = 12*C, y = dU [Mc][4][3][C] (slabs);
  // a column tile never straddles a position (3*C % BN == 0).  See the kernel.
};

// KG > 1 (round 3): a workgroup is KG GROUPS of 4 waves (256*KG threads); group g reduces the g-th quarter / half of the split's
// pixel range into its own accumulators with its own LDS staging buffers, and the groups' accumulators are added through LDS in
// group order (deterministic) before group 0 writes ONE slab.  Same waves per CU as KG co-resident 256-thread workgroups, but KG
// times fewer split-K slabs written to and re-read from HBM (the 128x128 tile at KG = 4: 66 MB -> 17 MB per launch).
// WINO (round 3): the weight gradient of a 3x3 / stride 1 / pad 1 convolution in the Winograd F(2,3) domain of wino.hip — with
//   y(2j) = M0 + M1 + M2, y(2j+1) = M1 - M2 + M3',  M_xi = sum V_xi U_xi:   dU_xi[co][r][ci] = sum over pairs of dM_xi[co] * V_xi(r)[ci],
//   dM0 = dy(2j), dM1 = dy(2j) + dy(2j+1), dM2 = dy(2j) - dy(2j+1), dM3' = dy(2j+1);  V as in the forward (d0-d2, d1+d2, d2-d1, d3-d1).
// Both transforms happen while the operand tiles are staged (two buffer loads and one add per element each); the reduction runs
// over HALF as many rows against 12/9 as many columns: 1.5x fewer MFMA multiplies.  The slab reduce that follows folds dU back
// to dW (dg0 = dU0 + (dU1 + dU2)/2, dg1 = (dU1 - dU2)/2, dg2 = (dU1 + dU2)/2 + dU3: wino_wgrad_finish_kernel, conv2d.hip).
// PAIR (round 5): a thread stages ONE pixel row and TWO 16-byte chunks of it per operand (columns c and c + BM / 2 resp. c + BN / 2, the
// second through the load's scalar offset) instead of two rows and one chunk, so the per-step gather state (pixel index, (pp, qq) wraps,
// border tests) is advanced once for both — on gfx950 that VALU work is paid in MFMA time (tools/coissue_probe.hip).  Legal when both
// chunks of a column tile share one filter tap: 1x1 filters / linear layers, or C % BN == 0 (the launcher checks).  Columns beyond Mc / Nc
// in the second chunk read a neighbouring pixel's channels (or 0 past the tensor): they only reach output rows / columns that are not stored.
template <int BM, int BN, int BK, int WGM, int WGN, bool PIPE = false, int KG = 1, bool WINO = false, bool PAIR = false>
__global__ __launch_bounds__(256 * KG, KG > 1 ? 1 : ((BM * BN >= 128 * 128) ? 3 : 4)) void igemm_wgrad_kernel(const IgemmWgradParams p) {
  static_assert(WGM * WGN == 4, "4 waves per group");
  static_assert(KG == 1 || (size_t)KG * 2 * BK * (BM + BN) >= (size_t)BM * BN, "the staging LDS must hold one accumulator tile for the group reduction");
  static_assert(!PAIR || (!WINO && BK * BM / 8 >= 256 && BK * BN / 8 >= 256), "PAIR: two chunks per thread on both operands");
  constexpr int NCH = PAIR ? 2 : 1;
  constexpr int CA = BM / (4 * NCH), CB = BN / (4 * NCH);
  constexpr int RA = 256 / CA, RB = 256 / CB;
  constexpr int PA = BK / RA, PB = BK / RB;
  static_assert(PA >= 1 && PB >= 1 && BK % RA == 0 && BK % RB == 0, "tile/thread mapping");
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int TM = WTM / 32, TN = WTN / 32;

  // dynamic LDS (KG * 2 * BK * (BM + BN) floats: 128 KB for the 128x128 tile at KG = 4, above the 64 KB static limit)
  extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
  typedef float StageBuf[BK * (BM + BN)];
  const int kg = KG > 1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8) : 0;      // wave-uniform group index
  StageBuf* lds = reinterpret_cast<StageBuf*>(lds_dyn) + kg * 2;
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
#ifdef NNL_TAPS_TIMING
  unsigned long long* const dbg_t = (p.dbg_t != nullptr && blockIdx.x < 3276 && threadIdx.x < 256) ? p.dbg_t + (long)blockIdx.x * 5 : nullptr;
  if (dbg_t && threadIdx.x == 0) {
    dbg_t[0] = wall_clock64();
    dbg_t[4] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
  }
#endif
  const int tiles = p.grid_m * p.grid_n;
  // XCD-aware order: the hardware deals workgroups round-robin over the 8 XCDs; the remap gives each XCD a CONTIGUOUS range of
  // (split, tile) pairs, so the tiles of one split — which all stream the same pixel range of dy and x — share one L2
  const int lb = nnl_xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int split = lb / tiles;
  const int t_id = lb - split * tiles;
  const int tile_n = p.n_fast ? t_id % p.grid_n : t_id / p.grid_m;
  const int tile_m = p.n_fast ? t_id / p.grid_n : t_id - tile_n * p.grid_m;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  int k_begin = split * p.k_per_split;
  int k_end = min(k_begin + p.k_per_split, p.Kp);
  int nk = (k_end - k_begin + BK - 1) / BK;
  if constexpr (KG > 1) {
    // every group runs the SAME number of k tiles (the barriers inside the loop are workgroup-wide); a group whose range is short or
    // empty fetches zeros for the rest (the loads are predicated on k < k_end)
    const int sub = ((p.k_per_split / KG + BK - 1) / BK) * BK;
    nk = sub / BK;
    k_begin = min(k_begin + kg * sub, k_end);
    k_end = min(k_begin + sub, k_end);
  }

  const __amdgpu_buffer_rsrc_t ra_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.b), 0, (int)p.b_bytes, 0x00020000);

  const int ca = tid % CA, ra_row = tid / CA;
  const int a_col = m0 + ca * 4;
  const bool a_cok = a_col < p.Mc;
  const int cb = tid % CB, rb_row = tid / CB;
  const int b_col = n0 + cb * 4;
  const bool b_cok = b_col < p.Nc;
  const int btap = (b_cok ? b_col : 0) / p.C;
  const int bc = (b_cok ? b_col : 0) - btap * p.C;
  const int br = WINO ? btap % 3 : btap / p.S, bs = WINO ? 0 : btap - br * p.S;
  const int PQ = p.P * p.Q;
  // WINO: position xi of this column tile (tile-uniform), the two input columns (2*qq - 1 + wca / wcb) of V_xi and their sign,
  // the two coefficients of dM_xi = ac0 * dy(2*qq) + ac1 * dy(2*qq + 1)
  const int xi = WINO ? n0 / (3 * p.C) : 0;
  const int wca = xi, wcb = xi < 2 ? 2 : 1;
  const float bsgn = xi == 1 ? 1.f : -1.f;
  const float ac0 = xi == 3 ? 0.f : 1.f, ac1 = xi == 0 ? 0.f : (xi == 2 ? -1.f : 1.f);
  const int dh = br - p.pad, dw = WINO ? wca - 1 : bs - p.pad;
  const int qstep = WINO ? 2 : p.stride;                                     // input columns per unit of qq

  f32x4 ra[PA], rb[PB];
  f32x4 ra2[WINO ? PA : 1], rb2[WINO ? PB : 1];                              // WINO: the second pixel of each pair
  f32x4 rah[PAIR ? PA : 1], rbh[PAIR ? PB : 1];                              // PAIR: the second chunk of each row
  // Per-row gather state, advanced INCREMENTALLY by BK pixels per k tile (no division, multiplication or branch in the loop):
  // a pixel index k = (n, pp, qq) moves by BK = dn*P*Q + dp*Q + dq, with at most one carry out of qq and one out of pp; the
  // byte offset of its input pixel moves by a constant plus one constant per carry.  Row validity is two unsigned range
  // tests on (pp, qq) against per-thread bounds (the thread's tap (dh, dw) is fixed) and k < k_end.
  const int adv_n = BK / PQ, adv_r = BK - adv_n * PQ, adv_p = adv_r / p.Q, adv_q = adv_r - adv_p * p.Q;
  const int step_px = qstep * p.C * 4;                                       // bytes per unit of qq
  const int step_row = p.stride * p.W * p.C * 4;                             // bytes per unit of pp
  const int off_adv = adv_n * p.H * p.W * p.C * 4 + adv_p * step_row + adv_q * step_px;
  const int off_qwrap = step_row - p.Q * step_px;                            // qq -= Q, pp += 1
  const int off_pwrap = p.H * p.W * p.C * 4 - p.P * step_row;                // pp -= P, n += 1
  // pp valid  <=>  0 <= pp*stride + dh < H  <=>  pp in [pp_lo, pp_hi)
  const int pp_lo = dh < 0 ? (-dh + p.stride - 1) / p.stride : 0, qq_lo = dw < 0 ? (-dw + qstep - 1) / qstep : 0;
  const int pp_hi = min(p.P, (p.H - dh + p.stride - 1) / p.stride), qq_hi = min(p.Q, (p.W - dw + qstep - 1) / qstep);
  const unsigned pp_span = pp_hi > pp_lo ? (unsigned)(pp_hi - pp_lo) : 0u, qq_span = qq_hi > qq_lo ? (unsigned)(qq_hi - qq_lo) : 0u;
  // WINO: the second column 2*qq - 1 + wcb of the pair, delta bytes from the first
  const int dw2 = wcb - 1;
  const int q2_lo = dw2 < 0 ? 1 : 0, q2_hi = min(p.Q, (p.W - dw2 + 1) / 2);
  const unsigned q2_span = q2_hi > q2_lo ? (unsigned)(q2_hi - q2_lo) : 0u;
  const int delta_b = (wcb - wca) * p.C * 4;
  int a_k[PA]; unsigned a_off[PA];
  int b_k[PB], b_pp[PB], b_qq[PB], b_off[PB];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    a_k[i] = k_begin + ra_row + i * RA;
    a_off[i] = WINO ? (unsigned)(2 * a_k[i] * p.Mc + a_col) * 4u : (unsigned)(a_k[i] * p.Mc + a_col) * 4u;     // WINO: pixel 2k (Q even)
  }
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int k = k_begin + rb_row + i * RB;
    const int n = k / PQ, rem = k - n * PQ;
    b_k[i] = k; b_pp[i] = rem / p.Q; b_qq[i] = rem - b_pp[i] * p.Q;
    b_off[i] = (((n * p.H + b_pp[i] * p.stride + dh) * p.W + b_qq[i] * qstep + dw) * p.C + bc) * 4;
  }
  auto advance = [&]() {
#pragma unroll
    for (int i = 0; i < PA; ++i) { a_k[i] += BK; a_off[i] += (unsigned)((WINO ? 2 : 1) * BK * p.Mc) * 4u; }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      b_k[i] += BK;
      int qq = b_qq[i] + adv_q, pp = b_pp[i] + adv_p, off = b_off[i] + off_adv;
      const bool qw = qq >= p.Q;
      qq -= qw ? p.Q : 0; pp += qw ? 1 : 0; off += qw ? off_qwrap : 0;
      const bool pw = pp >= p.P;
      pp -= pw ? p.P : 0; off += pw ? off_pwrap : 0;
      b_qq[i] = qq; b_pp[i] = pp; b_off[i] = off;
    }
  };
  auto load_tile = [&]() {
    if constexpr (WINO) {
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const unsigned o = (a_cok && a_k[i] < k_end) ? a_off[i] : 0xFFFFFFFFu;
        ra[i] = buf_load4(ra_src, ac0 != 0.f ? o : 0xFFFFFFFFu, 0);                                   // (wave-uniform selects)
        ra2[i] = buf_load4(ra_src, ac1 != 0.f ? o : 0xFFFFFFFFu, (unsigned)p.Mc * 4u);
      }
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        const bool okp = b_cok && b_k[i] < k_end && (unsigned)(b_pp[i] - pp_lo) < pp_span;
        rb[i] = buf_load4(rb_src, (okp && (unsigned)(b_qq[i] - qq_lo) < qq_span) ? (unsigned)b_off[i] : 0xFFFFFFFFu, 0);
        rb2[i] = buf_load4(rb_src, (okp && (unsigned)(b_qq[i] - q2_lo) < q2_span) ? (unsigned)(b_off[i] + delta_b) : 0xFFFFFFFFu, 0);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const unsigned o = (a_cok && a_k[i] < k_end) ? a_off[i] : 0xFFFFFFFFu;
      ra[i] = buf_load4(ra_src, o, 0);
      if constexpr (PAIR) rah[i] = buf_load4(ra_src, o, (unsigned)(BM / 2 * 4));
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const bool ok = b_cok && b_k[i] < k_end && (unsigned)(b_pp[i] - pp_lo) < pp_span && (unsigned)(b_qq[i] - qq_lo) < qq_span;
      const unsigned o = ok ? (unsigned)b_off[i] : 0xFFFFFFFFu;
      rb[i] = buf_load4(rb_src, o, 0);
      if constexpr (PAIR) rbh[i] = buf_load4(rb_src, o, (unsigned)(BN / 2 * 4));
    }
  };
  auto store_tile = [&](int buf) {
    float* As = lds[buf];
    float* Bs = As + BK * BM;
    if constexpr (WINO) {
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ac0 * ra[i][e] + ac1 * ra2[i][e];
        *reinterpret_cast<f32x4*>(As + (ra_row + i * RA) * BM + ca * 4) = v;
      }
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = rb[i][e] + bsgn * rb2[i][e];
        *reinterpret_cast<f32x4*>(Bs + (rb_row + i * RB) * BN + cb * 4) = v;
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      *reinterpret_cast<f32x4*>(As + (ra_row + i * RA) * BM + ca * 4) = ra[i];
      if constexpr (PAIR) *reinterpret_cast<f32x4*>(As + (ra_row + i * RA) * BM + BM / 2 + ca * 4) = rah[i];
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      *reinterpret_cast<f32x4*>(Bs + (rb_row + i * RB) * BN + cb * 4) = rb[i];
      if constexpr (PAIR) *reinterpret_cast<f32x4*>(Bs + (rb_row + i * RB) * BN + BN / 2 + cb * 4) = rbh[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int l31 = lane & 31, lh = lane >> 5;
  auto compute = [&](int buf) {
    const float* As = lds[buf] + lh * BM + wm * WTM + l31;
    const float* Bs = lds[buf] + BK * BM + lh * BN + wn * WTN + l31;
    if constexpr (PIPE) {
      // software-pipelined: the fragment reads of k step kk+1 are issued before the MFMAs of step kk (a second, 4-register
      // fragment set), so their LDS latency hides behind TM*TN MFMAs instead of being exposed at every step
      float af[2][TM], bf[2][TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[0][i] = As[i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[0][j] = Bs[j * 32];
#pragma unroll
      for (int kk = 0; kk < BK / 2; ++kk) {
        const int c = kk & 1, n = c ^ 1;
        if (kk + 1 < BK / 2) {
#pragma unroll
          for (int i = 0; i < TM; ++i) af[n][i] = As[(kk + 1) * 2 * BM + i * 32];
#pragma unroll
          for (int j = 0; j < TN; ++j) bf[n][j] = Bs[(kk + 1) * 2 * BN + j * 32];
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][i], bf[c][j], acc[i][j], 0, 0, 0);
      }
      return;
    }
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      float af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = As[kk * 2 * BM + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = Bs[kk * 2 * BN + j * 32];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  };

  if (nk > 0) {
    load_tile();
    store_tile(0);
    __syncthreads();
    NNL_TSTAMP(1);
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) advance();                // the last iteration re-fetches the last tile (uniform; keeps one loop body)
      load_tile();
      __builtin_amdgcn_sched_barrier(0);
      compute(cur);
      __builtin_amdgcn_sched_barrier(0);
      store_tile(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
  }

  NNL_TSTAMP(2);
  if constexpr (KG > 1) {
    // acc(group 0) += acc(group 1) += ... in group order, through the staging LDS (free now: the k loop ended with a barrier)
    float* scratch = lds_dyn;
    for (int g = 1; g < KG; ++g) {
      if (kg == g) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) scratch[((i * TN + j) * 16 + e) * 256 + tid] = acc[i][j][e];
      }
      __syncthreads();
      if (kg == 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] += scratch[((i * TN + j) * 16 + e) * 256 + tid];
      }
      __syncthreads();
    }
    if (kg != 0) return;
  }
  float* out = p.y + (long)split * p.Mc * p.Nc;
  const int row_h = lh * 4;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn * WTN + j * 32 + l31;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + row_h;
        if (col < p.Nc && row < p.Mc) out[(long)row * p.Nc + col] = acc[i][j][e];
      }
    }
  }
#ifdef NNL_TAPS_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  NNL_TSTAMP(3);
#endif
}
