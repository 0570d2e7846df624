// Kernel templates for igemm.h (device code only; included by conv2d.hip and gemm.hip).
#pragma once
#include "igemm.h"

enum { IGEMM_MODE_FWD = 0, IGEMM_MODE_DGRAD = 1 };

// ---------------------------------------------------------------------------------------------------------
// igemm_rowk: 256 threads = 4 waves arranged WGM x WGN; block tile BM x BN; k-step BK (multiple of 8).
// LDS: two buffers of [BM+BN][BK+4] floats (row pad of one 16-B slot makes the ds_read_b128 fragment reads
// conflict-free: 16-lane groups hit 16 distinct slots because the row stride is 5 slots, coprime with 16).
// Pipeline: global->register prefetch of tile t+1 is issued before the MFMAs of tile t and written to the
// other LDS buffer after them; one barrier per k-step.
// ---------------------------------------------------------------------------------------------------------
template <int BM, int BN, int BK, int WGM, int WGN, int MODE>
__global__ __launch_bounds__(256) void igemm_rowk_kernel(const IgemmRowkParams p) {
  static_assert(WGM * WGN == 4, "4 waves per block");
  static_assert(BK % 8 == 0, "BK multiple of 8");
  constexpr int BKP = BK + 4;
  constexpr int KC = BK / 4;                 // float4 chunks per tile row
  constexpr int RPP = 256 / KC;              // tile rows filled per pass of the 256 threads
  constexpr int PA = BM / RPP, PB = BN / RPP;
  static_assert(PA >= 1 && PB >= 1 && BM % RPP == 0 && BN % RPP == 0, "tile/thread mapping");
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  static_assert(TM >= 1 && TN >= 1, "wave tile");

  __shared__ __attribute__((aligned(16))) float lds[2][(BM + BN) * BKP];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int logical = nnl_xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = logical / p.grid_n, tile_n = logical - tile_m * p.grid_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int kc = tid % KC, lrow = tid / KC;

  // ---- per-thread row bookkeeping (rows are fixed for the whole k loop) ----
  long a_img[PA];
  int a_h0[PA], a_w0[PA];
  const int PQ = p.P * p.Q;
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int m = m0 + lrow + i * RPP;
    const bool valid = m < p.M;
    const int mm = valid ? m : 0;
    const int n = mm / PQ;
    const int rem = mm - n * PQ;
    const int pp = rem / p.Q;
    const int qq = rem - pp * p.Q;
    a_img[i] = (long)n * p.H * p.W * p.C;
    if (MODE == IGEMM_MODE_FWD) {
      a_h0[i] = pp * p.stride - p.pad;
      a_w0[i] = qq * p.stride - p.pad;
    } else {
      a_h0[i] = pp + p.pad;
      a_w0[i] = qq + p.pad;
    }
    if (!valid) a_h0[i] = -(1 << 28);        // every tap fails the range test
  }
  const float* b_row[PB];
  bool b_ok[PB];
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int nr = n0 + lrow + i * RPP;
    b_ok[i] = nr < p.Nc;
    b_row[i] = p.b + (long)(b_ok[i] ? nr : 0) * p.Kg;
  }

  f32x4 ra[PA], rb[PB];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  auto load_tile = [&](int kt) {
    const int k4 = kt * BK + kc * 4;
    const bool kvalid = k4 < p.Kg;
    const int tap = k4 / p.C;
    const int c = k4 - tap * p.C;
    const int r = tap / p.S;
    const int s = tap - r * p.S;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      int h, w;
      bool ok = kvalid;
      if (MODE == IGEMM_MODE_FWD) {
        h = a_h0[i] + r;
        w = a_w0[i] + s;
      } else {
        const int hh = a_h0[i] - r, ww = a_w0[i] - s;
        if (p.stride == 1) {
          h = hh; w = ww;
        } else {
          ok = ok && hh >= 0 && ww >= 0 && (hh % p.stride == 0) && (ww % p.stride == 0);
          h = hh / p.stride; w = ww / p.stride;
        }
      }
      ok = ok && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
      ra[i] = ok ? *reinterpret_cast<const f32x4*>(p.a + a_img[i] + ((long)h * p.W + w) * p.C + c) : zero4;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i)
      rb[i] = (b_ok[i] && kvalid) ? *reinterpret_cast<const f32x4*>(b_row[i] + k4) : zero4;
  };

  auto store_tile = [&](int buf) {
    float* As = lds[buf];
    float* Bs = As + BM * BKP;
#pragma unroll
    for (int i = 0; i < PA; ++i) *reinterpret_cast<f32x4*>(As + (lrow + i * RPP) * BKP + kc * 4) = ra[i];
#pragma unroll
    for (int i = 0; i < PB; ++i) *reinterpret_cast<f32x4*>(Bs + (lrow + i * RPP) * BKP + kc * 4) = rb[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int frag_off = (lane & 31) * BKP + (lane >> 5) * 4;
  auto compute = [&](int buf) {
    const float* As = lds[buf] + wm * WTM * BKP + frag_off;
    const float* Bs = lds[buf] + BM * BKP + wn * WTN * BKP + frag_off;
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      f32x4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(As + i * 32 * BKP + kk * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(Bs + j * 32 * BKP + kk * 8);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][t], bf[j][t], acc[i][j], 0, 0, 0);
    }
  };

  const int nk = (p.Kg + BK - 1) / BK;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) load_tile(kt + 1);
    compute(cur);
    if (more) store_tile(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue: bias + ReLU, NHWC store (each store instruction writes two 128-B row segments) ----
  const int col_l = lane & 31, row_h = (lane >> 5) * 4;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn * WTN + j * 32 + col_l;
    const bool cok = col < p.Nc;
    const float bv = (p.bias != nullptr && cok) ? p.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + row_h;
        if (cok && row < p.M) {
          float v = acc[i][j][e] + bv;
          if (p.add) v += p.add[(long)row * p.Nc + col];
          if (p.relu) v = fmaxf(v, 0.f);
          p.y[(long)row * p.Nc + col] = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// igemm_kmajor (wgrad / dW): C[Mc][Nc] = sum over pixels k of dy[k][Mc] * xgather[k][Nc].
// LDS tiles are [BK][BM] and [BK][BN] (k-major, unpadded): fragment reads are ds_read_b32 of 32 consecutive
// floats per lane half (conflict-free), tile fills are ds_write_b128 of whole rows.
// grid = grid_m * grid_n * splits; split s reduces pixels [s*k_per_split, (s+1)*k_per_split).
// ---------------------------------------------------------------------------------------------------------
template <int BM, int BN, int BK, int WGM, int WGN>
__global__ __launch_bounds__(256) void igemm_kmajor_kernel(const IgemmKmajorParams p) {
  static_assert(WGM * WGN == 4, "4 waves per block");
  constexpr int CA = BM / 4, CB = BN / 4;              // float4 chunks per k-row
  constexpr int RA = 256 / CA, RB = 256 / CB;          // k-rows per pass
  constexpr int PA = BK / RA, PB = BK / RB;
  static_assert(PA >= 1 && PB >= 1 && BK % RA == 0 && BK % RB == 0, "tile/thread mapping");
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  static_assert(TM >= 1 && TN >= 1, "wave tile");

  __shared__ __attribute__((aligned(16))) float lds[2][BK * (BM + BN)];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int tiles = p.grid_m * p.grid_n;
  const int split = blockIdx.x / tiles;
  const int t_id = blockIdx.x - split * tiles;
  const int tile_n = t_id / p.grid_m, tile_m = t_id - tile_n * p.grid_m;   // m fastest: blocks share the x gather
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const long k_begin = (long)split * p.k_per_split;
  long k_end = k_begin + p.k_per_split;
  if (k_end > p.Kp) k_end = p.Kp;

  // A (dy) mapping
  const int ca = tid % CA, ra_row = tid / CA;
  const int a_col = m0 + ca * 4;
  const bool a_cok = a_col < p.Mc;                      // Mc % 4 == 0 is required by the host wrapper
  // B (x gather) mapping: this thread's 4 columns = one tap (r,s) and channels c..c+3, fixed for the k loop
  const int cb = tid % CB, rb_row = tid / CB;
  const int b_col = n0 + cb * 4;
  const bool b_cok = b_col < p.Nc;
  const int btap = (b_cok ? b_col : 0) / p.C;
  const int bc = (b_cok ? b_col : 0) - btap * p.C;
  const int br = btap / p.S, bs = btap - br * p.S;
  const int PQ = p.P * p.Q;

  f32x4 ra[PA], rb[PB];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  auto load_tile = [&](long k0) {
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const long k = k0 + ra_row + i * RA;
      ra[i] = (a_cok && k < k_end) ? *reinterpret_cast<const f32x4*>(p.a + k * p.Mc + a_col) : zero4;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const long k = k0 + rb_row + i * RB;
      bool ok = b_cok && k < k_end;
      const int kk = ok ? (int)k : 0;
      const int n = kk / PQ;
      const int rem = kk - n * PQ;
      const int pp = rem / p.Q;
      const int qq = rem - pp * p.Q;
      const int h = pp * p.stride - p.pad + br;
      const int w = qq * p.stride - p.pad + bs;
      ok = ok && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
      rb[i] = ok ? *reinterpret_cast<const f32x4*>(p.b + (((long)n * p.H + h) * p.W + w) * p.C + bc) : zero4;
    }
  };
  auto store_tile = [&](int buf) {
    float* As = lds[buf];
    float* Bs = As + BK * BM;
#pragma unroll
    for (int i = 0; i < PA; ++i) *reinterpret_cast<f32x4*>(As + (ra_row + i * RA) * BM + ca * 4) = ra[i];
#pragma unroll
    for (int i = 0; i < PB; ++i) *reinterpret_cast<f32x4*>(Bs + (rb_row + i * RB) * BN + cb * 4) = rb[i];
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

  const int nk = (int)((k_end - k_begin + BK - 1) / BK);
  if (nk > 0) {
    load_tile(k_begin);
    store_tile(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
      const bool more = kt + 1 < nk;
      if (more) load_tile(k_begin + (long)(kt + 1) * BK);
      compute(cur);
      if (more) store_tile(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
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
}
