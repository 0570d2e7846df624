// igemm_wgrad2d_kernel — the weight gradient of a 3x3 / stride 1 / pad 1 convolution in the 2-D Winograd F(2x2, 3x3) domain of wino2.hip
// (round 4; the 1-D F(2,3) domain is the WINO instantiation of igemm_wgrad.h).  With
//     y[2i+p][2j+q] = sum_{xi, nu} c_p[xi] c_q[nu] M_{xi nu},   M_{xi nu} = sum_c V_{xi nu}(c) U_{xi nu}(c),   c_0 = (1,1,1,0), c_1 = (0,1,-1,1)
// the gradient of the transformed filter is a reduction over output QUADS (n, i, j):
//     dU_{xi nu}[co][ci] = sum over quads of dM_{xi nu}[co] * V_{xi nu}[ci],   dM_{xi nu} = sum_{p,q} c_p[xi] c_q[nu] dy[2i+p][2j+q]   (1, 2 or 4 pixels)
//     V_{xi nu} = (d[xi][nu] + sc d[xi][nu']) + sr (d[xi'][nu] + sc d[xi'][nu'])  as the forward (wino2.hip: row / column pairs 0-2, 1+2, 2-1, 3-1)
// i.e. 16 GEMMs [Cout x quads] x [quads x Cin] instead of 9 x [Cout x pixels] x [pixels x Cin]: a QUARTER of the rows against 16/9 of the
// columns, 2.25x fewer MFMA multiplies than the direct kernel (1.5x fewer than the 1-D domain).  Both transforms happen while the operand
// tiles are staged — up to four buffer loads and three exact +-1 adds per element each (a tile's position is uniform, so the dy pixels
// with coefficient 0 are never fetched: 2.25 loads on average) — so nothing transformed exists in HBM.  Rows are reduced in split-K
// slabs dU [splits][Cout][16][Cin]; the slab reduce that follows folds them back, dW = G^T dU G (wino2d_wgrad_finish_kernel, conv2d.hip:
// slabs in index order => bitwise reproducible).  Odd heights / widths: the missing dy pixels of the last quad row / column read as 0.
// Replaces the cuDNN weight-gradient algorithms behind the reference's 3x3 layers (retinanet.py:43-59,77-97,126-148,187-217,260-295).
#pragma once
#include "igemm_taps.h"

struct IgemmWgrad2dParams {
  const float* a;      // dy [N][H][W][Mc]   (3x3 / stride 1 / pad 1: the output has the input's height and width)
  const float* b;      // x  [N][H][W][C]
  float* y;            // dU slabs [splits][Mc][16][C]
  unsigned a_bytes, b_bytes;
  int H, W, C, H2, W2;
  int Mc, Nc, Kp;      // Nc = 16 * C, Kp = N * H2 * W2 quads
  int splits, k_per_split, grid_m, grid_n;
  int n_fast;
};

// BT x BT output tile (BT = 64: BK 32, BT = 128: BK 16), 4 waves (2 x 2) per wave group, KG wave groups per workgroup as igemm_wgrad.h
template <int BT, int BK, bool PIPE, int KG>
__global__ __launch_bounds__(256 * KG, KG > 1 ? 1 : (BT >= 128 ? 2 : 3)) void igemm_wgrad2d_kernel(const IgemmWgrad2dParams p) {
  static_assert(KG == 1 || (size_t)KG * 2 * BK * (2 * BT) >= (size_t)BT * BT, "the staging LDS must hold one accumulator tile for the group reduction");
  // staging map: a thread owns quad row(s) row0 + s * RA of the k tile and NCH = TWO 16-byte channel chunks of each (columns ca * 4 and
  // ca * 4 + BT / 2), so the quad's pixel offsets and border masks are computed once for both: the fp32 MFMA does not overlap with VALU
  // work on this part (tools/coissue_probe.hip), every address instruction of the k loop is paid in MFMA time
  constexpr int NCH = (BK * BT / 8 >= 256) ? 2 : 1;        // (64 x 16 tile: one chunk per thread)
  constexpr int CA = BT / (4 * NCH), RA = 256 / CA, PA = BK / RA, C2 = BT / 2;
  static_assert(PA >= 1 && BK % RA == 0, "tile/thread mapping");
  constexpr int WT = BT / 2, TM = WT / 32, TN = WT / 32;
  extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
  typedef float StageBuf[BK * 2 * BT];
  const int kg = KG > 1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8) : 0;
  StageBuf* lds = reinterpret_cast<StageBuf*>(lds_dyn) + kg * 2;
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles = p.grid_m * p.grid_n;
  const int lb = nnl_xcd_remap((int)blockIdx.x, (int)gridDim.x);      // a split's tiles stream the same quad range: one XCD, one L2
  const int split = lb / tiles;
  const int t_id = lb - split * tiles;
  const int tile_n = p.n_fast ? t_id % p.grid_n : t_id / p.grid_m;
  const int tile_m = p.n_fast ? t_id / p.grid_n : t_id - tile_n * p.grid_m;
  const int m0 = tile_m * BT, n0 = tile_n * BT;
  int k_begin = split * p.k_per_split;
  int k_end = min(k_begin + p.k_per_split, p.Kp);
  int nk = (k_end - k_begin + BK - 1) / BK;
  if constexpr (KG > 1) {
    const int sub = ((p.k_per_split / KG + BK - 1) / BK) * BK;
    nk = sub / BK;
    k_begin = min(k_begin + kg * sub, k_end);
    k_end = min(k_begin + sub, k_end);
  }
  const __amdgpu_buffer_rsrc_t ra_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.b), 0, (int)p.b_bytes, 0x00020000);

  // the tile's position (a column tile never straddles one: C % BT == 0) and its coefficients — all wave-uniform
  const int pos = n0 / p.C, xi = pos >> 2, nu = pos & 3;
  const float cp0 = xi < 3 ? 1.f : 0.f, cp1 = xi == 0 ? 0.f : (xi == 2 ? -1.f : 1.f);      // c_0[xi], c_1[xi]
  const float cq0 = nu < 3 ? 1.f : 0.f, cq1 = nu == 0 ? 0.f : (nu == 2 ? -1.f : 1.f);
  const float a00 = cp0 * cq0, a01 = cp0 * cq1, a10 = cp1 * cq0, a11 = cp1 * cq1;          // dM = sum a_pq dy[2i+p][2j+q]
  const int rb_ = xi < 2 ? 2 : 1, cb_ = nu < 2 ? 2 : 1;                                     // V: rows (xi, rb_), columns (nu, cb_)
  const float sr = xi == 1 ? 1.f : -1.f, sc = nu == 1 ? 1.f : -1.f;

  const int ca = tid % CA, row0 = tid / CA;
  const int a_col = m0 + ca * 4, b_col = n0 - pos * p.C + ca * 4;        // channel offsets inside a pixel
  const bool a_cok = a_col < p.Mc;
  // byte deltas of the four pixels of each operand from the quad's base pixel: dy (2i, 2j), x (2i - 1, 2j - 1)
  const int a_px = p.Mc * 4, a_row = p.W * p.Mc * 4, b_px = p.C * 4, b_row = p.W * p.C * 4;
  const int bd00 = xi * b_row + nu * b_px, bd01 = xi * b_row + cb_ * b_px, bd10 = rb_ * b_row + nu * b_px, bd11 = rb_ * b_row + cb_ * b_px;

  // per staged row: quad (n, i, j), advanced incrementally by BK quads per k tile (adds and selects only)
  const int HW2 = p.H2 * p.W2;
  const int adv_n = BK / HW2, adv_r = BK - adv_n * HW2, adv_i = adv_r / p.W2, adv_j = adv_r - adv_i * p.W2;
  const int a_adv = adv_n * p.H * a_row + adv_i * 2 * a_row + adv_j * 2 * a_px, b_adv = adv_n * p.H * b_row + adv_i * 2 * b_row + adv_j * 2 * b_px;
  const int a_jwrap = 2 * a_row - p.W2 * 2 * a_px, b_jwrap = 2 * b_row - p.W2 * 2 * b_px;    // j -= W2, i += 1
  const int a_iwrap = p.H * a_row - p.H2 * 2 * a_row, b_iwrap = p.H * b_row - p.H2 * 2 * b_row;   // i -= H2, n += 1
  // (r_i, r_j hold the quad's PIXEL row / column 2i, 2j: what the border tests below use)
  int r_k[PA], r_i[PA], r_j[PA], a_off[PA], b_off[PA];
#pragma unroll
  for (int s = 0; s < PA; ++s) {
    const int k = k_begin + row0 + s * RA;
    const int n = k / HW2, rem = k - n * HW2;
    const int qi = rem / p.W2, qj = rem - qi * p.W2;
    r_k[s] = k; r_i[s] = 2 * qi; r_j[s] = 2 * qj;
    a_off[s] = ((n * p.H + 2 * qi) * p.W + 2 * qj) * a_px + a_col * 4;
    b_off[s] = ((n * p.H + 2 * qi - 1) * p.W + 2 * qj - 1) * b_px + b_col * 4;      // may be negative: used only with a valid pixel delta
  }
  const int adv_i2 = 2 * adv_i, adv_j2 = 2 * adv_j, W22 = 2 * p.W2, H22 = 2 * p.H2;
  auto advance = [&]() {
#pragma unroll
    for (int s = 0; s < PA; ++s) {
      r_k[s] += BK;
      int j = r_j[s] + adv_j2, i = r_i[s] + adv_i2, ao = a_off[s] + a_adv, bo = b_off[s] + b_adv;
      const bool jw = j >= W22;
      j -= jw ? W22 : 0; i += jw ? 2 : 0; ao += jw ? a_jwrap : 0; bo += jw ? b_jwrap : 0;
      const bool iw = i >= H22;
      i -= iw ? H22 : 0; ao += iw ? a_iwrap : 0; bo += iw ? b_iwrap : 0;
      r_j[s] = j; r_i[s] = i; a_off[s] = ao; b_off[s] = bo;
    }
  };
  f32x4 ra[PA][NCH][4], rb[PA][NCH][4];
  auto load_tile = [&]() {
#pragma unroll
    for (int s = 0; s < PA; ++s) {
      const bool ok = r_k[s] < k_end;
      const int i2 = r_i[s], j2 = r_j[s];
      const bool okA = ok && a_cok, h1 = i2 < p.H - 1, w1 = j2 < p.W - 1;
      const unsigned ao = (unsigned)a_off[s];
      // dy pixels with a zero coefficient are not fetched (wave-uniform selects); the pixel's byte delta rides in the scalar offset
      const unsigned a0 = (a00 != 0.f && okA) ? ao : 0xFFFFFFFFu, a1 = (a01 != 0.f && okA && w1) ? ao : 0xFFFFFFFFu;
      const unsigned a2 = (a10 != 0.f && okA && h1) ? ao : 0xFFFFFFFFu;
      const unsigned a3 = (a11 != 0.f && okA && h1 && w1) ? ao : 0xFFFFFFFFu;
      const bool ra_ok = (unsigned)(i2 - 1 + xi) < (unsigned)p.H, rb_ok = (unsigned)(i2 - 1 + rb_) < (unsigned)p.H;
      const bool ca_ok = (unsigned)(j2 - 1 + nu) < (unsigned)p.W, cb_ok = (unsigned)(j2 - 1 + cb_) < (unsigned)p.W;
      const unsigned b0 = (ok && ra_ok && ca_ok) ? (unsigned)(b_off[s] + bd00) : 0xFFFFFFFFu;
      const unsigned b1 = (ok && ra_ok && cb_ok) ? (unsigned)(b_off[s] + bd01) : 0xFFFFFFFFu;
      const unsigned b2 = (ok && rb_ok && ca_ok) ? (unsigned)(b_off[s] + bd10) : 0xFFFFFFFFu;
      const unsigned b3 = (ok && rb_ok && cb_ok) ? (unsigned)(b_off[s] + bd11) : 0xFFFFFFFFu;
      // the second chunk of each pixel: the same (masked) offset + C2 floats in the scalar offset.  dy columns beyond Mc (a row tile that
      // overhangs Mc) then read the neighbouring pixel's channels or 0 past the tensor: they only reach output rows >= Mc, never stored.
#pragma unroll
      for (int h = 0; h < NCH; ++h) {
        const unsigned so = (unsigned)(h * C2 * 4);
        ra[s][h][0] = buf_load4(ra_src, a0, so); ra[s][h][1] = buf_load4(ra_src, a1, so + (unsigned)a_px);
        ra[s][h][2] = buf_load4(ra_src, a2, so + (unsigned)a_row); ra[s][h][3] = buf_load4(ra_src, a3, so + (unsigned)(a_row + a_px));
        rb[s][h][0] = buf_load4(rb_src, b0, so); rb[s][h][1] = buf_load4(rb_src, b1, so);
        rb[s][h][2] = buf_load4(rb_src, b2, so); rb[s][h][3] = buf_load4(rb_src, b3, so);
      }
    }
  };
  auto store_tile = [&](int buf) {
    float* As = lds[buf];
    float* Bs = As + BK * BT;
#pragma unroll
    for (int s = 0; s < PA; ++s) {
      // coefficients 0 / +-1: exact sums.  a00 is 0 or 1 and an unfetched pixel reads as 0, so the first term needs no multiply; the terms
      // of the pixels this position does not use are skipped by REAL wave-uniform branches (corner positions use one dy pixel, edges two,
      // centre positions four: 1.25 of the 3 terms on average; the empty asm keeps the compiler from turning them into selects)
      f32x4 va[NCH], vb[NCH];
#pragma unroll
      for (int h = 0; h < NCH; ++h) va[h] = ra[s][h][0];
      if (a01 != 0.f) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int h = 0; h < NCH; ++h)
#pragma unroll
          for (int e = 0; e < 4; ++e) va[h][e] = __builtin_fmaf(a01, ra[s][h][1][e], va[h][e]);
      }
      if (a10 != 0.f) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int h = 0; h < NCH; ++h)
#pragma unroll
          for (int e = 0; e < 4; ++e) va[h][e] = __builtin_fmaf(a10, ra[s][h][2][e], va[h][e]);
      }
      if (a11 != 0.f) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int h = 0; h < NCH; ++h)
#pragma unroll
          for (int e = 0; e < 4; ++e) va[h][e] = __builtin_fmaf(a11, ra[s][h][3][e], va[h][e]);
      }
#pragma unroll
      for (int h = 0; h < NCH; ++h) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          vb[h][e] = __builtin_fmaf(sr, __builtin_fmaf(sc, rb[s][h][3][e], rb[s][h][2][e]), __builtin_fmaf(sc, rb[s][h][1][e], rb[s][h][0][e]));
        *reinterpret_cast<f32x4*>(As + (row0 + s * RA) * BT + h * C2 + ca * 4) = va[h];
        *reinterpret_cast<f32x4*>(Bs + (row0 + s * RA) * BT + h * C2 + ca * 4) = vb[h];
      }
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
    const float* As = lds[buf] + lh * BT + wm * WT + l31;
    const float* Bs = lds[buf] + BK * BT + lh * BT + wn * WT + l31;
    if constexpr (PIPE) {
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
          for (int i = 0; i < TM; ++i) af[n][i] = As[(kk + 1) * 2 * BT + i * 32];
#pragma unroll
          for (int j = 0; j < TN; ++j) bf[n][j] = Bs[(kk + 1) * 2 * BT + j * 32];
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
      for (int i = 0; i < TM; ++i) af[i] = As[kk * 2 * BT + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = Bs[kk * 2 * BT + j * 32];
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
  if constexpr (KG > 1) {
    float* scratch = lds_dyn;                    // acc(group 0) += acc(group 1) += ... in group order (deterministic)
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
    const int col = n0 + wn * WT + j * 32 + l31;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * WT + i * 32 + (e & 3) + 8 * (e >> 2) + row_h;
        if (col < p.Nc && row < p.Mc) out[(long)row * p.Nc + col] = acc[i][j][e];
      }
    }
  }
}
