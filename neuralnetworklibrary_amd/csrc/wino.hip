// 3x3 / stride 1 / pad 1 convolution (forward and stride-1 dgrad) as a 1-D Winograd F(2,3) along the width, FUSED into the
// implicit-GEMM kernel — no transformed tensors in HBM, the input transform happens while the operand tile is staged.
// Replaces cuDNN's Winograd convolutions in the reference's 3x3 layers (BasicBlock / Bottleneck conv3x3, FPN / head convs:
// Applications/VisionModels/retinanet.py:43-59,77-97,126-148,187-217,260-295).
//
//   y(2j)   = M0 + M1 + M2,   y(2j+1) = M1 - M2 + M3',   M_xi = sum_{r, c} V_xi(r, c) * U_xi(r, c)
//   V0 = d0 - d2, V1 = d1 + d2, V2 = d2 - d1, V3' = d3 - d1      (d_a = in[n, h+r-1, 2j-1+a, c]: formed while STAGING — two
//   U0 = g0, U1 = (g0+g1+g2)/2, U2 = (g0-g1+g2)/2, U3 = g2        buffer loads and one add per element; U: a small pre-pass)
//
// A GEMM row is an output PAIR (n, h, j); per row 4 positions x 3 filter rows = 12 "taps" of Cin channels against 9 for ONE
// output pixel of the direct kernel: 6 MFMA k-steps per output instead of 9 (1.5x fewer), 12 A-tile loads per output instead
// of 9.  Positions 0 and 3 accumulate straight into the two output accumulators, 1 and 2 go through a third that is folded in
// (+,+ / +,-) at the position boundary.  Tile = 64 pairs x 64 channels, 4 waves (2x2), double-buffered LDS, one barrier per k
// step, buffer loads with the validity folded into the offset (igemm_taps.h).  Odd widths: the last pair of a line has one
// output.  The balanced schedule of igemm_taps_kernel (whole "main" tiles + k-sliced tail tiles, the last slice to arrive sums
// the slabs in slice order: bitwise reproducible) carries over with k = (position, filter row, channel block).
// Measured (tools/bench_wino.py, 64 images): see profiles/README.md.
#include "wino.h"
#include "wino_filter.h"
#include "igemm_taps.h"

namespace {

constexpr int kCUs = 256;

struct WinoParams {
  const float* a;      // in [N][H][W][C]
  const float* b;      // U  [Nc][4][3][C]
  float* y;            // [N][H][W][Nc]
  const float* bias;   // [Nc] or null
  const float* add;    // [N][H][W][Nc] or null
  unsigned a_bytes, b_bytes;
  int H, W, C, W2;     // W2 = ceil(W / 2) pairs per line
  int M2;              // N * H * W2 rows
  int Nc;
  int relu;
  int grid_m, grid_n;
  int bal, main_ks, n_main_tiles, tail_slices, tail_row0;      // as IgemmTapsParams; rows are PAIR rows
  float* main_out; long main_slab_stride;                      // slabs [slices][2 * rows][Nc]
  float* tail_out; long tail_slab_stride;
  int* tile_counters;
  float* bn_part; const float* bn_pivot;
  int epi4;            // 1: row-major float4 epilogue of the unsplit tiles (NNL_WINO_EPI4, default)
};

// filt [Nc][3][3][C] -> U [Nc][4][3][C]; flip: read filt[.][2-r][2-s][.] (the dgrad filter)
__global__ void wino_filter_kernel(const float* __restrict__ w, float* __restrict__ u, long KC3, int C, int flip) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // over (k, r, c)
  if (i >= KC3) return;
  const long c = i % C, kr = i / C, r = kr % 3, k = kr / 3;
  const long sr = flip ? 2 - r : r;
  const float* src = w + ((k * 3 + sr) * 3) * C + c;
  const float ga = src[0], gb = src[C], gc = src[2L * C];
  const float g0 = flip ? gc : ga, g1 = gb, g2 = flip ? ga : gc;
  float* o = u + (k * 12 + r) * C + c;
  o[0] = g0;
  o[3L * C] = 0.5f * (g0 + g1 + g2);
  o[6L * C] = 0.5f * (g0 - g1 + g2);
  o[9L * C] = g2;
}

// every filter of a model in ONE launch: block b transforms 256 (row, r, c) items — 2-D layout: (row, c) items — of descriptor block_desc[b]
__global__ void wino_filter_multi_kernel(const nnl_wino_desc_t* __restrict__ desc, const int32_t* __restrict__ block_desc) {
  const nnl_wino_desc_t d = desc[block_desc[blockIdx.x]];
  const long i = ((long)blockIdx.x - d.first_block) * 256 + threadIdx.x;
  const long C = d.ch;
  if (d.two_d) {                                     // U [rows][16][ch] of the 2-D kernel: one (row, channel) item per thread
    if (i < (long)d.rows * C) wino2_filter_item(d.src + (i / C) * 9 * C + i % C, d.dst + (i / C) * 16 * C + i % C, C, d.flip);
    return;
  }
  if (i >= (long)d.rows * 3 * C) return;
  const long c = i % C, kr = i / C, r = kr % 3, k = kr / 3;
  const long sr = d.flip ? 2 - r : r;
  const float* src = d.src + ((k * 3 + sr) * 3) * C + c;
  const float ga = src[0], gb = src[C], gc = src[2L * C];
  const float g0 = d.flip ? gc : ga, g1 = gb, g2 = d.flip ? ga : gc;
  float* o = d.dst + (k * 12 + r) * C + c;
  o[0] = g0;
  o[3L * C] = 0.5f * (g0 + g1 + g2);
  o[6L * C] = 0.5f * (g0 - g1 + g2);
  o[9L * C] = g2;
}

template <int BK>
__global__ __launch_bounds__(256, 4) void wino_kernel(const WinoParams p) {
  constexpr int BM = 64, BN = 64, BKP = BK + 4, KC = BK / 4, RPP = 256 / KC, PA = BM / RPP, PB = BN / RPP;
  __shared__ __attribute__((aligned(16))) float lds[2][(BM + BN) * BKP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  int logical, kslice = 0, nslices = 1, row0 = 0;
  bool in_tail = false;
  if (p.bal) {
    const int nmb = p.n_main_tiles * p.main_ks;
    if ((int)blockIdx.x < nmb) {
      const int u = nnl_xcd_remap(blockIdx.x, nmb);
      logical = u / p.main_ks;
      kslice = u - logical * p.main_ks;
      nslices = p.main_ks;
    } else {
      const int tb = (int)blockIdx.x - nmb;
      const int t = tb / p.tail_slices;
      kslice = tb - t * p.tail_slices;
      logical = p.n_main_tiles + t;
      nslices = p.tail_slices;
      if (nslices > 1) { row0 = p.tail_row0; in_tail = true; }
    }
  } else {
    logical = nnl_xcd_remap(blockIdx.x, gridDim.x);
  }
  const bool partial = nslices > 1;
  const int tile_m = logical / p.grid_n, tile_n = logical - tile_m * p.grid_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kc = tid % KC, lrow = tid / KC;
  const __amdgpu_buffer_rsrc_t ra_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.b), 0, (int)p.b_bytes, 0x00020000);

  // per staged row: byte offset of pixel (n, h, 2j-1) (+ this thread's 16-B chunk; for j = 0 it points before the line: masked),
  // 3 row-validity bits and 4 column-validity bits
  int a_off[PA];
  unsigned rmask[PA], cmask[PA];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int m = m0 + lrow + i * RPP;
    const bool valid = m < p.M2;
    const int mm = valid ? m : 0;
    const int line = mm / p.W2, j = mm - line * p.W2;          // line = n * H + h
    const int hh = line % p.H;
    a_off[i] = ((line * p.W + 2 * j - 1) * p.C + kc * 4) * 4;
    unsigned rm = 0, cm = 0;
    if (valid) {
      for (int r = 0; r < 3; ++r) if ((unsigned)(hh + r - 1) < (unsigned)p.H) rm |= 1u << r;
      for (int a = 0; a < 4; ++a) if ((unsigned)(2 * j - 1 + a) < (unsigned)p.W) cm |= 1u << a;
    }
    rmask[i] = rm; cmask[i] = cm;
  }
  unsigned b_off[PB];
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int nr = n0 + lrow + i * RPP;
    b_off[i] = nr < p.Nc ? (unsigned)(nr * 12 * p.C + kc * 4) * 4u : 0xFFFFFFFFu;
  }
  // per-tap state: tap t = xi * 3 + r -> the two pixel columns (ca, cb) and the sign of the second
  unsigned va[PA], vb[PA], b_tap = 0;
  float sgn = 1.f;
  auto set_tap = [&](int t) {
    const int xi = t / 3, r = t - xi * 3;
    const int ca = xi;                                          // 0, 1, 2, 3
    const int cb = xi < 2 ? 2 : 1;                              // 2, 2, 1, 1
    sgn = xi == 1 ? 1.f : -1.f;
    const int ro = (r - 1) * p.W * p.C * 4;
    b_tap = (unsigned)(t * p.C) * 4u;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const bool rok = (rmask[i] >> r) & 1u;
      va[i] = (rok && ((cmask[i] >> ca) & 1u)) ? (unsigned)(a_off[i] + ro + ca * p.C * 4) : 0xFFFFFFFFu;
      vb[i] = (rok && ((cmask[i] >> cb) & 1u)) ? (unsigned)(a_off[i] + ro + cb * p.C * 4) : 0xFFFFFFFFu;
    }
  };
  f32x4 ra[PA], ra2[PA], rb[PB];
  auto load_tile = [&](int c0) {
#pragma unroll
    for (int i = 0; i < PA; ++i) { ra[i] = buf_load4(ra_src, va[i], (unsigned)c0 * 4u); ra2[i] = buf_load4(ra_src, vb[i], (unsigned)c0 * 4u); }
#pragma unroll
    for (int i = 0; i < PB; ++i) rb[i] = buf_load4(rb_src, b_off[i], b_tap + (unsigned)c0 * 4u);
  };
  float sgn_ld = 1.f;                               // the sign that belongs to the tile sitting in ra / ra2
  auto store_tile = [&](int buf) {
    float* As = lds[buf];
    float* Bs = As + BM * BKP;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = ra[i][e] + sgn_ld * ra2[i][e];
      *reinterpret_cast<f32x4*>(As + (lrow + i * RPP) * BKP + kc * 4) = v;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) *reinterpret_cast<f32x4*>(Bs + (lrow + i * RPP) * BKP + kc * 4) = rb[i];
  };
  const int frag_off = (lane & 31) * BKP + (lane >> 5) * 4;
  auto compute = [&](int buf, f32x16& acc) {
    const float* As = lds[buf] + wm * 32 * BKP + frag_off;
    const float* Bs = lds[buf] + BM * BKP + wn * 32 * BKP + frag_off;
    f32x4 fa[BK / 8], fb[BK / 8];
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      fa[kk] = *reinterpret_cast<const f32x4*>(As + kk * 8);
      fb[kk] = *reinterpret_cast<const f32x4*>(Bs + kk * 8);
    }
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk][t], fb[kk][t], acc, 0, 0, 0);
  };

  f32x16 y0, y1, tm;
#pragma unroll
  for (int e = 0; e < 16; ++e) { y0[e] = 0.f; y1[e] = 0.f; tm[e] = 0.f; }
  const int csteps = p.C / BK, per_pos = 3 * csteps, nk_all = 12 * csteps;
  int kt0 = 0, nk = nk_all;
  if (partial) {
    const int per = (nk_all + nslices - 1) / nslices;
    kt0 = kslice * per;
    nk = min(per, nk_all - kt0);
    if (nk < 0) nk = 0;
  }
  const int kend = kt0 + nk;
  int t_nx = kt0 / csteps, c_nx = (kt0 - t_nx * csteps) * BK;
  auto advance = [&]() {
    c_nx += BK;
    if (c_nx >= p.C) { c_nx = 0; ++t_nx; set_tap(t_nx); }
  };
  if (nk > 0) {
    set_tap(t_nx);
    load_tile(c_nx);
    sgn_ld = sgn;
    store_tile(0);
  }
  __syncthreads();
  int cur = 0, kt = kt0;
  // the iterations of position xi that fall into this workgroup's k range, into `acc`; the loads of the NEXT tile (possibly of
  // the next position) go out before the MFMAs of the current one
  auto run_position = [&](int xi, f32x16& acc) {
    const int hi = min(kend, (xi + 1) * per_pos);
    for (; kt < hi; ++kt) {
      if (kt + 1 < kend) advance();
      load_tile(c_nx);
      const float s_next = sgn;
      __builtin_amdgcn_sched_barrier(0);
      compute(cur, acc);
      __builtin_amdgcn_sched_barrier(0);
      sgn_ld = s_next;
      store_tile(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
  };
  run_position(0, y0);                               // xi = 0: d0 - d2
  run_position(1, tm);                               // xi = 1: d1 + d2
#pragma unroll
  for (int e = 0; e < 16; ++e) { y0[e] += tm[e]; y1[e] += tm[e]; tm[e] = 0.f; }
  run_position(2, tm);                               // xi = 2: d2 - d1
#pragma unroll
  for (int e = 0; e < 16; ++e) { y0[e] += tm[e]; y1[e] -= tm[e]; }
  run_position(3, y1);                               // xi = 3: d3 - d1  (= -V3)

  // ---- epilogue: pair row -> pixels (line * W + 2j) and, when 2j + 1 < W, the next one ----
  const int col_l = lane & 31, row_h = (lane >> 5) * 4;
  if (partial) {
    // split tile: sc1 stores of the partial pair sums, drain, ticket; the last slice sums the slabs in slice order and finishes
    __shared__ int ticket;
    float* const base = in_tail ? p.tail_out : p.main_out;
    const long sstride = in_tail ? p.tail_slab_stride : p.main_slab_stride;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)((long)nslices * sstride * 4), 0x00020000);
    constexpr int kSc1 = 1 << 4;
    const int cl = n0 + wn * 32 + col_l;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + row_h;
      const long off = (long)kslice * sstride + (long)(row - row0) * 2 * p.Nc + cl;
      const bool ok = row < p.M2 && cl < p.Nc;
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y0[e]), rs, ok ? (int)(off * 4) : -1, 0, kSc1);
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y1[e]), rs, ok ? (int)((off + p.Nc) * 4) : -1, 0, kSc1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) ticket = __hip_atomic_fetch_add(&p.tile_counters[logical], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (ticket != nslices - 1) return;
    if (tid == 0) __hip_atomic_store(&p.tile_counters[logical], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // zero at rest
    float fs1[4] = {0.f, 0.f, 0.f, 0.f}, fs2[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k8 = 0; k8 < 8; ++k8) {                 // 128 slab rows x 16 float4 columns = 8 float4 per thread
      const int idx4 = tid + k8 * 256, srow = idx4 >> 4, c4 = n0 + (idx4 & 15) * 4;
      const int row = m0 + (srow >> 1), h = srow & 1;
      if (row >= p.M2 || c4 >= p.Nc) continue;
      const int line = row / p.W2, j = row - line * p.W2;
      if (2 * j + h >= p.W) continue;                // the missing second output of an odd line
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      for (int sl0 = 0; sl0 < nslices; sl0 += 8) {
        f32x4 part[8];
#pragma unroll
        for (int sl = 0; sl < 8; ++sl)
          if (sl0 + sl < nslices) part[sl] = buf_load4_pol(rs, (unsigned)(((long)(sl0 + sl) * sstride + ((long)(row - row0) * 2 + h) * p.Nc + c4) * 4), kSc1);
#pragma unroll
        for (int sl = 0; sl < 8; ++sl)
          if (sl0 + sl < nslices) v += part[sl];
      }
      const long o = ((long)line * p.W + 2 * j + h) * p.Nc + c4;
      if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + c4);
      if (p.add) v += *reinterpret_cast<const f32x4*>(p.add + o);
      if (p.relu == 1) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
      *reinterpret_cast<f32x4*>(p.y + o) = v;
      if (p.bn_part) {
        const f32x4 pv = *reinterpret_cast<const f32x4*>(p.bn_pivot + c4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[e] - pv[e]; fs1[e] += d; fs2[e] += d * d; }
      }
    }
    if (p.bn_part) {                                 // thread t owns columns (t & 15)*4..+3 of slab rows t>>4, +16, ...
      __syncthreads();
      float* red = &lds[0][0];                       // [16 row lanes][64 cols][2]
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        red[(((tid >> 4) * 64) + (tid & 15) * 4 + e) * 2 + 0] = fs1[e];
        red[(((tid >> 4) * 64) + (tid & 15) * 4 + e) * 2 + 1] = fs2[e];
      }
      __syncthreads();
      if (tid < 64 && n0 + tid < p.Nc) {
        float a = 0.f, b = 0.f;
        for (int r = 0; r < 16; ++r) { a += red[(r * 64 + tid) * 2]; b += red[(r * 64 + tid) * 2 + 1]; }
        p.bn_part[((long)tile_m * p.Nc + n0 + tid) * 2 + 0] = a;
        p.bn_part[((long)tile_m * p.Nc + n0 + tid) * 2 + 1] = b;
      }
    }
    return;
  }
  if (p.epi4 && p.Nc % 4 == 0) {
    // ---- row-major float4 epilogue (as igemm_taps_kernel's): the two accumulator tiles go through LDS one after the other (the k
    // loop ended with a barrier) and every thread finishes four float4 pieces of output rows per half: 16 lanes write a 256-B row
    // segment per store; addend / bias / BatchNorm pivot come in as float4 ----
    constexpr int LDT = 68;
    float* tl = &lds[0][0];                                // 64 x 68 floats (BK 16: 5120 available)
    float fs1[4] = {0.f, 0.f, 0.f, 0.f}, fs2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (h) __syncthreads();
#pragma unroll
      for (int e = 0; e < 16; ++e) tl[(wm * 32 + (e & 3) + 8 * (e >> 2) + row_h) * LDT + wn * 32 + col_l] = h ? y1[e] : y0[e];
      __syncthreads();
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) {
        const int idx4 = tid + k4 * 256, rl = idx4 >> 4, c4 = n0 + (idx4 & 15) * 4;
        const int row = m0 + rl;
        if (row >= p.M2 || c4 >= p.Nc) continue;
        const int line = row / p.W2, j = row - line * p.W2;
        if (2 * j + h >= p.W) continue;                    // the missing second output of an odd line
        f32x4 v = *reinterpret_cast<const f32x4*>(tl + rl * LDT + (idx4 & 15) * 4);
        const long o = ((long)line * p.W + 2 * j + h) * p.Nc + c4;
        if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + c4);
        if (p.add) v += *reinterpret_cast<const f32x4*>(p.add + o);
        if (p.relu == 1) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
        *reinterpret_cast<f32x4*>(p.y + o) = v;
        if (p.bn_part) {
          const f32x4 pv = *reinterpret_cast<const f32x4*>(p.bn_pivot + c4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float d = v[e] - pv[e]; fs1[e] += d; fs2[e] += d * d; }
        }
      }
    }
    if (p.bn_part) {                                       // thread t owns columns (t & 15)*4..+3 of rows t>>4, +16, +32, +48 (both halves)
      __syncthreads();
      float* red = &lds[0][0];                             // [16 row lanes][64 cols][2]
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        red[(((tid >> 4) * 64) + (tid & 15) * 4 + e) * 2 + 0] = fs1[e];
        red[(((tid >> 4) * 64) + (tid & 15) * 4 + e) * 2 + 1] = fs2[e];
      }
      __syncthreads();
      if (tid < 64 && n0 + tid < p.Nc) {
        float a = 0.f, b = 0.f;
        for (int r = 0; r < 16; ++r) { a += red[(r * 64 + tid) * 2]; b += red[(r * 64 + tid) * 2 + 1]; }
        p.bn_part[((long)tile_m * p.Nc + n0 + tid) * 2 + 0] = a;
        p.bn_part[((long)tile_m * p.Nc + n0 + tid) * 2 + 1] = b;
      }
    }
    return;
  }
  const int col = n0 + wn * 32 + col_l;
  const bool cok = col < p.Nc;
  const float bv = (p.bias != nullptr && cok) ? p.bias[col] : 0.f;
  const float piv = (p.bn_part != nullptr && cok) ? p.bn_pivot[col] : 0.f;
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int row = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + row_h;
    if (cok && row < p.M2) {
      const int line = row / p.W2, j = row - line * p.W2;
      const long o = ((long)line * p.W + 2 * j) * p.Nc + col;
      float v0 = y0[e] + bv;
      if (p.add) v0 += p.add[o];
      if (p.relu == 1) v0 = fmaxf(v0, 0.f);
      p.y[o] = v0;
      { const float d = v0 - piv; s1 += d; s2 += d * d; }
      if (2 * j + 1 < p.W) {
        float v1 = y1[e] + bv;
        if (p.add) v1 += p.add[o + p.Nc];
        if (p.relu == 1) v1 = fmaxf(v1, 0.f);
        p.y[o + p.Nc] = v1;
        const float d = v1 - piv; s1 += d; s2 += d * d;
      }
    }
  }
  if (p.bn_part != nullptr) {
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    float* red = &lds[0][0];                         // [2 row waves][64 cols][2]: the k loop ended with a barrier
    if (lane < 32) { red[((wm * 64) + wn * 32 + lane) * 2] = s1; red[((wm * 64) + wn * 32 + lane) * 2 + 1] = s2; }
    __syncthreads();
    if (tid < 64 && n0 + tid < p.Nc) {
      p.bn_part[((long)tile_m * p.Nc + n0 + tid) * 2 + 0] = red[tid * 2] + red[(64 + tid) * 2];
      p.bn_part[((long)tile_m * p.Nc + n0 + tid) * 2 + 1] = red[tid * 2 + 1] + red[(64 + tid) * 2 + 1];
    }
  }
}

// ---- schedule: the balanced plan of conv2d.hip (plan_balance_tile) for I = 12 * C / BK iterations per tile ----
struct WPlan {
  int bk, on, main_ks, n_main_tiles, tail_slices, tail_row0;
  size_t main_floats, tail_floats;
  double t_us;                           // predicted launch time (same cost model as conv2d.hip's plan_balance_tile)
};

WPlan wino_plan(long M2, int Nc, int C) {
  WPlan best{};
  const long gm = nnl_cdiv(M2, 64), gn = nnl_cdiv(Nc, 64), T = gm * gn;
  const int e_bk = NNL_AB_INT("NNL_WINO_BK", 0);
  // BK 32 where the per-tap channel loop is short (C = 64: 134.5 -> 137.5 TF/s) or the grid small (7x7 stage: 116 -> 125); BK 16
  // otherwise (28x28 / 14x14 stages: 144 / 151 against 141 / 147) — tools/bench_conv.py --ab NNL_WINO_BK=16,32
  best.bk = (e_bk == 16 || e_bk == 32) ? e_bk : ((C % 32 == 0 && (C == 64 || T < 300)) ? 32 : 16);
  if (C % best.bk != 0) best.bk = 16;
  const int bk = best.bk;
  const long I = 12L * (C / bk);
  const double c_it = bk == 32 ? 0.60 : 0.30;
  const long occ = 4;                                               // 108-127 VGPRs: four workgroups per CU
  auto wave_iters = [&](long blocks, long iters) {
    if (blocks <= 0) return 0.0;
    const long cap = occ * kCUs;
    const long full = blocks / cap, rem = blocks - full * cap;
    return (double)(full * occ + nnl_cdiv(rem, (long)kCUs)) * iters;
  };
  const double plain = wave_iters(T, I) * c_it;
  best.t_us = plain;
  if (NNL_ENV_INT("NNL_WINO_BALANCE", 1) == 0) return best;
  double best_t = plain * 0.99;
  const int plan_extra = 2;
  const double plan_bw = 16000.0e3;
  const int f_ks = NNL_ENV_INT("NNL_WINO_PLAN_KS", 0), f_S = NNL_ENV_INT("NNL_WINO_PLAN_S", 0);
  if (f_ks > 0 || f_S > 0) best_t = 1e300;
  for (int ks = 1; ks <= 4; ks *= 2) {
    if (f_ks > 0 && ks != f_ks) continue;
    if (I / ks < 8) break;
    const long units = T * ks;
    long n_main = ((units / kCUs) * kCUs / ks / gn) * gn;
    if (n_main > T) n_main = T;
    const long tail = T - n_main;
    const long it_main = nnl_cdiv(I, (long)ks);
    static const int kSlices[] = {1, 2, 3, 4, 6, 8, 9, 12, 16, 18, 24, 32, 36, 48};
    for (int S : kSlices) {
      if (tail == 0 && S > 1) break;
      if (S > 1 && I / S < 4) break;
      if (f_S > 0 && tail > 0 && S != f_S) continue;
      const long it_tail = nnl_cdiv(I, (long)S);
      double t = (wave_iters(n_main * ks, it_main) + wave_iters(tail * S, it_tail + (S > 1 ? plan_extra : 0))) * c_it;
      const long row0 = (n_main / gn) * 64 < M2 ? (n_main / gn) * 64 : M2;
      const double main_b = ks > 1 ? (2.0 * ks + 1) * row0 * 2 * Nc * 4 : 0;
      const double tail_b = S > 1 ? (2.0 * S + 1) * (M2 - row0) * 2 * Nc * 4 : 0;
      t += (main_b + tail_b) / plan_bw + (ks > 1 ? 1 : 0) + (S > 1 && tail ? 1 : 0);
      if (t < best_t) {
        best_t = t;
        best.t_us = t;
        best.on = 1; best.main_ks = ks; best.n_main_tiles = (int)n_main; best.tail_slices = tail ? S : 1;
        best.tail_row0 = (int)row0;
        best.main_floats = ks > 1 ? (size_t)ks * row0 * 2 * Nc : 0;
        best.tail_floats = (S > 1 && tail) ? (size_t)S * (M2 - row0) * 2 * Nc : 0;
      }
    }
  }
  if (best.on && best.main_ks == 1 && best.tail_slices == 1) best.on = 0;
  return best;
}

size_t align4(size_t floats) { return (floats + 3) & ~(size_t)3; }

}  // namespace

bool nnl_wino_ok(int N, int H, int W, int Cin, int Nc, int R, int S, int stride, int pad) {
  if (NNL_ENV_INT("NNL_CONV_WINO", 1) == 0) return false;
  if (R != 3 || S != 3 || stride != 1 || pad != 1 || W < 2 || Cin % 16 != 0 || Nc % 4 != 0) return false;
  const long a_b = (long)N * H * W * Cin * 4, b_b = (long)Nc * 12 * Cin * 4, y_b = (long)N * H * W * Nc * 4;
  return a_b < (1L << 31) && b_b < (1L << 31) && y_b < (1L << 31) && (long)N * H * ((W + 1) / 2) < (1L << 30);
}

// the plan's predicted time x the measured cost of a Winograd k iteration relative to the direct kernel's (BK 16: 1.18, BK 32: 1.08)
double nnl_wino_plan_time_us(int N, int H, int W, int Cin, int Nc) {
  const WPlan pl = wino_plan((long)N * H * ((W + 1) / 2), Nc, Cin);
  return pl.t_us * (pl.bk == 32 ? 1.08 : 1.18);
}

size_t nnl_wino_workspace_bytes(int N, int H, int W, int Cin, int Nc) {
  const long M2 = (long)N * H * ((W + 1) / 2);
  const WPlan pl = wino_plan(M2, Nc, Cin);
  return (align4((size_t)Nc * 12 * Cin) + (pl.on ? pl.main_floats + pl.tail_floats : 0)) * sizeof(float);
}

int nnl_wino_bn_rows(int N, int H, int W) { return (int)nnl_cdiv((long)N * H * ((W + 1) / 2), 64L); }

int nnl_wino_launch(const WinoProblem& q, void* ws, size_t ws_bytes, int* tile_counters, long n_counters, hipStream_t s) {
  const long M2 = (long)q.N * q.H * ((q.W + 1) / 2);
  const size_t u_floats = align4((size_t)q.Nc * 12 * q.Cin);
  if (ws == nullptr || ws_bytes < u_floats * sizeof(float)) return nnl_set_error(NNL_ERR_WORKSPACE, "wino: workspace too small");
  float* u = (float*)ws;
  if (q.u_pre == nullptr) {
    const long KC3 = (long)q.Nc * 3 * q.Cin;
    hipLaunchKernelGGL(wino_filter_kernel, dim3((unsigned)nnl_cdiv(KC3, 256L)), dim3(256), 0, s, q.filt, u, KC3, q.Cin, q.flip);
    NNL_CHECK_LAUNCH();
  }
  WinoParams p{};
  p.a = q.in; p.b = q.u_pre ? q.u_pre : u; p.y = q.out; p.bias = q.bias; p.add = q.add;
  p.a_bytes = (unsigned)((long)q.N * q.H * q.W * q.Cin * 4); p.b_bytes = (unsigned)((long)q.Nc * 12 * q.Cin * 4);
  p.H = q.H; p.W = q.W; p.C = q.Cin; p.W2 = (q.W + 1) / 2; p.M2 = (int)M2; p.Nc = q.Nc; p.relu = q.relu;
  p.grid_m = (int)nnl_cdiv(M2, 64L); p.grid_n = (int)nnl_cdiv(q.Nc, 64);
  p.bn_part = q.bn_part; p.bn_pivot = q.bn_pivot;
  p.epi4 = NNL_AB_INT("NNL_WINO_EPI4", 1);
  const long T = (long)p.grid_m * p.grid_n;
  WPlan pl = wino_plan(M2, q.Nc, q.Cin);
  if (pl.on && (tile_counters == nullptr || T > n_counters || ws_bytes < (u_floats + pl.main_floats + pl.tail_floats) * sizeof(float) ||
                pl.main_floats * sizeof(float) >= (1UL << 31) || pl.tail_floats * sizeof(float) >= (1UL << 31)))
    pl.on = 0;                                       // split tiles are finished in-kernel only; a slab region is one 2 GB buffer resource
  unsigned grid = (unsigned)T;
  if (pl.on) {
    p.bal = 1; p.main_ks = pl.main_ks; p.n_main_tiles = pl.n_main_tiles; p.tail_slices = pl.tail_slices; p.tail_row0 = pl.tail_row0;
    p.main_out = u + u_floats; p.main_slab_stride = (long)pl.tail_row0 * 2 * q.Nc;
    p.tail_out = p.main_out + pl.main_floats; p.tail_slab_stride = (long)(M2 - pl.tail_row0) * 2 * q.Nc;
    p.tile_counters = tile_counters;
    grid = (unsigned)(pl.n_main_tiles * pl.main_ks + (T - pl.n_main_tiles) * pl.tail_slices);
  }
  if (pl.bk == 32) hipLaunchKernelGGL(wino_kernel<32>, dim3(grid), dim3(256), 0, s, p);
  else hipLaunchKernelGGL(wino_kernel<16>, dim3(grid), dim3(256), 0, s, p);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

extern "C" int nnl_wino_filter_multi(const nnl_wino_desc_t* desc, const int32_t* block_desc, int64_t n_blocks, void* stream) {
  NNL_CHECK_ARG(desc && block_desc && n_blocks > 0 && n_blocks < (1LL << 31), "wino_filter_multi: bad arguments");
  hipLaunchKernelGGL(wino_filter_multi_kernel, dim3((unsigned)n_blocks), dim3(256), 0, (hipStream_t)stream, desc, block_desc);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

// debug / A-B entry (tools/bench_wino.py): ws as nnl_wino_workspace_bytes; counters: >= tiles zeroed int32 or null
extern "C" size_t nnl_debug_conv_wino_workspace_bytes(int N, int H, int W, int C, int K) { return nnl_wino_workspace_bytes(N, H, W, C, K); }
extern "C" int nnl_debug_conv_wino_fwd(const float* x, const float* w, const float* bias, const float* add, float* y, void* ws,
                                       size_t ws_bytes, int32_t* counters, long n_counters, float* bn_part, const float* bn_pivot, int N,
                                       int H, int W, int C, int K, int relu, int flip, void* stream) {
  NNL_CHECK_ARG(nnl_wino_ok(N, H, W, C, K, 3, 3, 1, 1), "wino: unsupported shape");
  WinoProblem q{};
  q.in = x; q.filt = w; q.out = y; q.bias = bias; q.add = add; q.N = N; q.H = H; q.W = W; q.Cin = C; q.Nc = K; q.relu = relu; q.flip = flip;
  q.bn_part = bn_part; q.bn_pivot = bn_pivot;
  return nnl_wino_launch(q, ws, ws_bytes, counters, n_counters, (hipStream_t)stream);
}
