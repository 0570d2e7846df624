// 3x3 / stride 1 / pad 1 convolution (forward and stride-1 dgrad) as a 2-D Winograd F(2x2, 3x3), FUSED into the implicit-GEMM
// kernel like wino.hip's 1-D F(2,3): no transformed tensors in HBM, the input transform B^T d B happens while the operand tile
// is staged (four buffer loads and three FMAs per element), the output transform A^T M A in the accumulators.
//
//   Y[p][q] = sum_{xi, nu} c_p[xi] c_q[nu] M_{xi nu},   M_{xi nu} = sum_c V_{xi nu}(c) U_{xi nu}(c),   c_0 = (1,1,1,0), c_1 = (0,1,-1,1)
//   V_{xi nu} = sum over the row pair of xi and the column pair of nu, pairs (a, b, sign): xi 0: d0 - d2, 1: d1 + d2, 2: d2 - d1,
//   3: d3 - d1 (as wino.hip: the last one is -V3 of the textbook form, c_1[3] carries the sign);  U = G g G^T (a small pre-pass)
//
// A GEMM row is a 2x2 output QUAD (n, i, j); per row 16 positions of Cin channels against 36 for the direct kernel: 4 MFMA k
// steps per output instead of 9 (2.25x fewer; 1-D: 6).  All positions accumulate into one tile that is folded into the four
// output tiles with the position's coefficients when the position's channel loop ends.  Tile = 64 quads x 64 channels, 4 waves.
// Replaces cuDNN's Winograd convolutions in the reference's 3x3 layers (retinanet.py:43-59,77-97,126-148,187-217,260-295), where the
// dispatcher (conv2d.hip: wino_mode) predicts it >= 10 % under the 1-D kernel.  Schedule: the balanced plan of the other conv kernels
// with its own fitted cost model (w2_cost below).  Measured: profiles/README.md, r3_wino2d_*.
//
// SMALL GRIDS (round 5; the strong-scaling regime, 8 - 32 images per GPU): a 14^2 / 7^2 stage at 8 images is 28 / 16 tiles for 256 CUs, and
// the k-sliced plan above makes every slice carry FOUR output-pixel slabs and the last arriver of a tile read them all back (1 MB for 16
// slices).  The POSITION-SPLIT instantiation (POS = true) gives every workgroup ONE position (xi, nu) of one tile — or a channel slice
// of one — so its accumulator IS M_{xi nu}: no fold, no y tiles (16 accumulator registers instead of 80), one slab per slice; the tile's
// last arriver sums the channel slices of each position and applies A^T M A (coefficients 0 / +-1) from the 16 M tiles.  The natural
// 16-way (x channel slices) split fills the chip where the direct kernel runs one or two workgroups per CU at ~47 TF/s.
#include "wino.h"
#include "wino_filter.h"
#include "igemm_taps.h"
#include <algorithm>

namespace {

constexpr int kCUs = 256;

// sc1 (device-scope) 16-B load with the wave-uniform part of the address in the scalar offset operand (no VGPR per distinct slab)
template <int POL>
__device__ __forceinline__ f32x4 buf_load4_spol(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
  const i32x4 v = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)soff, POL));
  return __builtin_bit_cast(f32x4, v);
}

struct Wino2Params {
  const float* a;      // in [N][H][W][C]
  const float* b;      // U  [Nc][16][C]
  float* y;            // [N][H][W][Nc]
  const float* bias;
  const float* add;
  unsigned a_bytes, b_bytes;
  int H, W, C, H2, W2; // H2 x W2 quads per image
  int ch;              // channel chunk of the k order (a multiple of BK dividing C)
  int prio;            // 1: raise the wave priority around the MFMA block (NNL_WINO2_PRIO)
  int fold_skip;       // 1 (default): a position is folded only into the output tiles it feeds (NNL_WINO2_FOLD_SKIP=0: all four, A/B)
  int M4;              // N * H2 * W2 rows
  unsigned mg_W2, mg_H2;   // ceil(2^32 / d) (0: d = 1): quad row -> (n, i, j) by multiply-high instead of division (the epilogue did 32 runtime
                       // divisions per thread and tile: most of the 12 us fixed cost per workgroup generation the planner had fitted)
  int Nc;
  int relu;
  int grid_m, grid_n;
  int bal, main_ks, n_main_tiles, tail_slices, tail_row0;      // as IgemmTapsParams; rows are QUAD rows
  float* main_out; long main_slab_stride;                      // slabs [slices][4 * rows][Nc]
  float* tail_out; long tail_slab_stride;
  int* tile_counters;
  float* bn_part; const float* bn_pivot;
  int pos_cs;          // POS instantiation: channel slices per position (slices per tile = 16 * pos_cs; C / BK divisible by pos_cs)
};

__global__ void wino2_filter_kernel(const float* __restrict__ w, float* __restrict__ u, long KC, int C, int flip) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // over (k, c)
  if (i >= KC) return;
  const long c = i % C, k = i / C;
  wino2_filter_item(w + k * 9 * C + c, u + k * 16 * C + c, C, flip);
}

template <int BK, int OCC, bool POS = false>
__global__ __launch_bounds__(256, OCC) void wino2_kernel(const Wino2Params p) {
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
  auto qdiv = [](int n, unsigned mg) { return mg ? (int)__umulhi((unsigned)n, mg) : n; };          // n / d, mg = ceil(2^32 / d); 0: d = 1
  const __amdgpu_buffer_rsrc_t ra_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.b), 0, (int)p.b_bytes, 0x00020000);

  // per staged row: byte offset of pixel (n, 2i-1, 2j-1) (+ this thread's 16-B chunk; may point before the image: masked),
  // 4 row-validity bits and 4 column-validity bits
  int a_off[PA];
  unsigned rmask[PA], cmask[PA];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int m = m0 + lrow + i * RPP;
    const bool valid = m < p.M4;
    const int mm = valid ? m : 0;
    const int l2 = qdiv(mm, p.mg_W2), j = mm - l2 * p.W2;      // l2 = n * H2 + i
    const int n = qdiv(l2, p.mg_H2), ii = l2 - n * p.H2;
    a_off[i] = (((n * p.H + 2 * ii - 1) * p.W + 2 * j - 1) * p.C + kc * 4) * 4;
    unsigned rm = 0, cm = 0;
    if (valid) {
      for (int a = 0; a < 4; ++a) if ((unsigned)(2 * ii - 1 + a) < (unsigned)p.H) rm |= 1u << a;
      for (int a = 0; a < 4; ++a) if ((unsigned)(2 * j - 1 + a) < (unsigned)p.W) cm |= 1u << a;
    }
    rmask[i] = rm; cmask[i] = cm;
  }
  unsigned b_off[PB];
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int nr = n0 + lrow + i * RPP;
    b_off[i] = nr < p.Nc ? (unsigned)(nr * 16 * p.C + kc * 4) * 4u : 0xFFFFFFFFu;
  }
  // per-position state: position t = xi * 4 + nu -> the four pixels (row pair) x (column pair) and the two signs
  unsigned v00[PA], v01[PA], v10[PA], v11[PA], b_tap = 0;
  float sr = 1.f, sc = 1.f;
  auto set_pos = [&](int t) {
    const int xi = t >> 2, nu = t & 3;
    const int ra_ = xi, rb_ = xi < 2 ? 2 : 1;
    const int ca = nu, cb = nu < 2 ? 2 : 1;
    sr = xi == 1 ? 1.f : -1.f;
    sc = nu == 1 ? 1.f : -1.f;
    b_tap = (unsigned)(t * p.C) * 4u;
    const int wc = p.W * p.C * 4, c4 = p.C * 4;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const bool ra_ok = (rmask[i] >> ra_) & 1u, rb_ok = (rmask[i] >> rb_) & 1u;
      const bool ca_ok = (cmask[i] >> ca) & 1u, cb_ok = (cmask[i] >> cb) & 1u;
      v00[i] = (ra_ok && ca_ok) ? (unsigned)(a_off[i] + ra_ * wc + ca * c4) : 0xFFFFFFFFu;
      v01[i] = (ra_ok && cb_ok) ? (unsigned)(a_off[i] + ra_ * wc + cb * c4) : 0xFFFFFFFFu;
      v10[i] = (rb_ok && ca_ok) ? (unsigned)(a_off[i] + rb_ * wc + ca * c4) : 0xFFFFFFFFu;
      v11[i] = (rb_ok && cb_ok) ? (unsigned)(a_off[i] + rb_ * wc + cb * c4) : 0xFFFFFFFFu;
    }
  };
  f32x4 r00[PA], r01[PA], r10[PA], r11[PA], rb[PB];
  auto load_tile = [&](int c0) {
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      r00[i] = buf_load4(ra_src, v00[i], (unsigned)c0 * 4u); r01[i] = buf_load4(ra_src, v01[i], (unsigned)c0 * 4u);
      r10[i] = buf_load4(ra_src, v10[i], (unsigned)c0 * 4u); r11[i] = buf_load4(ra_src, v11[i], (unsigned)c0 * 4u);
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) rb[i] = buf_load4(rb_src, b_off[i], b_tap + (unsigned)c0 * 4u);
  };
  float sr_ld = 1.f, sc_ld = 1.f;                    // the signs that belong to the tile sitting in the staging registers
  auto store_tile = [&](int buf) {
    float* As = lds[buf];
    float* Bs = As + BM * BKP;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e)                   // signs are +-1: the FMAs are exact sums
        v[e] = __builtin_fmaf(sr_ld, __builtin_fmaf(sc_ld, r11[i][e], r10[i][e]), __builtin_fmaf(sc_ld, r01[i][e], r00[i][e]));
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

  f32x16 y00, y01, y10, y11, tm;
#pragma unroll
  for (int e = 0; e < 16; ++e) { tm[e] = 0.f; }
  if constexpr (!POS) {
#pragma unroll
    for (int e = 0; e < 16; ++e) { y00[e] = 0.f; y01[e] = 0.f; y10[e] = 0.f; y11[e] = 0.f; }
  }
  // k order: channel CHUNK (p.ch channels) outermost, the 16 positions, the chunk's BK blocks innermost.  Default chunk = all of C
  // (position-major: one fold per position).  Smaller chunks bring the four uses of a patch pixel closer together (the position-
  // major order streams the input 16 times from beyond the L2: profiles/r3_traffic.json) at the price of a fold every chunk —
  // measured SLOWER (NNL_WINO2_CHUNK=64: 28^2 stage -5 %, 14^2 stage -7 %, profiles/r3_wino2d_chunk_ab_bs64.log): the re-reads hit the
  // Infinity Cache and are not what bounds the loop.
  const int csteps = p.C / BK, nk_all = 16 * csteps, cpc = p.ch / BK, per_chunk = 16 * cpc;
  if constexpr (POS) {
    // ONE position, a channel slice of it: offsets and signs are fixed for the whole loop, only the channel offset advances.  (A second
    // staging register set — two tiles in flight — measured no better at 8 images and 10 % worse at 32, where it costs the fourth
    // co-resident workgroup: profiles/r5_wino2_pos_*.log.)
    const int t_pos = kslice / p.pos_cs, sub = kslice - t_pos * p.pos_cs, nkb = csteps / p.pos_cs;
    set_pos(t_pos);
    sr_ld = sr; sc_ld = sc;
    const int cb0 = sub * nkb * BK;
    load_tile(cb0);
    store_tile(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nkb; ++kt) {
      if (kt + 1 < nkb) load_tile(cb0 + (kt + 1) * BK);
      __builtin_amdgcn_sched_barrier(0);
      compute(cur, tm);
      __builtin_amdgcn_sched_barrier(0);
      if (kt + 1 < nkb) store_tile(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
  }
  int kt0 = 0, nk = nk_all;
  if (POS) nk = 0;
  else if (partial) {
    const int per = (nk_all + nslices - 1) / nslices;
    kt0 = kslice * per;
    nk = min(per, nk_all - kt0);
    if (nk < 0) nk = 0;
  }
  const int kend = kt0 + nk;
  int chunk_nx = kt0 / per_chunk, t_nx = (kt0 - chunk_nx * per_chunk) / cpc, cs_nx = kt0 - chunk_nx * per_chunk - t_nx * cpc;
  int c_nx = chunk_nx * p.ch + cs_nx * BK;
  int t_cur = t_nx, cs_cur = cs_nx;
  auto advance = [&]() {
    if (++cs_nx == cpc) {
      cs_nx = 0;
      if (++t_nx == 16) { t_nx = 0; ++chunk_nx; }
      set_pos(t_nx);
    }
    c_nx = chunk_nx * p.ch + cs_nx * BK;
  };
  if (nk > 0) {
    set_pos(t_nx);
    load_tile(c_nx);
    sr_ld = sr; sc_ld = sc;
    store_tile(0);
  }
  __syncthreads();
  int cur = 0;
  for (int kt = kt0; kt < kend; ++kt) {
    if (kt + 1 < kend) advance();
    load_tile(c_nx);
    const float sr_n = sr, sc_n = sc;
    __builtin_amdgcn_sched_barrier(0);
    if (p.prio) __builtin_amdgcn_s_setprio(2);         // (A/B: NNL_WINO2_PRIO — the MFMA block of this wave ahead of the other waves' staging)
    compute(cur, tm);
    if (p.prio) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    sr_ld = sr_n; sc_ld = sc_n;
    store_tile(cur ^ 1);
    __syncthreads();
    cur ^= 1;
    if (++cs_cur == cpc || kt + 1 == kend) {          // the position's blocks of this chunk (or this slice of them) are done: fold tm in
      const int xi = t_cur >> 2, nu = t_cur & 3;
      const float cp0 = xi < 3 ? 1.f : 0.f, cp1 = xi == 0 ? 0.f : (xi == 2 ? -1.f : 1.f);
      const float cq0 = nu < 3 ? 1.f : 0.f, cq1 = nu == 0 ? 0.f : (nu == 2 ? -1.f : 1.f);
      const float k00 = cp0 * cq0, k01 = cp0 * cq1, k10 = cp1 * cq0, k11 = cp1 * cq1;
      // coefficients 0 / +-1, wave-uniform: corner positions feed one output tile, edge positions two, centre positions four
      if (k00 != 0.f || !p.fold_skip) {
#pragma unroll
        for (int e = 0; e < 16; ++e) y00[e] = __builtin_fmaf(k00, tm[e], y00[e]);
      }
      if (k01 != 0.f || !p.fold_skip) {
#pragma unroll
        for (int e = 0; e < 16; ++e) y01[e] = __builtin_fmaf(k01, tm[e], y01[e]);
      }
      if (k10 != 0.f || !p.fold_skip) {
#pragma unroll
        for (int e = 0; e < 16; ++e) y10[e] = __builtin_fmaf(k10, tm[e], y10[e]);
      }
      if (k11 != 0.f || !p.fold_skip) {
#pragma unroll
        for (int e = 0; e < 16; ++e) y11[e] = __builtin_fmaf(k11, tm[e], y11[e]);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) tm[e] = 0.f;
      cs_cur = 0; t_cur = (t_cur + 1) & 15;
    }
  }

  // ---- epilogue: quad row -> pixels (2i + p, 2j + q) ----
  const int col_l = lane & 31, row_h = (lane >> 5) * 4;
  constexpr int kSc1 = 1 << 4;
  if constexpr (POS) {
    // slab [slice = t * cs + s][quad row][Nc] of M_t partial sums (sc1 stores), drain, ticket; the tile's last slice sums the channel
    // slices of every position in slice order and applies the output transform — bitwise reproducible whoever is last
    __shared__ int ticket;
    const long sstride = p.main_slab_stride;                 // M4 * Nc
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.main_out, 0, (int)((long)nslices * sstride * 4), 0x00020000);
    const int cl = n0 + wn * 32 + col_l;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + row_h;
      const long off = (long)kslice * sstride + (long)row * p.Nc + cl;
      const bool ok = row < p.M4 && cl < p.Nc;
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(tm[e]), rs, ok ? (int)(off * 4) : -1, 0, kSc1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) ticket = __hip_atomic_fetch_add(&p.tile_counters[logical], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (ticket != nslices - 1) return;
    if (tid == 0) __hip_atomic_store(&p.tile_counters[logical], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // zero at rest
    float fs1[4] = {0.f, 0.f, 0.f, 0.f}, fs2[4] = {0.f, 0.f, 0.f, 0.f};
    const int c4 = n0 + (tid & 15) * 4, cs = p.pos_cs;
    const int wrow = p.W * p.Nc;
    // this thread's four (quad row, 4-column) items, one at a time with its 16 slab loads in flight (two at a time needs 168 VGPRs: slower)
    auto item_off = [&](int k4, bool& ok) {
      const int row = m0 + (tid >> 4) + 16 * k4;
      ok = row < p.M4 && c4 < p.Nc;
      return ok ? (unsigned)(((long)row * p.Nc + c4) * 4) : 0xFFFFFFFFu;
    };
    auto finish_item = [&](int k4, f32x4 (&m)[16]) {
      const int row = m0 + (tid >> 4) + 16 * k4;
      // A^T M A: over nu first (r0 = m0 + m1 + m2, r1 = m1 - m2 + m3), then the same pattern over xi
      f32x4 r0[4], r1[4];
#pragma unroll
      for (int xi = 0; xi < 4; ++xi) {
        r0[xi] = (m[xi * 4 + 0] + m[xi * 4 + 1]) + m[xi * 4 + 2];
        r1[xi] = (m[xi * 4 + 1] - m[xi * 4 + 2]) + m[xi * 4 + 3];
      }
      f32x4 out[4];
      out[0] = (r0[0] + r0[1]) + r0[2];                        // (p, q) = (0, 0)
      out[1] = (r1[0] + r1[1]) + r1[2];                        // (0, 1)
      out[2] = (r0[1] - r0[2]) + r0[3];                        // (1, 0)
      out[3] = (r1[1] - r1[2]) + r1[3];                        // (1, 1)
      const int l2 = qdiv(row, p.mg_W2), j = row - l2 * p.W2;
      const int n = qdiv(l2, p.mg_H2), ii = l2 - n * p.H2;
      const long ob = (((long)n * p.H + 2 * ii) * p.W + 2 * j) * p.Nc + c4;
      const bool h1 = 2 * ii + 1 < p.H, w1 = 2 * j + 1 < p.W;
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        if (((h >> 1) && !h1) || ((h & 1) && !w1)) continue;  // the missing outputs of an odd height / width
        f32x4 v = out[h];
        const long o = ob + (h >> 1) * wrow + (h & 1) * p.Nc;
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
    };
#pragma unroll 1
    for (int k4 = 0; k4 < 4; ++k4) {
      bool ok_a;
      const unsigned oa = item_off(k4, ok_a);
      if (!ok_a) continue;
      f32x4 ma[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) ma[t] = buf_load4_spol<kSc1>(rs, oa, (unsigned)((long)(t * cs) * sstride * 4));      // 16 loads in flight
      for (int sl = 1; sl < cs; ++sl) {
#pragma unroll
        for (int t0 = 0; t0 < 16; t0 += 8) {
          f32x4 pa[8];
#pragma unroll
          for (int t = 0; t < 8; ++t) pa[t] = buf_load4_spol<kSc1>(rs, oa, (unsigned)((long)((t0 + t) * cs + sl) * sstride * 4));
#pragma unroll
          for (int t = 0; t < 8; ++t) ma[t0 + t] += pa[t];
        }
      }
      finish_item(k4, ma);
    }
    if (p.bn_part) {
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
  if (partial) {
    // split tile: sc1 stores of the partial quad sums, drain, ticket; the last slice sums the slabs in slice order and finishes
    __shared__ int ticket;
    float* const base = in_tail ? p.tail_out : p.main_out;
    const long sstride = in_tail ? p.tail_slab_stride : p.main_slab_stride;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)((long)nslices * sstride * 4), 0x00020000);
    const int cl = n0 + wn * 32 + col_l;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + row_h;
      const long off = (long)kslice * sstride + (long)(row - row0) * 4 * p.Nc + cl;
      const bool ok = row < p.M4 && cl < p.Nc;
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y00[e]), rs, ok ? (int)(off * 4) : -1, 0, kSc1);
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y01[e]), rs, ok ? (int)((off + p.Nc) * 4) : -1, 0, kSc1);
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y10[e]), rs, ok ? (int)((off + 2 * p.Nc) * 4) : -1, 0, kSc1);
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y11[e]), rs, ok ? (int)((off + 3 * p.Nc) * 4) : -1, 0, kSc1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) ticket = __hip_atomic_fetch_add(&p.tile_counters[logical], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (ticket != nslices - 1) return;
    if (tid == 0) __hip_atomic_store(&p.tile_counters[logical], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // zero at rest
    float fs1[4] = {0.f, 0.f, 0.f, 0.f}, fs2[4] = {0.f, 0.f, 0.f, 0.f};
    // 256 slab rows (quad row, sub-pixel) x 16 float4 columns: thread -> sub-pixel h = (tid >> 4) & 3, quad rows (tid >> 6) + 4 k
    const int h = (tid >> 4) & 3, c4 = n0 + (tid & 15) * 4;
    for (int k16 = 0; k16 < 16; ++k16) {
      const int row = m0 + (tid >> 6) + 4 * k16;
      if (row >= p.M4 || c4 >= p.Nc) continue;
      const int l2 = qdiv(row, p.mg_W2), j = row - l2 * p.W2;
      const int n = qdiv(l2, p.mg_H2), ii = l2 - n * p.H2;
      const int oh = 2 * ii + (h >> 1), ow = 2 * j + (h & 1);
      if (oh >= p.H || ow >= p.W) continue;          // the missing outputs of an odd height / width
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      for (int sl0 = 0; sl0 < nslices; sl0 += 8) {
        f32x4 part[8];
#pragma unroll
        for (int sl = 0; sl < 8; ++sl)
          if (sl0 + sl < nslices) part[sl] = buf_load4_pol(rs, (unsigned)(((long)(sl0 + sl) * sstride + ((long)(row - row0) * 4 + h) * p.Nc + c4) * 4), kSc1);
#pragma unroll
        for (int sl = 0; sl < 8; ++sl)
          if (sl0 + sl < nslices) v += part[sl];
      }
      const long o = (((long)n * p.H + oh) * p.W + ow) * p.Nc + c4;
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
    if (p.bn_part) {
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
  // ---- row-major float4 epilogue: the four accumulator tiles go through LDS one after the other (the k loop ended with a
  // barrier); every thread finishes four float4 pieces of output rows per tile ----
  constexpr int LDT = 68;
  float* tl = &lds[0][0];                                  // 64 x 68 floats (BK 16: 5120 available)
  float fs1[4] = {0.f, 0.f, 0.f, 0.f}, fs2[4] = {0.f, 0.f, 0.f, 0.f};
  // this thread's four quad rows (tid >> 4) + 16 k, columns c4: output pixel base and which of the four sub-pixels exist — computed
  // ONCE (round 4: it was recomputed, with two runtime divisions, for every sub-pixel pass)
  const int c4 = n0 + (tid & 15) * 4;
  int pb[4];
  unsigned okm[4];
#pragma unroll
  for (int k4 = 0; k4 < 4; ++k4) {
    const int row = m0 + (tid >> 4) + 16 * k4;
    const int l2 = qdiv(row, p.mg_W2), j = row - l2 * p.W2;
    const int n = qdiv(l2, p.mg_H2), ii = l2 - n * p.H2;
    pb[k4] = ((n * p.H + 2 * ii) * p.W + 2 * j) * p.Nc + c4;
    const bool ok = row < p.M4 && c4 < p.Nc, h1 = 2 * ii + 1 < p.H, w1 = 2 * j + 1 < p.W;
    okm[k4] = ok ? (1u | (w1 ? 2u : 0u) | (h1 ? 4u : 0u) | ((h1 && w1) ? 8u : 0u)) : 0u;
  }
  const int wrow = p.W * p.Nc;
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    if (h) __syncthreads();
#pragma unroll
    for (int e = 0; e < 16; ++e)
      tl[(wm * 32 + (e & 3) + 8 * (e >> 2) + row_h) * LDT + wn * 32 + col_l] = h == 0 ? y00[e] : (h == 1 ? y01[e] : (h == 2 ? y10[e] : y11[e]));
    __syncthreads();
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      if (!((okm[k4] >> h) & 1u)) continue;
      f32x4 v = *reinterpret_cast<const f32x4*>(tl + ((tid >> 4) + 16 * k4) * LDT + (tid & 15) * 4);
      const int o = pb[k4] + (h >> 1) * wrow + (h & 1) * p.Nc;
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
  if (p.bn_part) {
    __syncthreads();
    float* red = &lds[0][0];
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
}

// ---- schedule: the balanced plan of conv2d.hip (plan_balance_tile) for I = 16 * C / BK iterations per tile ----
struct W2Plan {
  int bk, on, main_ks, n_main_tiles, tail_slices, tail_row0;
  size_t main_floats, tail_floats;
  double t_us;
  int pos_cs;          // > 0: the position-split instantiation with that many channel slices per position (main_floats = its M slabs)
};

// Cost of the position-split schedule (us): T tiles x 16 positions x cs channel slices, every workgroup runs C / 32 / cs iterations of the
// BK 32 loop with one accumulator; a CU's nb workgroups, c of them co-resident, cost nb * I * (b + a / c) like the plan above (same loop,
// same fitted a / b); per workgroup generation a prologue + one-tile slab store, then per tile the last arriver's serial tail: 16 * cs slab
// tiles read back (4 dependent rounds of 16 loads per slice count) + the four-pixel epilogue.  Constants set from forced-mode sweeps at
// 8 / 16 / 32 images (tools/bench_conv.py --ab NNL_WINO2_POS=0,1..., profiles/r5_wino2_pos_*.log).
double w2_pos_cost(long T, long M4, int Nc, int C, int cs) {
  const double a = 0.363, b = 0.490, tfix = 4.0;
  const long occ = 4, nwg = T * 16 * cs, I = (C / 32) / cs;
  const long nb = nnl_cdiv(nwg, (long)kCUs);
  const double c = (double)(nb < occ ? nb : occ);
  const double slab_b = 2.0 * 16 * cs * (double)M4 * Nc * 4;
  // (+ 4 us of launch: the filter pass is not charged — Learner's steps transform all filters in one batched launch, ops.prepare_forward;
  // measured kernel-only at 8 images: 24.2 us against the direct kernel's 28.5, profiles/r5_wino2_pos_bs8_kernel_stats.csv)
  return nb * I * (b + a / c) + tfix * nb / c + slab_b / 11.7e6 + (3.0 + 1.5 * cs) + 4.0;
}

// Cost of one schedule (us), fitted to 477 forced-schedule timings of this kernel (tools/wino2_plan_sweep.py, profiles/r3_wino2d_plan_sweep.log:
// five layer shapes x 16 / 32 / 64 images x both k blocks x 21 (main slices, tail slices) settings, rms error 9 %).  A CU's busiest set of
// workgroups — m main + q tail blocks, up to `occ` of them resident — costs W * (b + a / c): W their k iterations, c = min(occ, m + q)
// the co-resident count; `a` is the per-iteration latency a lone workgroup cannot hide (five loads, one barrier), `b` the MFMA / LDS
// share that co-resident workgroups divide.  Plus a fixed cost per workgroup generation (prologue, 16 folds, four-tile epilogue),
// the slab round trips at a fitted 11.7 TB/s and 10 us of launch + filter pre-pass.
double w2_cost(long T, long gn, long M4, int Nc, long I, int bk, int ks, int S, W2Plan* out) {
  const double a = bk == 32 ? 0.363 : 0.226, b = bk == 32 ? 0.490 : 0.243, tfix = bk == 32 ? 12.3 : 15.0;
  const long occ = (bk == 16 && NNL_AB_INT("NNL_WINO2_OCC", 4) != 3) ? 4 : 3;
  long n_main = ((T * ks / kCUs) * kCUs / ks / gn) * gn;
  if (n_main > T) n_main = T;
  const long tail = T - n_main;
  if (tail == 0 && S > 1) return -1.0;
  const long m = nnl_cdiv(n_main * ks, (long)kCUs), q = nnl_cdiv(tail * S, (long)kCUs);
  const long W = m * nnl_cdiv(I, (long)ks) + q * nnl_cdiv(I, (long)S), nb = m + q;
  const double c = (double)(nb < occ ? nb : occ);
  const long row0 = (n_main / gn) * 64 < M4 ? (n_main / gn) * 64 : M4;
  const double main_b = ks > 1 ? (2.0 * ks + 1) * row0 * 4 * Nc * 4 : 0;
  const double tail_b = (S > 1 && tail) ? (2.0 * S + 1) * (M4 - row0) * 4 * Nc * 4 : 0;
  if (out) {
    out->bk = bk; out->on = (ks > 1 || (S > 1 && tail)) ? 1 : 0;
    out->main_ks = ks; out->n_main_tiles = (int)n_main; out->tail_slices = tail ? S : 1; out->tail_row0 = (int)row0;
    out->main_floats = ks > 1 ? (size_t)ks * row0 * 4 * Nc : 0;
    out->tail_floats = (S > 1 && tail) ? (size_t)S * (M4 - row0) * 4 * Nc : 0;
  }
  return W * (b + a / c) + tfix * nb / c + (main_b + tail_b) / 11.7e6 + 10.0;
}

W2Plan wino2_plan(long M4, int Nc, int C) {
  W2Plan best{};
  const long gm = nnl_cdiv(M4, 64), gn = nnl_cdiv(Nc, 64), T = gm * gn;
  const int e_bk = NNL_AB_INT("NNL_WINO2_BK", 0);
  const int f_ks = NNL_ENV_INT("NNL_WINO_PLAN_KS", 0), f_S = NNL_ENV_INT("NNL_WINO_PLAN_S", 0);
  const bool balance = NNL_ENV_INT("NNL_WINO_BALANCE", 1) != 0;
  double best_t = 1e300;
  static const int kSlices[] = {1, 2, 3, 4, 6, 8, 12, 16};
  for (int bk = 16; bk <= 32; bk *= 2) {
    if (C % bk != 0 || ((e_bk == 16 || e_bk == 32) && bk != e_bk && C % e_bk == 0)) continue;
    const long I = 16L * (C / bk);
    for (int ks = 1; ks <= 4; ks *= 2) {
      if (ks > 1 && (!balance || I / ks < 8)) break;
      if (f_ks > 0 && ks != f_ks && balance) continue;
      for (int S : kSlices) {
        if (S > 1 && (!balance || I / S < 4)) break;
        W2Plan cand{};
        const double t = w2_cost(T, gn, M4, Nc, I, bk, ks, S, &cand);
        if (t < 0) break;                                                  // no tail tiles: S is meaningless beyond 1
        if (f_S > 0 && balance && cand.n_main_tiles < T && S != f_S) continue;
        if (t < best_t) { best_t = t; best = cand; best.t_us = t; }
      }
    }
  }
  if (best_t == 1e300) {                                                   // (forced settings the shape does not allow)
    const int bk = (e_bk == 32 && C % 32 == 0) ? 32 : 16;
    best.t_us = w2_cost(T, gn, M4, Nc, 16L * (C / bk), bk, 1, 1, &best);
    best_t = best.t_us;
  }
  // the position-split instantiation (small grids): NNL_WINO2_POS = -1 (default) by predicted time, 0 never, n > 0 forces n channel slices
  const int e_pos = NNL_ENV_INT("NNL_WINO2_POS", -1);
  if (e_pos != 0 && (balance || e_pos > 0) && C % 32 == 0 && (e_bk == 0 || e_bk == 32)) {      // (a split schedule: off with NNL_WINO_BALANCE=0)
    const int csteps = C / 32;
    double pt = 1e300; int pcs = 0;
    for (int cs = 1; cs <= 8; cs *= 2) {
      if (csteps % cs != 0 || (csteps / cs < 2 && cs > 1)) break;
      if (e_pos > 0 && cs != e_pos && csteps % e_pos == 0) continue;
      const size_t fl = (size_t)16 * cs * M4 * Nc;
      if (fl * sizeof(float) >= (1UL << 31)) break;
      const double t = w2_pos_cost(T, M4, Nc, C, cs);
      if (t < pt) { pt = t; pcs = cs; }
    }
    if (pcs && (e_pos > 0 || pt < best_t)) {
      best = W2Plan{};
      best.bk = 32; best.on = 1; best.main_ks = 16 * pcs; best.n_main_tiles = (int)T; best.tail_slices = 1; best.tail_row0 = (int)M4;
      best.main_floats = (size_t)16 * pcs * M4 * Nc; best.tail_floats = 0; best.t_us = pt; best.pos_cs = pcs;
    }
  }
  return best;
}

size_t align4(size_t floats) { return (floats + 3) & ~(size_t)3; }

long quads(int N, int H, int W) { return (long)N * ((H + 1) / 2) * ((W + 1) / 2); }

}  // namespace

bool nnl_wino2_ok(int N, int H, int W, int Cin, int Nc, int R, int S, int stride, int pad) {
  if (R != 3 || S != 3 || stride != 1 || pad != 1 || W < 2 || H < 2 || Cin % 16 != 0 || Nc % 4 != 0) return false;
  const long a_b = (long)N * H * W * Cin * 4, b_b = (long)Nc * 16 * Cin * 4, y_b = (long)N * H * W * Nc * 4;
  if ((quads(N, H, W) + 128) * (long)std::max((W + 1) / 2, (H + 1) / 2) >= (1L << 32)) return false;        // the kernel's multiply-high divisions
  return a_b < (1L << 31) && b_b < (1L << 31) && y_b < (1L << 31);
}

double nnl_wino2_plan_time_us(int N, int H, int W, int Cin, int Nc) { return wino2_plan(quads(N, H, W), Nc, Cin).t_us; }
bool nnl_wino2_plan_is_pos(int N, int H, int W, int Cin, int Nc) { return wino2_plan(quads(N, H, W), Nc, Cin).pos_cs > 0; }

size_t nnl_wino2_workspace_bytes(int N, int H, int W, int Cin, int Nc) {
  const W2Plan pl = wino2_plan(quads(N, H, W), Nc, Cin);
  return (align4((size_t)Nc * 16 * Cin) + (pl.on ? pl.main_floats + pl.tail_floats : 0)) * sizeof(float);
}

int nnl_wino2_bn_rows(int N, int H, int W) { return (int)nnl_cdiv(quads(N, H, W), 64L); }

int nnl_wino2_launch(const WinoProblem& q, void* ws, size_t ws_bytes, int* tile_counters, long n_counters, hipStream_t s) {
  const long M4 = quads(q.N, q.H, q.W);
  const size_t u_floats = align4((size_t)q.Nc * 16 * q.Cin);
  if (ws == nullptr || ws_bytes < u_floats * sizeof(float)) return nnl_set_error(NNL_ERR_WORKSPACE, "wino2: workspace too small");
  float* u = (float*)ws;
  if (q.u_pre == nullptr) {
    const long KC = (long)q.Nc * q.Cin;
    hipLaunchKernelGGL(wino2_filter_kernel, dim3((unsigned)nnl_cdiv(KC, 256L)), dim3(256), 0, s, q.filt, u, KC, q.Cin, q.flip);
    NNL_CHECK_LAUNCH();
  }
  Wino2Params p{};
  p.a = q.in; p.b = q.u_pre ? q.u_pre : u; p.y = q.out; p.bias = q.bias; p.add = q.add;
  p.a_bytes = (unsigned)((long)q.N * q.H * q.W * q.Cin * 4); p.b_bytes = (unsigned)((long)q.Nc * 16 * q.Cin * 4);
  {
    const int e_ch = NNL_ENV_INT("NNL_WINO2_CHUNK", 0);                  // 0: the whole C (position-major order); else a multiple of 32 dividing C
    p.ch = (e_ch > 0 && e_ch % 32 == 0 && q.Cin % e_ch == 0) ? e_ch : q.Cin;
    p.prio = NNL_AB_INT("NNL_WINO2_PRIO", 0);
    p.fold_skip = NNL_AB_INT("NNL_WINO2_FOLD_SKIP", 1);
  }
  p.H = q.H; p.W = q.W; p.C = q.Cin; p.H2 = (q.H + 1) / 2; p.W2 = (q.W + 1) / 2; p.M4 = (int)M4; p.Nc = q.Nc; p.relu = q.relu;
  {
    auto magic = [](int d) { return d <= 1 ? 0u : (unsigned)(((1ULL << 32) + d - 1) / (unsigned)d); };
    p.mg_W2 = magic(p.W2); p.mg_H2 = magic(p.H2);
  }
  p.grid_m = (int)nnl_cdiv(M4, 64L); p.grid_n = (int)nnl_cdiv(q.Nc, 64);
  p.bn_part = q.bn_part; p.bn_pivot = q.bn_pivot;
  const long T = (long)p.grid_m * p.grid_n;
  W2Plan pl = wino2_plan(M4, q.Nc, q.Cin);
  if (pl.on && (tile_counters == nullptr || T > n_counters || ws_bytes < (u_floats + pl.main_floats + pl.tail_floats) * sizeof(float) ||
                pl.main_floats * sizeof(float) >= (1UL << 31) || pl.tail_floats * sizeof(float) >= (1UL << 31))) {
    if (pl.pos_cs) {                                                       // (no counters / workspace for the slabs: the unsplit plan)
      const int bk = q.Cin % 32 == 0 ? 32 : 16;
      w2_cost(T, p.grid_n, M4, q.Nc, 16L * (q.Cin / bk), bk, 1, 1, &pl);
      pl.pos_cs = 0;
    }
    pl.on = 0;
  }
  unsigned grid = (unsigned)T;
  if (pl.pos_cs) {
    p.bal = 1; p.main_ks = pl.main_ks; p.n_main_tiles = pl.n_main_tiles; p.tail_slices = 1; p.tail_row0 = (int)M4;
    p.main_out = u + u_floats; p.main_slab_stride = M4 * q.Nc; p.tile_counters = tile_counters; p.pos_cs = pl.pos_cs;
    grid = (unsigned)(T * pl.main_ks);
    hipLaunchKernelGGL((wino2_kernel<32, 4, true>), dim3(grid), dim3(256), 0, s, p);
    NNL_CHECK_LAUNCH();
    return NNL_OK;
  }
  if (pl.on) {
    p.bal = 1; p.main_ks = pl.main_ks; p.n_main_tiles = pl.n_main_tiles; p.tail_slices = pl.tail_slices; p.tail_row0 = pl.tail_row0;
    p.main_out = u + u_floats; p.main_slab_stride = (long)pl.tail_row0 * 4 * q.Nc;
    p.tail_out = p.main_out + pl.main_floats; p.tail_slab_stride = (long)(M4 - pl.tail_row0) * 4 * q.Nc;
    p.tile_counters = tile_counters;
    grid = (unsigned)(pl.n_main_tiles * pl.main_ks + (T - pl.n_main_tiles) * pl.tail_slices);
  }
  if (pl.bk == 32) hipLaunchKernelGGL((wino2_kernel<32, 3>), dim3(grid), dim3(256), 0, s, p);
  else if (NNL_AB_INT("NNL_WINO2_OCC", 4) == 3) hipLaunchKernelGGL((wino2_kernel<16, 3>), dim3(grid), dim3(256), 0, s, p);
  else hipLaunchKernelGGL((wino2_kernel<16, 4>), dim3(grid), dim3(256), 0, s, p);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

// debug / A-B entry (tools/bench_wino.py --two-d): ws as nnl_debug_conv_wino2_workspace_bytes; counters: >= tiles zeroed int32 or null
extern "C" size_t nnl_debug_conv_wino2_workspace_bytes(int N, int H, int W, int C, int K) { return nnl_wino2_workspace_bytes(N, H, W, C, K); }
extern "C" int nnl_debug_conv_wino2_fwd(const float* x, const float* w, const float* bias, const float* add, float* y, void* ws,
                                        size_t ws_bytes, int32_t* counters, long n_counters, float* bn_part, const float* bn_pivot, int N,
                                        int H, int W, int C, int K, int relu, int flip, void* stream) {
  NNL_CHECK_ARG(nnl_wino2_ok(N, H, W, C, K, 3, 3, 1, 1), "wino2: unsupported shape");
  WinoProblem q{};
  q.in = x; q.filt = w; q.out = y; q.bias = bias; q.add = add; q.N = N; q.H = H; q.W = W; q.Cin = C; q.Nc = K; q.relu = relu; q.flip = flip;
  q.bn_part = bn_part; q.bn_pivot = bn_pivot;
  return nnl_wino2_launch(q, ws, ws_bytes, counters, n_counters, (hipStream_t)stream);
}
