// 3x3 / stride 1 / pad 1 convolution (forward and stride-1 dgrad) as a 2-D Winograd F(2x2, 3x3) with a SPATIALLY STAGED input:
// the successor of wino2.hip for the layers the dispatcher gives it (conv2d.hip: wino_mode == 3).  Same algebra, same GEMM rows
// (a row is a 2x2 output quad), same filter transform, another data path:
//
//   wino2.hip   builds V_{xi nu} while STAGING: four buffer loads + three FMAs per staged element, every position its own pass
//               over the input -> 64 pixel fetches per quad and channel, five loads and a barrier per eight MFMAs (MFMA busy 0.42)
//   this file   stages RAW input rows: a workgroup owns 64 raster-consecutive quads; for the position ROW xi it needs the two
//               pixel rows (xi, partner) of every quad row segment of its tile — 2 x (2 len + 2) pixels per segment, shared by
//               neighbouring quads and by the four positions nu of the row — which go global -> LDS by LDS-DMA (`buffer_load ...
//               lds`, no staging registers, no ds_write), 8 channels per stage.  The row transform T = d[xi] +- d[partner] is
//               one FMA per element while the PREVIOUS stage's MFMAs run, the column transform V = T[nu] +- T[partner] one FMA
//               per MFMA operand: ~8 pixel fetches per quad and channel instead of 64, and the loop has no global -> register
//               loads at all.  The transformed filter comes PRE-TILED (wino2s_filter: [k tile][xi][channel block][nu][128
//               swizzled 16-B slots]) so that a stage's 8 KB of U are one contiguous LDS-DMA.
//
//   k order: position row xi outermost, the C/8 channel blocks inside ("quarters": 16 MFMAs per wave each); the four M_{xi nu}
//   accumulators of the row are folded into the four output tiles when the row ends (Z_q = sum_nu c_q[nu] M_nu, Y_pq += c_p[xi] Z_q):
//   128 accumulator registers, two workgroups per CU (<= 256 VGPRs) whose barriers / epilogues cover each other.
//   Stage ring: 3 x (12 KB raw + 8 KB U); quarter g computes from stage g, reads stage g+1 (next T, next first B fragment) and
//   issues stage g+2; `s_waitcnt vmcnt(0)` + one barrier per quarter.
//
// LDS images (16-B slots, written lane-linear by the DMA, so the SOURCE side applies the layout):
//   raw   [pair row 0/1][column parity][t][channel half ^ ((t >> 3) & 1)]   t = position in the tile's concatenated segments
//         (segment s holds len_s + 1 entries per parity; a quad at t needs entries t and t + 1)
//   U     [nu][2 * k + (channel half ^ ((k >> 3) & 1))]
//   the XOR makes every 16-lane group of a ds_read_b128 hit 16 different 16-B slots mod 16 (conflict-free; t or k = lane + const).
//
// Replaces cuDNN's Winograd convolutions in the reference's 3x3 layers (retinanet.py:43-59,77-97,126-148,187-217,260-295).
// Balanced schedule, slabs, ticket fix-up, float4 epilogue and BatchNorm partials: as wino2.hip (slices are quarter ranges).
#include "wino.h"
#include "wino_filter.h"
#include "igemm_taps.h"
#include <algorithm>
#include <map>
#include <mutex>
#include <tuple>

namespace {

constexpr int kRing = 4;              // slots of the raw ring and of the U ring
constexpr int kRawFloats = 2816;      // 11 x 1 KB pieces: 4 planes x 2 L slots, L = 64 + segments <= 82 (656 slots)
constexpr int kUFloats = 2048;        // 4 positions x 128 slots x 16 B

struct Wino2sParams {
  const float* a;      // in [N][H][W][C]
  const float* u;      // pre-tiled U (wino2s_filter)
  float* y;            // [N][H][W][Nc]
  const float* bias;
  const float* add;
  unsigned a_bytes, u_bytes;
  int N, H, W, C, H2, W2;
  int NB;              // C / 8 channel blocks
  unsigned mg_W2, mg_W21, mg_H2, mg_NB;   // ceil(2^32 / d) for d = W2, W2 + 1, H2, NB: n / d = umulhi(n, mg) (exact: n * d < 2^32, checked by the host)
  int M4;              // N * H2 * W2 quad rows
  int Nc;
  int relu;
  int grid_m, grid_n;
  int bal, main_ks, n_main_tiles, tail_slices, tail_row0;      // as Wino2Params; rows are QUAD rows
  float* main_out; long main_slab_stride;                      // slabs [slices][4 * rows][Nc]
  float* tail_out; long tail_slab_stride;
  int* tile_counters;
  float* bn_part; const float* bn_pivot;
  int dbg;             // timing experiments (NNL_W2S_DBG; results invalid): bit 0 no raw traffic, bit 1 no U traffic, bit 2 no MFMA
};

__global__ __launch_bounds__(256) void wino2s_filter_kernel(const float* __restrict__ w, float* __restrict__ u, int Nc, int C, int flip) {
  wino2s_filter_block(w, u, blockIdx.x, threadIdx.x, Nc, C, flip);
}

// One LDS-DMA piece: 64 lanes x 16 B from buffer offsets (voff + soff) to LDS `dst` + lane * 16.  Inline asm on purpose: hipcc's waitcnt
// pass cannot tell that the ds_reads which follow a `buffer_load ... lds` touch OTHER ring slots and drains vmcnt before each of them
// (the builtin form serialised the loop); the kernel does its own counted waits.  No compiler-tracked VMEM operation is in flight
// anywhere these are (the loop drains with vmcnt(0) before the epilogue).
typedef int w2s_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void w2s_dma(w2s_i32x4 rsrc, const float* dst, unsigned voff, int soff) {
  const unsigned lds_addr = (unsigned)(uintptr_t)NNL_LDSP(dst);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");      // (m0 is a reserved register: not in the clobber list; nothing else in this kernel uses it)
}
__device__ __forceinline__ w2s_i32x4 w2s_rsrc(const float* base, unsigned bytes) {      // raw buffer resource: base, stride 0, num_records = bytes
  const unsigned long long a = (unsigned long long)(uintptr_t)base;
  return w2s_i32x4{(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}
#define W2S_DMA(rsrc, dst, voff, soff) w2s_dma(rsrc, dst, (unsigned)(voff), (int)(soff))

template <int DBG>
__global__ __launch_bounds__(256, 2) void wino2s_kernel(const Wino2sParams p) {
  __shared__ __attribute__((aligned(1024))) float lds_u[kRing][kUFloats];
  __shared__ __attribute__((aligned(1024))) float lds_r[kRing][kRawFloats];
  __shared__ __attribute__((aligned(1024))) float lds_dump[256];      // destination of the twelfth raw piece (no slots: never read); epilogue: the ticket word
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
  const int h = lane >> 5, l31 = lane & 31;
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
  const int m0 = tile_m * 64, n0 = tile_n * 64;
  const int NB = p.NB, G = 4 * NB;
  int g0 = 0, g1 = G;
  if (partial) {
    const int per = (G + nslices - 1) / nslices;
    g0 = kslice * per;
    g1 = min(g0 + per, G);
    if (g1 < g0) g1 = g0;
  }
  if (DBG & 8) g1 = g0;                                               // (timing: prologue + epilogue only)

  const w2s_i32x4 ra_src = w2s_rsrc(p.a, p.a_bytes), ru_src = w2s_rsrc(p.u, p.u_bytes);

  // ---- tile geometry: 64 raster-consecutive quads = segments of quad rows; segment s holds len_s + 1 entries per column parity ----
  auto qdiv = [](int n, unsigned mg) { return mg ? (int)__umulhi((unsigned)n, mg) : n; };          // mg == 0: d = 1
  const int W2 = p.W2, W21 = W2 + 1;
  const int l20 = qdiv(m0, p.mg_W2), j0 = m0 - l20 * W2;
  const int len0 = min(W2 - j0, 64);
  const int nseg = 1 + qdiv(64 - len0 + W2 - 1, p.mg_W2);
  const int L = 64 + nseg, L2 = 2 * L;
  // this thread's three raw DMA slots (piece = wave * 3 + i; every wave issues three pieces per quarter, slots past 8 L are out of
  // range): byte offset of pixel row 2i - 1 (row a = 0 of the quad row's patch), row validity bits, pair row
  int raw_off[3];
  unsigned raw_rm[3];
  const int row_bytes = p.W * p.C * 4;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int sigma = (wave * 3 + i) * 64 + lane;
    const int pl = (sigma >= L2) + (sigma >= 2 * L2) + (sigma >= 3 * L2) + (sigma >= 4 * L2), rem = sigma - pl * L2;
    const int t = rem >> 1, hh = (rem & 1) ^ ((t >> 3) & 1);
    raw_off[i] = 0;
    raw_rm[i] = 0;
    if (pl < 4) {
      int s, idx;
      if (t <= len0) { s = 0; idx = t; }
      else { const int tt = t - len0 - 1; s = 1 + qdiv(tt, p.mg_W21); idx = tt - (s - 1) * W21; }
      const int l2 = l20 + s, js = s == 0 ? j0 : 0;
      const int n = qdiv(l2, p.mg_H2), i2 = l2 - n * p.H2;
      const int x = 2 * js - 1 + 2 * idx + (pl & 1);
      if (n < p.N && (unsigned)x < (unsigned)p.W) {
        unsigned rm = 0, vm = 0;
        for (int a = 0; a < 4; ++a) if ((unsigned)(2 * i2 - 1 + a) < (unsigned)p.H) rm |= 1u << a;
        // bit x (0..3): the slot's pixel row exists for position row xi = x; bit 4: the slot belongs to the PARTNER row of the pair
        for (int x4 = 0; x4 < 4; ++x4) vm |= ((rm >> ((pl >> 1) ? (x4 < 2 ? 2 : 1) : x4)) & 1u) << x4;
        raw_rm[i] = vm | ((unsigned)(pl >> 1) << 4);
        raw_off[i] = (((n * p.H + 2 * i2 - 1) * p.W + x) * p.C + hh * 4) * 4;
      }
    }
  }
  // reads: quad r of the tile sits at entry tq; patch column b = 0..3 -> parity b & 1, entry tq + (b >> 1)
  int o0, o1;
  {
    const int r = wm * 32 + l31;
    int s_r, jj;
    if (r < len0) { s_r = 0; jj = r; }
    else { const int rr = r - len0; s_r = 1 + qdiv(rr, p.mg_W2); jj = rr - (s_r - 1) * W2; }
    const int tq = (s_r == 0 ? 0 : len0 + 1 + (s_r - 1) * W21) + jj;
    o0 = (2 * tq + (h ^ ((tq >> 3) & 1))) * 4;
    o1 = (2 * (tq + 1) + (h ^ (((tq + 1) >> 3) & 1))) * 4;
  }
  const int pstr = L2 * 4;                                           // floats per raw plane
  const int kl = wn * 32 + l31;
  const int u_rd = (2 * kl + (h ^ ((kl >> 3) & 1))) * 4;
  const unsigned u_vo0 = (unsigned)((wave * 2) * 1024 + lane * 16), u_vo1 = u_vo0 + 1024u;
  const int u_tile = tile_n * G * 8192;                              // bytes

  // quarter q = (position row xi, channel block b).  U ring: quarter g reads slot g % 4 (and the first fragment of g+1), U(g+3) is
  // issued in it; raw ring: quarter g reads raw(g+1) (the row transform of the next quarter) and issues raw(g+4) into the slot of
  // raw(g), which died when T(g) was formed.  Every wave issues FIVE pieces per quarter (out-of-range ones past the end of its slice or
  // of the raw image: the DMA writes zeros into a free slot / the dump), so the counted wait before the barrier is a constant.
  auto dma_u = [&](int i, int g, int slot, bool ok) {
    if (DBG & 2) ok = false;
    W2S_DMA(ru_src, &lds_u[slot][0] + (wave * 2 + i) * 256, ok ? (i ? u_vo1 : u_vo0) : 0xFFFFFFFFu, u_tile + g * 8192);
  };
  auto dma_raw = [&](int i, int xq, int bq, int slot, bool ok) {
    if (DBG & 1) ok = false;
    const int arow = (raw_rm[i] & 16u) ? (xq < 2 ? 2 : 1) * row_bytes : xq * row_bytes;      // select between two scalars
    const unsigned cand = (unsigned)(raw_off[i] + arow);
    unsigned vo = (ok & (((raw_rm[i] >> xq) & 1u) != 0)) ? cand : 0xFFFFFFFFu;
    if (DBG & 16) vo = (unsigned)((((long)(tile_m * G + xq * NB + bq) * 11 + wave * 3 + i) * 1024 + lane * 16) % (long)(p.a_bytes - 64)) & ~15u;   // (timing: CONTIGUOUS pieces, same bytes)
    float* dst = (wave * 3 + i < 11) ? &lds_r[slot][0] + (wave * 3 + i) * 256 : &lds_dump[0];
    W2S_DMA(ra_src, dst, vo, bq * 32);
  };
  // raw rows of the stage in `slot` for this lane's quad and channel half, patch column b: d1 = row xi, d2 = its partner
  auto rd_raw = [&](int slot, int b, f32x4& d1, f32x4& d2) {
    const float* rs = &lds_r[slot][0] + (b & 1) * pstr + ((b >> 1) ? o1 : o0);
    d1 = *reinterpret_cast<const f32x4*>(rs);
    d2 = *reinterpret_cast<const f32x4*>(rs + 2 * pstr);
  };
  auto rowt = [](float sr, const f32x4& d1, const f32x4& d2) {        // T = d[xi] +- d[partner]; sr = +-1: exact sums
    f32x4 t;
#pragma unroll
    for (int e = 0; e < 4; ++e) t[e] = __builtin_fmaf(sr, d2[e], d1[e]);
    return t;
  };

  f32x16 m[4], y[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) { m[i][e] = 0.f; y[i][e] = 0.f; }

  int xi = __builtin_amdgcn_readfirstlane(qdiv(g0, p.mg_NB)), b = g0 - xi * NB;      // current quarter (wave-uniform: scalar registers)
  int xr = xi, br = b;                                                // the quarter whose raw rows are issued next
  auto adv = [&](int& x, int& bb) { if (++bb == NB) { bb = 0; ++x; } };
  f32x4 Tc[4], Bc, Ac;
  if (g0 < g1) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k < 3) { dma_u(0, g0 + k, k, g0 + k < g1); dma_u(1, g0 + k, k, g0 + k < g1); }
#pragma unroll
      for (int i = 0; i < 3; ++i) dma_raw(i, xr, br, k, g0 + k < g1);
      adv(xr, br);
    }
  }
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                     // stages 0 and 1; U(2), raw(2), raw(3) stay in flight
  __syncthreads();
  if (g0 < g1) {
    const float sr = xi == 1 ? 1.f : -1.f;
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) { f32x4 d1, d2; rd_raw(0, bb, d1, d2); Tc[bb] = rowt(sr, d1, d2); }
    Bc = *reinterpret_cast<const f32x4*>(&lds_u[0][0] + u_rd);
#pragma unroll
    for (int e = 0; e < 4; ++e) Ac[e] = Tc[0][e] - Tc[2][e];
  }
  for (int g = g0; g < g1; ++g) {
    const int st = (g - g0) & 3, sn = (st + 1) & 3;
    int xin = xi, bn = b;
    adv(xin, bn);
    const float srn = xin == 1 ? 1.f : -1.f;
    const float* us = &lds_u[st][0] + u_rd;
    const bool ok_u = g + 3 < g1, ok_r = g + 4 < g1;
    // 16 MFMAs, the quarter's other work pinned into their shadows (a 32x32x2 fp32 MFMA occupies the pipe for 64 cycles; whatever
    // this wave issues meanwhile is free, whatever it issues in a block before or after them is not): slots 0-4 the five DMA
    // pieces, 5-8 the raw rows of the next quarter, 9-12 their row transform, the column transform of position nu+1 and its U
    // fragment during position nu.
    f32x4 d1[4], d2[4], Tn[4], An, Bn;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int nu = s >> 2, e = s & 3;
      if (DBG & 4) m[nu][e] += Ac[e] * Bc[e];
      else m[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ac[e], Bc[e], m[nu], 0, 0, 0);
      if (s == 0) dma_u(0, g + 3, (st + 3) & 3, ok_u);                // slot of U(g-1): every wave left it at the last barrier
      if (s == 1) dma_u(1, g + 3, (st + 3) & 3, ok_u);
      if (s >= 2 && s <= 4) dma_raw(s - 2, xr, br, st, ok_r);         // slot of raw(g): dead since T(g) was formed in quarter g-1
      if (s >= 5 && s <= 8) rd_raw(sn, s - 5, d1[s - 5], d2[s - 5]);  // raw(g+1) landed before the last barrier
      if (s >= 9 && s <= 12) Tn[s - 9] = rowt(srn, d1[s - 9], d2[s - 9]);
      if (e == 0) Bn = nu < 3 ? *reinterpret_cast<const f32x4*>(us + (nu + 1) * 512) : *reinterpret_cast<const f32x4*>(&lds_u[sn][0] + u_rd);
      if (e >= 1) {                                                    // A of the next position: V = T[nu'] +- T[partner]
        if (nu == 0) { if (e == 1) { An[0] = Tc[1][0] + Tc[2][0]; An[1] = Tc[1][1] + Tc[2][1]; } else An[e] = Tc[1][e] + Tc[2][e]; }
        if (nu == 1) { if (e == 1) { An[0] = Tc[2][0] - Tc[1][0]; An[1] = Tc[2][1] - Tc[1][1]; } else An[e] = Tc[2][e] - Tc[1][e]; }
        if (nu == 2) { if (e == 1) { An[0] = Tc[3][0] - Tc[1][0]; An[1] = Tc[3][1] - Tc[1][1]; } else An[e] = Tc[3][e] - Tc[1][e]; }
        if (nu == 3) { if (e == 1) { An[0] = Tn[0][0] - Tn[2][0]; An[1] = Tn[0][1] - Tn[2][1]; } else An[e] = Tn[0][e] - Tn[2][e]; }
      }
      if (e == 3) { Ac = An; Bc = Bn; }
      __builtin_amdgcn_sched_barrier(0);
    }
    adv(xr, br);
#pragma unroll
    for (int i = 0; i < 4; ++i) Tc[i] = Tn[i];
    if (xin != xi || g + 1 == g1) {
      // the position row (or this slice of it) is complete: Z_q = sum_nu c_q[nu] M_nu, Y_pq += c_p[xi] Z_q; c_0 = (1,1,1,0), c_1 = (0,1,-1,1)
      f32x16 z0, z1;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        z0[e] = (m[0][e] + m[1][e]) + m[2][e];
        z1[e] = (m[1][e] - m[2][e]) + m[3][e];
      }
      if (xi < 3) {
#pragma unroll
        for (int e = 0; e < 16; ++e) { y[0][e] += z0[e]; y[1][e] += z1[e]; }
      }
      if (xi > 0) {
        const float c1 = xi == 2 ? -1.f : 1.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) { y[2][e] = __builtin_fmaf(c1, z0[e], y[2][e]); y[3][e] = __builtin_fmaf(c1, z1[e], y[3][e]); }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) m[i][e] = 0.f;
    }
    xi = xin; b = bn;
    // before quarter g+1: U(g+2) (issued in quarter g-1, BEFORE that quarter's raw pieces) and raw(g+2) (older) must have landed; the
    // eight pieces issued after U(g+2) — raw(g+3), U(g+3), raw(g+4) — may stay in flight
    if (DBG & 32) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");   // (timing: one more quarter of slack — a data race)
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __syncthreads();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // drain the out-of-range tail pieces before the LDS is reused
  __syncthreads();

  // ---- epilogue: quad row -> pixels (2i + p, 2j + q); the ring is idle (the loop ended with a drained barrier) ----
  const int col_l = l31, row_h = h * 4;
  constexpr int kSc1 = 1 << 4;
  float* const scratch = &lds_r[0][0];                  // 44 KB contiguous
  int* const ticket = reinterpret_cast<int*>(&lds_dump[0]);
  // output pixel of quad row `row`, sub-pixel hq = 2 p + q: float offset, or -1 where the tile / an odd image ends
  auto quad_base = [&](int row, int c4, int& pb, unsigned& okm) {
    const int l2 = qdiv(row, p.mg_W2), j = row - l2 * p.W2;
    const int n = qdiv(l2, p.mg_H2), ii = l2 - n * p.H2;
    pb = ((n * p.H + 2 * ii) * p.W + 2 * j) * p.Nc + c4;
    const bool ok = row < p.M4 && c4 < p.Nc, h1 = 2 * ii + 1 < p.H, w1 = 2 * j + 1 < p.W;
    okm = ok ? (1u | (w1 ? 2u : 0u) | (h1 ? 4u : 0u) | ((h1 && w1) ? 8u : 0u)) : 0u;      // bit hq
  };
  auto finish = [&](f32x4 v, int o, int c4, float (&fs1)[4], float (&fs2)[4]) {
    if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + c4);
    if (p.add) v += *reinterpret_cast<const f32x4*>(p.add + o);
    if (p.relu == 1) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
    *reinterpret_cast<f32x4*>(p.y + o) = v;
    if (p.bn_part) {
      const f32x4 pv = *reinterpret_cast<const f32x4*>(p.bn_pivot + c4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = v[e] - pv[e]; fs1[e] += d; fs2[e] += d * d; }
    }
  };
  auto bn_reduce = [&](const float (&fs1)[4], const float (&fs2)[4]) {
    __syncthreads();
    float* red = scratch;                          // [16 row lanes][64 cols][2]
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[(((tid >> 4) * 64) + (tid & 15) * 4 + e) * 2 + 0] = fs1[e];
      red[(((tid >> 4) * 64) + (tid & 15) * 4 + e) * 2 + 1] = fs2[e];
    }
    __syncthreads();
    if (tid < 64 && n0 + tid < p.Nc) {
      float a = 0.f, bsum = 0.f;
      for (int r = 0; r < 16; ++r) { a += red[(r * 64 + tid) * 2]; bsum += red[(r * 64 + tid) * 2 + 1]; }
      p.bn_part[((long)tile_m * p.Nc + n0 + tid) * 2 + 0] = a;
      p.bn_part[((long)tile_m * p.Nc + n0 + tid) * 2 + 1] = bsum;
    }
  };
  const int wrow = p.W * p.Nc;
  float fs1[4] = {0.f, 0.f, 0.f, 0.f}, fs2[4] = {0.f, 0.f, 0.f, 0.f};
  if (partial) {
    // split tile: sc1 stores of the partial quad sums, drain, ticket; the last slice sums the slabs in slice order and finishes
    float* const base = in_tail ? p.tail_out : p.main_out;
    const long sstride = in_tail ? p.tail_slab_stride : p.main_slab_stride;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)((long)nslices * sstride * 4), 0x00020000);
    const int cl = n0 + wn * 32 + col_l;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + row_h;
      const long off = (long)kslice * sstride + (long)(row - row0) * 4 * p.Nc + cl;
      const bool ok = row < p.M4 && cl < p.Nc;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y[q][e]), rs, ok ? (int)((off + q * p.Nc) * 4) : -1, 0, kSc1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) *ticket = __hip_atomic_fetch_add(&p.tile_counters[logical], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (*ticket != nslices - 1) return;
    if (tid == 0) __hip_atomic_store(&p.tile_counters[logical], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // zero at rest
    // 256 slab rows (quad row, sub-pixel) x 16 float4 columns: thread -> sub-pixel hq = (tid >> 4) & 3, quad rows (tid >> 6) + 4 k
    const int hq = (tid >> 4) & 3, c4 = n0 + (tid & 15) * 4;
    for (int k16 = 0; k16 < 16; ++k16) {
      const int row = m0 + (tid >> 6) + 4 * k16;
      int pb; unsigned okm;
      quad_base(row, c4, pb, okm);
      if (!((okm >> hq) & 1u)) continue;             // (incl. the missing outputs of an odd height / width)
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      for (int sl0 = 0; sl0 < nslices; sl0 += 8) {
        f32x4 part[8];
#pragma unroll
        for (int sl = 0; sl < 8; ++sl)
          if (sl0 + sl < nslices) part[sl] = buf_load4_pol(rs, (unsigned)(((long)(sl0 + sl) * sstride + ((long)(row - row0) * 4 + hq) * p.Nc + c4) * 4), kSc1);
#pragma unroll
        for (int sl = 0; sl < 8; ++sl)
          if (sl0 + sl < nslices) v += part[sl];
      }
      finish(v, pb + (hq >> 1) * wrow + (hq & 1) * p.Nc, c4, fs1, fs2);
    }
    if (p.bn_part) bn_reduce(fs1, fs2);
    return;
  }
  // ---- row-major float4 epilogue: the four output tiles go through LDS two at a time; thread -> columns c4, quad rows (tid >> 4) + 16 k ----
  constexpr int LDT = 68;
  int pb[4];
  unsigned okm[4];
  const int c4 = n0 + (tid & 15) * 4;
#pragma unroll
  for (int k4 = 0; k4 < 4; ++k4) quad_base(m0 + (tid >> 4) + 16 * k4, c4, pb[k4], okm[k4]);
#pragma unroll
  for (int hp = 0; hp < 2; ++hp) {
    if (hp) __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        scratch[q * 64 * LDT + (wm * 32 + (e & 3) + 8 * (e >> 2) + row_h) * LDT + wn * 32 + col_l] = y[hp * 2 + q][e];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) {
        const int hq = hp * 2 + q;
        if (!((okm[k4] >> hq) & 1u)) continue;
        const f32x4 v = *reinterpret_cast<const f32x4*>(scratch + q * 64 * LDT + ((tid >> 4) + 16 * k4) * LDT + (tid & 15) * 4);
        finish(v, pb[k4] + hp * wrow + q * p.Nc, c4, fs1, fs2);
      }
  }
  if (p.bn_part) bn_reduce(fs1, fs2);
}

// ---- schedule: the balanced plan of conv2d.hip (plan_balance_tile) for I = 4 * C / 8 quarters per tile ----
struct W2sPlan {
  int on, main_ks, n_main_tiles, tail_slices, tail_row0;
  size_t main_floats, tail_floats;
  double t_us;
};

// Cost of one schedule (us): an event simulation of the launch on 256 CUs with TWO workgroup slots each.  Workgroups are handed out in
// launch order — n_main * ks main workgroups of ~I / ks quarters, then tail * S of ~I / S — to the first free slot; a CU advances
// its resident workgroups by one quarter per tq1 us when one is resident and per tq2 us EACH when two are (they share the matrix
// pipe: two together are only ~1.25x one alone, which is why a plan is not a list schedule on 512 equal slots); a workgroup carries
// ef quarter-equivalents of fixed work (prologue, epilogue) and es more when it is a slice (slab stores / the fix-up); plus the slab
// round trips at `bw` bytes per us and the launch.  Constants fitted to forced-schedule sweeps (tools/wino2s_plan_sweep.py,
// tools/wino2s_fit.py, profiles/r4_wino2s_plan_sweep*.log).
struct W2sModel { double tq1, tq2, ef, es, t0, bw; };
const W2sModel kW2sModel = {0.90, 1.30, 2.0, 3.0, 10.0, 8.0e6};      // rms 8 % over 249 forced schedules (profiles/r4_wino2s_plan_sweep*.log)

double w2s_simulate(long n_main_wg, long q_main_total, int ks, long n_tail_wg, long q_tail_total, int S, const W2sModel& md) {
  // work of workgroup w (quarter-equivalents): slices of a tile get ceil / remaining quarters
  const long per_m = nnl_cdiv(q_main_total, (long)ks), per_t = nnl_cdiv(q_tail_total, (long)S);
  const long n_wg = n_main_wg + n_tail_wg;
  auto work = [&](long w) -> double {
    long q; bool slice;
    if (w < n_main_wg) { const long k = w % ks; q = std::min(per_m, q_main_total - k * per_m); slice = ks > 1; }
    else { const long k = (w - n_main_wg) % S; q = std::min(per_t, q_tail_total - k * per_t); slice = S > 1; }
    if (q < 0) q = 0;
    return (double)q + md.ef + (slice ? md.es : 0.0);
  };
  constexpr int kCU = 256;
  double t[kCU], qa[kCU], qb[kCU];
  long next = 0;
  for (int c = 0; c < kCU; ++c) { t[c] = 0.0; qa[c] = 0.0; qb[c] = 0.0; }
  for (int c = 0; c < kCU && next < n_wg; ++c) qa[c] = work(next++);
  for (int c = 0; c < kCU && next < n_wg; ++c) qb[c] = work(next++);
  auto nxt = [&](int c) { return (qa[c] > 0.0 && qb[c] > 0.0) ? t[c] + std::min(qa[c], qb[c]) * md.tq2 : ((qa[c] > 0.0 || qb[c] > 0.0) ? t[c] + std::max(qa[c], qb[c]) * md.tq1 : 1e300); };
  // binary heap over CUs by next completion time.  All CUs that complete at the same instant are retired together and the free slots
  // are then refilled BREADTH-first (one workgroup per CU and pass), as the dispatcher deals workgroups round-robin over the CUs.
  int heap[kCU], pos_n = 0, batch[kCU];
  double key[kCU];
  auto sift_down = [&](int i) {
    for (;;) {
      int l = 2 * i + 1, r = l + 1, m = i;
      if (l < pos_n && key[heap[l]] < key[heap[m]]) m = l;
      if (r < pos_n && key[heap[r]] < key[heap[m]]) m = r;
      if (m == i) break;
      std::swap(heap[i], heap[m]); i = m;
    }
  };
  auto sift_up = [&](int i) {
    while (i > 0) { const int pa = (i - 1) / 2; if (key[heap[pa]] <= key[heap[i]]) break; std::swap(heap[pa], heap[i]); i = pa; }
  };
  for (int c = 0; c < kCU; ++c) { key[c] = nxt(c); heap[pos_n++] = c; }
  for (int i = pos_n / 2 - 1; i >= 0; --i) sift_down(i);
  double span = 0.0;
  while (pos_n > 0 && key[heap[0]] < 1e299) {
    const double tn = key[heap[0]];
    int nb = 0;
    while (pos_n > 0 && key[heap[0]] <= tn + 1e-9) {                        // retire everything that completes now
      const int c = heap[0];
      heap[0] = heap[--pos_n];
      if (pos_n > 0) sift_down(0);
      if (qa[c] > 0.0 && qb[c] > 0.0) { const double d = std::min(qa[c], qb[c]); qa[c] -= d; qb[c] -= d; }
      else { qa[c] = 0.0; qb[c] = 0.0; }
      if (qa[c] < 1e-9) qa[c] = 0.0;
      if (qb[c] < 1e-9) qb[c] = 0.0;
      t[c] = tn;
      batch[nb++] = c;
    }
    span = tn;
    for (int pass = 0; pass < 2 && next < n_wg; ++pass)
      for (int i = 0; i < nb && next < n_wg; ++i) {
        const int c = batch[i];
        if (qa[c] == 0.0) qa[c] = work(next++);
        else if (qb[c] == 0.0) qb[c] = work(next++);
      }
    for (int i = 0; i < nb; ++i) { const int c = batch[i]; key[c] = nxt(c); heap[pos_n] = c; sift_up(pos_n++); }
  }
  return span;
}

double w2s_cost_model(long T, long gn, long M4, int Nc, long I, long P, int ks, int S, const W2sModel& md, W2sPlan* out) {
  long n_main = ((T * ks / P) * P / ks / gn) * gn;
  if (n_main > T) n_main = T;
  const long tail = T - n_main;
  if (tail == 0 && S > 1) return -1.0;
  const double span = w2s_simulate(n_main * ks, I, ks, tail * S, I, S, md);
  const long row0 = (n_main / gn) * 64 < M4 ? (n_main / gn) * 64 : M4;
  const double main_b = ks > 1 ? (2.0 * ks + 1) * row0 * 4 * Nc * 4 : 0;
  const double tail_b = (S > 1 && tail) ? (2.0 * S + 1) * (M4 - row0) * 4 * Nc * 4 : 0;
  if (out) {
    out->on = (ks > 1 || (S > 1 && tail)) ? 1 : 0;
    out->main_ks = ks; out->n_main_tiles = (int)n_main; out->tail_slices = tail ? S : 1; out->tail_row0 = (int)row0;
    out->main_floats = ks > 1 ? (size_t)ks * row0 * 4 * Nc : 0;
    out->tail_floats = (S > 1 && tail) ? (size_t)S * (M4 - row0) * 4 * Nc : 0;
  }
  return span + (main_b + tail_b) / md.bw + md.t0;
}

double w2s_cost(long T, long gn, long M4, int Nc, long I, long P, int ks, int S, W2sPlan* out) { return w2s_cost_model(T, gn, M4, Nc, I, P, ks, S, kW2sModel, out); }

W2sPlan wino2s_plan_search(long M4, int Nc, int C) {
  W2sPlan best{};
  const long gm = nnl_cdiv(M4, 64), gn = nnl_cdiv(Nc, 64), T = gm * gn;
  const int f_ks = NNL_ENV_INT("NNL_WINO_PLAN_KS", 0), f_S = NNL_ENV_INT("NNL_WINO_PLAN_S", 0), f_P = NNL_ENV_INT("NNL_W2S_UNIT", 0);
  const bool balance = NNL_ENV_INT("NNL_WINO_BALANCE", 1) != 0;
  const long I = 4L * (C / 8);
  double best_t = 1e300;
  static const int kSlices[] = {1, 2, 3, 4, 6, 8, 12, 16};
  for (long P = 256; P <= 512; P *= 2) {                                   // main tiles fill whole rounds of one / two workgroups per CU
    if (f_P > 0 && P != f_P) continue;
    for (int ks = 1; ks <= 4; ks *= 2) {
      if (ks > 1 && (!balance || I / ks < 8)) break;
      if (f_ks > 0 && ks != f_ks && balance) continue;
      for (int S : kSlices) {
        if (S > 1 && (!balance || I / S < 4)) break;
        W2sPlan cand{};
        const double t = w2s_cost(T, gn, M4, Nc, I, P, ks, S, &cand);
        if (t < 0) break;                                                  // no tail tiles: S is meaningless beyond 1
        if (f_S > 0 && balance && cand.n_main_tiles < T && S != f_S) continue;
        if (t < best_t) { best_t = t; best = cand; best.t_us = t; }
      }
    }
  }
  if (best_t == 1e300) best.t_us = w2s_cost(T, gn, M4, Nc, I, 512, 1, 1, &best);   // (forced settings the shape does not allow)
  return best;
}

// the search simulates every candidate (a few ms for a big problem): one search per shape and environment generation
W2sPlan wino2s_plan(long M4, int Nc, int C) {
  struct Key { long M4; int Nc, C, gen; bool operator<(const Key& o) const { return std::tie(M4, Nc, C, gen) < std::tie(o.M4, o.Nc, o.C, o.gen); } };
  static std::mutex mu;
  static std::map<Key, W2sPlan> cache;
  const Key k{M4, Nc, C, nnl_env_generation()};
  std::lock_guard<std::mutex> lock(mu);
  auto it = cache.find(k);
  if (it != cache.end()) return it->second;
  if (cache.size() > 4096) cache.clear();
  const W2sPlan pl = wino2s_plan_search(M4, Nc, C);
  cache[k] = pl;
  return pl;
}

size_t align4(size_t floats) { return (floats + 3) & ~(size_t)3; }
unsigned magic_of(int d) { return d <= 1 ? 0u : (unsigned)(((1ULL << 32) + d - 1) / (unsigned)d); }   // 0 stands for d = 1
long quads(int N, int H, int W) { return (long)N * ((H + 1) / 2) * ((W + 1) / 2); }
size_t u_floats_of(int Nc, int Cin) { return (size_t)((Nc + 63) / 64) * 64 * 16 * Cin; }

}  // namespace

bool nnl_wino2s_ok(int N, int H, int W, int Cin, int Nc, int R, int S, int stride, int pad) {
  if (R != 3 || S != 3 || stride != 1 || pad != 1 || W < 7 || H < 2 || Cin % 8 != 0 || Nc % 4 != 0) return false;    // W2 >= 4
  const long a_b = (long)N * H * W * Cin * 4, b_b = (long)u_floats_of(Nc, Cin) * 4, y_b = (long)N * H * W * Nc * 4;
  const long M4 = quads(N, H, W), dmax = std::max<long>(std::max((W + 1) / 2 + 1, (H + 1) / 2), Cin / 8);
  if ((M4 + 256) * dmax >= (1L << 32) || Cin / 8 < 2) return false;      // the kernel's umulhi divisions (quad rows / W2, / H2, quarters / NB)
  return a_b < (1L << 31) && b_b < (1L << 31) && y_b < (1L << 31);
}

double nnl_wino2s_plan_time_us(int N, int H, int W, int Cin, int Nc) { return wino2s_plan(quads(N, H, W), Nc, Cin).t_us; }

size_t nnl_wino2s_u_floats(int Cin, int Nc) { return u_floats_of(Nc, Cin); }

size_t nnl_wino2s_workspace_bytes(int N, int H, int W, int Cin, int Nc) {
  const W2sPlan pl = wino2s_plan(quads(N, H, W), Nc, Cin);
  return (align4(u_floats_of(Nc, Cin)) + (pl.on ? pl.main_floats + pl.tail_floats : 0)) * sizeof(float);
}

int nnl_wino2s_bn_rows(int N, int H, int W) { return (int)nnl_cdiv(quads(N, H, W), 64L); }

int nnl_wino2s_launch(const WinoProblem& q, void* ws, size_t ws_bytes, int* tile_counters, long n_counters, hipStream_t s) {
  const long M4 = quads(q.N, q.H, q.W);
  const size_t u_floats = align4(u_floats_of(q.Nc, q.Cin));
  if (ws == nullptr || ws_bytes < u_floats * sizeof(float)) return nnl_set_error(NNL_ERR_WORKSPACE, "wino2s: workspace too small");
  float* u = (float*)ws;
  if (q.u_pre == nullptr) {
    hipLaunchKernelGGL(wino2s_filter_kernel, dim3((unsigned)(((q.Nc + 63) / 64) * ((q.Cin + 15) / 16))), dim3(256), 0, s, q.filt, u, q.Nc, q.Cin, q.flip);
    NNL_CHECK_LAUNCH();
  }
  Wino2sParams p{};
  p.a = q.in; p.u = q.u_pre ? q.u_pre : u; p.y = q.out; p.bias = q.bias; p.add = q.add;
  p.a_bytes = (unsigned)((long)q.N * q.H * q.W * q.Cin * 4); p.u_bytes = (unsigned)(u_floats_of(q.Nc, q.Cin) * 4);
  p.N = q.N; p.H = q.H; p.W = q.W; p.C = q.Cin; p.H2 = (q.H + 1) / 2; p.W2 = (q.W + 1) / 2; p.NB = q.Cin / 8;
  p.M4 = (int)M4; p.Nc = q.Nc; p.relu = q.relu;
  p.mg_W2 = magic_of(p.W2); p.mg_W21 = magic_of(p.W2 + 1); p.mg_H2 = magic_of(p.H2); p.mg_NB = magic_of(p.NB);
  p.grid_m = (int)nnl_cdiv(M4, 64L); p.grid_n = (int)nnl_cdiv(q.Nc, 64);
  p.bn_part = q.bn_part; p.bn_pivot = q.bn_pivot;
  p.dbg = NNL_ENV_INT("NNL_W2S_DBG", 0);
  const long T = (long)p.grid_m * p.grid_n;
  W2sPlan pl = wino2s_plan(M4, q.Nc, q.Cin);
  if (pl.on && (tile_counters == nullptr || T > n_counters || ws_bytes < (u_floats + pl.main_floats + pl.tail_floats) * sizeof(float) ||
                pl.main_floats * sizeof(float) >= (1UL << 31) || pl.tail_floats * sizeof(float) >= (1UL << 31)))
    pl.on = 0;
  if (NNL_ENV_INT("NNL_W2S_VERBOSE", 0))
    fprintf(stderr, "wino2s plan: M4 %ld Nc %d C %d tiles %ld on %d ks %d main %d S %d t %.1f us\n", M4, q.Nc, q.Cin, T, pl.on, pl.main_ks, pl.n_main_tiles, pl.tail_slices, pl.t_us);
  unsigned grid = (unsigned)T;
  if (pl.on) {
    p.bal = 1; p.main_ks = pl.main_ks; p.n_main_tiles = pl.n_main_tiles; p.tail_slices = pl.tail_slices; p.tail_row0 = pl.tail_row0;
    p.main_out = u + u_floats; p.main_slab_stride = (long)pl.tail_row0 * 4 * q.Nc;
    p.tail_out = p.main_out + pl.main_floats; p.tail_slab_stride = (long)(M4 - pl.tail_row0) * 4 * q.Nc;
    p.tile_counters = tile_counters;
    grid = (unsigned)(pl.n_main_tiles * pl.main_ks + (T - pl.n_main_tiles) * pl.tail_slices);
  }
  switch (p.dbg) {                                                     // timing experiments (tools/w2s_dbg.py); 0 = the product kernel
    case 1: hipLaunchKernelGGL(wino2s_kernel<1>, dim3(grid), dim3(256), 0, s, p); break;
    case 2: hipLaunchKernelGGL(wino2s_kernel<2>, dim3(grid), dim3(256), 0, s, p); break;
    case 3: hipLaunchKernelGGL(wino2s_kernel<3>, dim3(grid), dim3(256), 0, s, p); break;
    case 4: hipLaunchKernelGGL(wino2s_kernel<4>, dim3(grid), dim3(256), 0, s, p); break;
    case 7: hipLaunchKernelGGL(wino2s_kernel<7>, dim3(grid), dim3(256), 0, s, p); break;
    case 32: hipLaunchKernelGGL(wino2s_kernel<32>, dim3(grid), dim3(256), 0, s, p); break;
    case 16: hipLaunchKernelGGL(wino2s_kernel<16>, dim3(grid), dim3(256), 0, s, p); break;
    case 8: hipLaunchKernelGGL(wino2s_kernel<8>, dim3(grid), dim3(256), 0, s, p); break;
    default: hipLaunchKernelGGL(wino2s_kernel<0>, dim3(grid), dim3(256), 0, s, p); break;
  }
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

// the schedule model alone (host only; tools/wino2s_fit.py fits its constants to forced-schedule sweeps): prm = {tq1, tq2, ef, es, t0, bw}
extern "C" double nnl_debug_w2s_model(long T, long gn, long M4, int Nc, long I, long P, int ks, int S, const double* prm) {
  const W2sModel md{prm[0], prm[1], prm[2], prm[3], prm[4], prm[5]};
  return w2s_cost_model(T, gn, M4, Nc, I, P, ks, S, md, nullptr);
}

// debug / A-B entry (tools/bench_wino.py --two-d-s): ws as nnl_debug_conv_wino2s_workspace_bytes; counters: >= tiles zeroed int32 or null
extern "C" size_t nnl_debug_conv_wino2s_workspace_bytes(int N, int H, int W, int C, int K) { return nnl_wino2s_workspace_bytes(N, H, W, C, K); }
extern "C" int nnl_debug_conv_wino2s_fwd(const float* x, const float* w, const float* bias, const float* add, float* y, void* ws,
                                         size_t ws_bytes, int32_t* counters, long n_counters, float* bn_part, const float* bn_pivot, int N,
                                         int H, int W, int C, int K, int relu, int flip, void* stream) {
  NNL_CHECK_ARG(nnl_wino2s_ok(N, H, W, C, K, 3, 3, 1, 1), "wino2s: unsupported shape");
  WinoProblem q{};
  q.in = x; q.filt = w; q.out = y; q.bias = bias; q.add = add; q.N = N; q.H = H; q.W = W; q.Cin = C; q.Nc = K; q.relu = relu; q.flip = flip;
  q.bn_part = bn_part; q.bn_pivot = bn_pivot;
  return nnl_wino2s_launch(q, ws, ws_bytes, counters, n_counters, (hipStream_t)stream);
}
