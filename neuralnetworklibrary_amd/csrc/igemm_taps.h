// igemm_taps_kernel — second-generation implicit-GEMM (rows gathered through a TAP TABLE) on the exact-fp32 MFMA.
//
//   C[m][n] = sum_{t < ntaps} sum_{c < C} A[pix(m) + tap t][c] * B[n][woff_t + c]        (C % BK == 0)
//
// Row m enumerates (img, p, q); its base input pixel is (p*in_stride + ih0, q*in_stride + iw0); tap t adds (dh_t, dw_t)
// and is skipped (zero) when it falls outside [0,H)x[0,W).  The output pixel is (p*out_stride + oh0, q*out_stride + ow0)
// of an [N][OH][OW][Nc] tensor.  This one kernel covers
//   * conv forward           : taps = all (r,s), (dh,dw) = (r,s), in_stride = stride, ih0 = -pad;
//   * dgrad, stride 1        : taps = all (r,s), (dh,dw) = (pad-r, pad-s) over dy, B = W^T [C][R][S][K];
//   * dgrad, stride 2        : per output-parity class (ph,pw) only the taps whose parity matches (no zero taps fed to the
//                              MFMA), out_stride = 2, (oh0,ow0) = (ph,pw); the four classes share ONE launch (ncls) when H, W even;
//   * Linear / plain NT GEMM : one tap, H = W = P = Q = 1.
// Differences from the first-generation kernel (igemm_kernels.h): operands are fetched with BUFFER loads (SRD +
// 32-bit per-lane offset + scalar per-k-step offset; out-of-range lanes return 0, so padding / ragged tiles need no
// branches), tap validity is a per-row bit mask computed once, the k loop carries only scalar state, and everything that
// depends on the tap (row offsets with the validity folded in, the two table entries) is refreshed at tap boundaries only:
// the steady-state iteration is buffer loads + ds_reads + MFMAs + ds_writes + one barrier (the last iteration re-fetches the
// last tile instead of changing shape).
#pragma once
#include "igemm.h"

#define IGEMM_MAX_TAPS 49

// LSTM recurrence fused into the GEMM (EPI = 1 forward, 2 backward; lstm.hip): the workgroup that finishes LAST among the
// k slices of an output tile (atomic ticket — nobody ever waits, so no deadlock is possible) sums the slabs in slice order
// (deterministic, whoever arrives last) and applies the pointwise cell to its tile.
//  EPI 1: logical column nl of tile j is gate (nl >> 4) of hidden unit j*16 + (nl & 15) — the B rows are GATHERED so that a
//         64-column tile holds all four gates of 16 units while W_hh, gx and the gate tape keep torch's [4H] = (gate, unit)
//         layout.  Writes the activated gates, c_t, h_t and the next step's padded A operand.
//  EPI 2: columns are hidden units; dh_t = dy_t + tile, then the cell backward: dgates_t (pre-activation grads), dc_{t-1}.
struct LstmEpi {
  int H, Hp, Gp;
  int* counters;             // [tiles of this timestep], zero before the launch
  const float* gx;           // EPI 1: [B,4H] input projections of this timestep
  const float* c_prev;       // [B,H]  c_{t-1}
  float* gates;              // EPI 1: out [B,4H] activated;  EPI 2: in (const use)
  float* c;                  // EPI 1: out c_t [B,H];          EPI 2: in c_t
  float* h;                  // EPI 1: out h_t [B,H]
  float* hpad;               // EPI 1: out [B,Hp] (columns >= H are never written: zeroed once by the host)
  const float* dy;           // EPI 2: [B,H] or null
  float* dc;                 // EPI 2: in/out [B,H]
  float* dgates;             // EPI 2: out [B,Gp]
};

struct IgemmTapsParams {
  const float* a; const float* b; float* y; const float* bias; const float* add;
  unsigned a_bytes, b_bytes;
  int H, W, C;
  int P, Q;
  int in_stride, ih0, iw0;
  int OH, OW, out_stride, oh0, ow0;
  int M, Nc;
  int b_row_stride;
  int ntaps;
  int add_up2;                         // 1: `add` is [N][OH/2][OW/2][Nc] and is read through a nearest-neighbour x2 upsampling (FPN top-down path)
  int relu, grid_m, grid_n;            // relu: output activation of the final values — 0 none, 1 ReLU, 2 sigmoid
  int ksplit;                          // >1: blockIdx.y picks a contiguous range of k tiles, result goes to slab y + blockIdx.y*slab_stride
  long slab_stride;                    //     (no bias / add / ReLU in that mode; the consumer sums the slabs in a fixed order)
  // Balanced schedule (bal != 0; dense outputs only; grid.x = n_main_tiles*main_ks + n_tail_tiles*tail_slices): on 256 CUs a
  // grid of e.g. 784 equal tiles leaves 16 CUs with 4 workgroups and 240 with 3 — the launch takes 4 units instead of 3.06.
  // Tiles [0, n_main_tiles) (a multiple of 256 workgroups) run whole (or in main_ks k slices), the remaining "tail" tiles are
  // cut into tail_slices short k slices that spread evenly over all CUs; slices write partial slabs that a fixed-order
  // reduce kernel sums (deterministic).  tail rows start at tail_row0 (n_main_tiles is a multiple of grid_n).
  int bal, main_ks, n_main_tiles, tail_slices, tail_row0;
  LstmEpi lstm;                        // EPI != 0 instantiations only
  int ktail;                           // 1: launch the KTAIL instantiation (one tap, C % 4 == 0 but C % 16 != 0; conv2d.hip: taps_kind)
  int epi4;                            // 1: row-major float4 epilogue of the unsplit 64x64 tile (dense outputs; NNL_IGEMM_EPI4)
  int variant;                         // 1: PIPE instantiation of the 64x64 kernel (A/B: tools/bench_conv.py --ab NNL_IGEMM_VARIANT=0,1)
  float* main_out; long main_slab_stride;      // main_ks > 1: slabs [main_ks][tail_row0][Nc]
  float* tail_out; long tail_slab_stride;      // tail_slices > 1: slabs [tail_slices][M - tail_row0][Nc]
  float* bn_part;                              // != null (64x64 tile, dense output): per tile row t and column c the workgroup that
  const float* bn_pivot;                       // produces the FINAL values also writes bn_part[(t*Nc + c)*2 + {0,1}] = sum_rows (y - pivot[c]),
                                               // sum_rows (y - pivot[c])^2 — the BatchNorm batch statistics without re-reading y
  int* tile_counters;                          // != null: the last k slice of a tile to finish (atomic ticket) sums the slabs in
                                               // slice order and writes the output itself — no separate reduce launch.  One int
                                               // per tile, ZERO at rest (the finishing workgroup resets it)
  // Output-parity CLASSES in one launch (ncls > 1; stride-2 dgrad with even OH, OW): workgroup tiles [c*cls_tiles, (c+1)*cls_tiles)
  // belong to class c, which owns taps [cls_tap0[c], cls_tap0[c] + cls_ntaps[c]) of the tables below and writes the output
  // pixels (out_stride*pp + cls_oh0[c], out_stride*qq + cls_ow0[c]); the host orders the classes by decreasing tap count, so
  // the long tiles start first and the short ones fill the tail of the launch (M, P, Q are per class and equal for all).
  int ncls, cls_tiles;
  int cls_tap0[4], cls_ntaps[4], cls_oh0[4], cls_ow0[4];
  // AFFINE taps (tap_affine != 0; dense forward / stride-1 dgrad / linear): tap t = r*tap_S + s of a full R x S raster has
  // (dh, dw) = (tap_dh0 + r*tap_dstep, tap_dw0 + s*tap_dstep), aoff = (dh*W + dw)*C and woff = t*C — everything below is computed
  // from these five scalars and the tables are never read.  (The table walk of the prologue was two dependent global byte loads
  // per tap and row: 4.5 us per workgroup, tools/conv_timing.py.)
  int tap_affine, tap_R, tap_S, tap_dh0, tap_dw0, tap_dstep;
  int tap_aoff[IGEMM_MAX_TAPS];        // (dh*W + dw)*C, elements (may be negative)
  int tap_woff[IGEMM_MAX_TAPS];        // offset of the tap's C weights inside a B row, elements
  signed char tap_dh[IGEMM_MAX_TAPS], tap_dw[IGEMM_MAX_TAPS];
};

typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int POL>
__device__ __forceinline__ f32x4 buf_load4_policy(__amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
  const i32x4 v = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, 0, POL));
  return __builtin_bit_cast(f32x4, v);
}
#define buf_load4_pol(rsrc, voff, pol) buf_load4_policy<pol>(rsrc, voff)

__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
  const i32x4 v = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)soff, 0));
  return __builtin_bit_cast(f32x4, v);
}

__device__ __forceinline__ float nnl_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

// DMA = true (64x64 tile, EPI 0): the operand tiles go global -> LDS directly (`buffer_load_dwordx4 ... lds`, LDS-DMA) instead
// of through VGPRs and ds_write_b128.  An LDS-DMA wave-instruction writes 64 x 16 B lane-linear (wave-uniform base + lane*16),
// so the LDS image is UNPADDED [row][BK]; bank conflicts of the ds_read_b128 fragment reads are avoided by an XOR swizzle of
// the 16-B chunk index with row bits, applied on the SOURCE side (which chunk a lane fetches) and again by the reader.
// Three LDS buffers keep one tile in flight across the barrier (counted vmcnt, raw s_barrier).
#define NNL_LDSP(ptr) ((__attribute__((address_space(3))) void*)(ptr))

// PF = 2 (register staging only): TWO tiles in flight — the buffer loads of tile kt+2 are issued while tile kt is computed and
// land in a second register set; tile kt+1 (requested one iteration earlier) is written to LDS at the end of the iteration.  A
// request then has two iterations (~2.5 us with three co-resident workgroups) to come back instead of one: with 64-channel
// layers almost every A tile contains a first-touch L2 miss (12 % of the lines, 32 lines per wave and tile), and the
// co-resident workgroups of a CU, phase-locked by the shared MFMA pipe, all wait for theirs at the same time.
// KTAIL = true (register staging, one tap: Linear / 1x1 layers whose channel count is a multiple of 4 but not of BK — the tabular
// MLP's 204 / 1000 / 500 columns; round 4): the k loop runs over ceil(C / BK) tiles and, in the last one, the lanes whose 16-B
// chunk starts at a channel >= C fetch from an out-of-range offset (zeros) on BOTH operands, so neither the next row's data nor
// a NaN in it can reach the sum.  Those shapes ran on the first-generation kernel before (23-38 TF/s against 60 here).
template <int BM, int BN, int BK, int WGM, int WGN, bool PIPE = false, int EPI = 0, bool DMA = false, int PF = 1, bool KTAIL = false>
__global__ __launch_bounds__(256, (BM * BN >= 128 * 128 || (PF == 2 && BK == 32)) ? 3 : 4) void igemm_taps_kernel(const IgemmTapsParams p) {
  static_assert(!KTAIL || (!DMA && EPI == 0 && PF == 1), "the k tail is masked in the register-staging loads");
  static_assert(PF == 1 || (PF == 2 && !DMA), "PF = 2 is a register-staging variant");
  static_assert(EPI == 0 || (BM == 64 && BN == 64), "the LSTM epilogue is written for the 64x64 tile");
  static_assert(WGM * WGN == 4 && BK % 8 == 0, "config");
  static_assert(!DMA || (BM == 64 && BN == 64 && EPI == 0 && (BK == 16 || BK == 32)), "LDS-DMA staging: 64x64 tile only");
  constexpr int BKP = DMA ? BK : BK + 4;
  constexpr int NBUF = (DMA && BK == 16) ? 3 : 2;      // BK=32: three 16 KB... 48 KB of LDS would cost a workgroup per CU
  constexpr int KC = BK / 4;
  constexpr int RPP = 256 / KC;
  constexpr int PA = BM / RPP, PB = BN / RPP;
  static_assert(PA >= 1 && PB >= 1 && BM % RPP == 0 && BN % RPP == 0, "tile/thread mapping");
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int TM = WTM / 32, TN = WTN / 32;

  __shared__ __attribute__((aligned(16))) float lds[NBUF][(BM + BN) * BKP];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
#ifdef NNL_TAPS_TIMING
  // debug builds only (tools/gpu/*timing*): four 100 MHz timestamps per workgroup (entry, loop start, loop end, exit) in the upper
  // half of the caller's tile-counter buffer — where does the fixed ~14 us per launch go?
  unsigned long long* const dbg_t = (p.tile_counters != nullptr && blockIdx.x < 3276 && blockIdx.y == 0)
                                        ? reinterpret_cast<unsigned long long*>(p.tile_counters + 32768) + (long)blockIdx.x * 5 : nullptr;
  if (dbg_t && tid == 0) {
    dbg_t[0] = wall_clock64();
    // where it runs: HW_ID (reg 4: cu_id bits 11:8, sh 12, se 15:13) and XCC_ID (reg 20, bits 3:0)
    dbg_t[4] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
  }
#define NNL_TSTAMP(i) do { if (dbg_t && tid == 0) dbg_t[i] = wall_clock64(); } while (0)
#else
#define NNL_TSTAMP(i) do { } while (0)
#endif
  int logical, kslice = 0, nslices = 1, row0 = 0, cls = 0;
  bool in_tail = false;
  float* yout = p.y;
  if (p.bal) {
    const int nmb = p.n_main_tiles * p.main_ks;
    if ((int)blockIdx.x < nmb) {
      const int u = nnl_xcd_remap(blockIdx.x, nmb);
      logical = u / p.main_ks;
      kslice = u - logical * p.main_ks;
      nslices = p.main_ks;
      if (nslices > 1) yout = p.main_out + (long)kslice * p.main_slab_stride;
    } else {
      const int tb = (int)blockIdx.x - nmb;
      const int t = tb / p.tail_slices;
      kslice = tb - t * p.tail_slices;
      logical = p.n_main_tiles + t;
      nslices = p.tail_slices;
      if (nslices > 1) { yout = p.tail_out + (long)kslice * p.tail_slab_stride; row0 = p.tail_row0; in_tail = true; }
    }
  } else if (p.ncls > 1) {
    // classes run one after the other in launch order (longest first); the XCD remap is applied INSIDE a class, so every XCD
    // gets its share of each class (a remap of the whole grid would hand the long classes to XCDs 0-1 and the short to 6-7)
    cls = (int)blockIdx.x / p.cls_tiles;
    logical = nnl_xcd_remap((int)blockIdx.x - cls * p.cls_tiles, p.cls_tiles);
  } else {
    logical = nnl_xcd_remap(blockIdx.x, gridDim.x);
    if (p.ksplit > 1) { kslice = (int)blockIdx.y; nslices = p.ksplit; yout = p.y + (long)blockIdx.y * p.slab_stride; }
  }
  int tap0 = 0, ntaps = p.ntaps, oh0 = p.oh0, ow0 = p.ow0;
  if (p.ncls > 1) {
    tap0 = p.cls_tap0[cls]; ntaps = p.cls_ntaps[cls]; oh0 = p.cls_oh0[cls]; ow0 = p.cls_ow0[cls];
  }
  const bool partial = nslices > 1;            // partial sums: no bias / add / ReLU (the reduce kernel applies them)
  const int tile_m = logical / p.grid_n, tile_n = logical - tile_m * p.grid_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kc = tid % KC, lrow = tid / KC;
  // staging map: pass i of this thread covers tile row srow(i) and fetches 16-B chunk schunk(i) of that row
  auto swz = [](int row) { return BK == 16 ? (row >> 2) & 3 : (row >> 1) & 7; };     // DMA image: chunk' = chunk ^ swz(row)
  auto srow = [&](int i) { return DMA ? (wave * PA + i) * (64 / KC) + lane / KC : lrow + i * RPP; };
  auto schunk = [&](int i) { return DMA ? (lane % KC) ^ swz(srow(i)) : kc; };

  const __amdgpu_buffer_rsrc_t ra_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.b), 0, (int)p.b_bytes, 0x00020000);

  // ---- per-thread rows: byte offset of the base pixel (+ this thread's 16-B chunk) and the tap validity mask ----
  int a_off[PA];
  unsigned long long a_mask[PA];
  const int PQ = p.P * p.Q;
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int m = m0 + srow(i);
    const bool valid = m < p.M;
    const int mm = valid ? m : 0;
    const int n = mm / PQ;
    const int rem = mm - n * PQ;
    const int pp = rem / p.Q;
    const int qq = rem - pp * p.Q;
    const int h0 = pp * p.in_stride + p.ih0, w0 = qq * p.in_stride + p.iw0;
    a_off[i] = (((n * p.H + h0) * p.W + w0) * p.C + schunk(i) * 4) * 4;
    unsigned long long mask = 0;
    if (valid) {
      if (p.tap_affine) {                           // R + S range tests, no table: row r is valid for every column bit of `cols`
        unsigned cols = 0;
        for (int ss = 0; ss < p.tap_S; ++ss)
          if ((unsigned)(w0 + p.tap_dw0 + ss * p.tap_dstep) < (unsigned)p.W) cols |= 1u << ss;
        for (int r = 0; r < p.tap_R; ++r)
          if ((unsigned)(h0 + p.tap_dh0 + r * p.tap_dstep) < (unsigned)p.H) mask |= (unsigned long long)cols << (r * p.tap_S);
      } else {
        for (int t = 0; t < ntaps; ++t) {
          const int h = h0 + p.tap_dh[tap0 + t], w = w0 + p.tap_dw[tap0 + t];
          if ((unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W) mask |= 1ull << t;
        }
      }
    }
    a_mask[i] = mask;
  }
  unsigned b_off[PB];
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    int nr = n0 + srow(i);
    bool okr = nr < p.Nc;
    if constexpr (EPI == 1) {                      // gate-gathered rows: (gate, unit) -> W_hh row gate*H + unit
      const int nl = lrow + i * RPP, u = tile_n * 16 + (nl & 15);
      okr = u < p.lstm.H;
      nr = (nl >> 4) * p.lstm.H + u;
    }
    b_off[i] = okr ? (unsigned)(nr * p.b_row_stride + schunk(i) * 4) * 4u : 0xFFFFFFFFu;
  }

  f32x4 ra[PA], rb[PB];
  // Per-TAP state, refreshed only when the k loop crosses into another tap (every C/BK iterations): the per-row byte offsets of
  // the tap's pixel (or an out-of-range offset for a padded / ragged row, which the buffer load turns into zeros) and the two
  // scalar table entries.  The steady-state iteration is then just buffer loads at (per-lane offset, scalar channel offset).
  unsigned a_voff[PA];
  unsigned b_tap = 0;
  auto set_tap = [&](int t) {
    int a_tap;                                           // may be negative; the sum with a valid row's base is not
    if (p.tap_affine) {                                  // wave-uniform scalar arithmetic, once per tap
      const int r = t / p.tap_S, ss = t - r * p.tap_S;
      a_tap = ((p.tap_dh0 + r * p.tap_dstep) * p.W + p.tap_dw0 + ss * p.tap_dstep) * p.C * 4;
      b_tap = (unsigned)(t * p.C) * 4u;
    } else {
      a_tap = p.tap_aoff[tap0 + t] * 4;
      b_tap = (unsigned)p.tap_woff[tap0 + t] * 4u;
    }
#pragma unroll
    for (int i = 0; i < PA; ++i) a_voff[i] = ((a_mask[i] >> t) & 1ull) ? (unsigned)(a_off[i] + a_tap) : 0xFFFFFFFFu;
  };
  auto load_tile = [&](int c0, int buf) {
    if constexpr (DMA) {
      // piece (wave, i) = 64 consecutive 16-B slots of the A image (rows (wave*PA+i)*(64/KC) ...) and the same of the B image
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        float* dst = lds[buf] + (wave * PA + i) * 256;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra_src, NNL_LDSP(dst), 16, (int)a_voff[i], c0 * 4, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb_src, NNL_LDSP(dst + BM * BK), 16, (int)b_off[i], (int)b_tap + c0 * 4, 0, 0);
      }
    } else {
      const bool okc = !KTAIL || c0 + kc * 4 < p.C;          // (KTAIL: false only in the last k tile, for the chunks beyond C)
#pragma unroll
      for (int i = 0; i < PA; ++i) ra[i] = buf_load4(ra_src, okc ? a_voff[i] : 0xFFFFFFFFu, (unsigned)c0 * 4u);
#pragma unroll
      for (int i = 0; i < PB; ++i) rb[i] = buf_load4(rb_src, okc ? b_off[i] : 0xFFFFFFFFu, b_tap + (unsigned)c0 * 4u);
    }
  };
  auto store_tile = [&](int buf) {
    if constexpr (DMA) return;
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
  // When a wave owns a single 32x32 accumulator (64x64 block tile) its MFMAs would form one dependent chain; a second
  // accumulator for the odd 8-wide k groups makes consecutive MFMAs independent (summed once in the epilogue).
  constexpr bool kTwoAcc = (TM * TN == 1);
  f32x16 acc2;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc2[e] = 0.f;
  // DMA image: fragment chunk (kk*2 + k half) of row r lives at chunk' = chunk ^ swz(r): per-lane float offsets per 8-wide k group
  int fa_off[BK / 8], fb_off[BK / 8];
  if constexpr (DMA) {
    const int ra_r = wm * 32 + (lane & 31), rb_r = wn * 32 + (lane & 31);
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      fa_off[kk] = ra_r * BK + (((kk * 2 + (lane >> 5)) ^ swz(ra_r)) * 4);
      fb_off[kk] = BM * BK + rb_r * BK + (((kk * 2 + (lane >> 5)) ^ swz(rb_r)) * 4);
    }
  }
  auto compute = [&](int buf) {
    if constexpr (DMA) {
      const float* base = lds[buf];
      f32x4 fa[BK / 8], fb[BK / 8];
#pragma unroll
      for (int kk = 0; kk < BK / 8; ++kk) {
        fa[kk] = *reinterpret_cast<const f32x4*>(base + fa_off[kk]);
        fb[kk] = *reinterpret_cast<const f32x4*>(base + fb_off[kk]);
      }
#pragma unroll
      for (int kk = 0; kk < BK / 8; kk += 2) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk][t], fb[kk][t], acc[0][0], 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk + 1][t], fb[kk + 1][t], acc2, 0, 0, 0);
        }
      }
      return;
    }
    const float* As = lds[buf] + wm * WTM * BKP + frag_off;
    const float* Bs = lds[buf] + BM * BKP + wn * WTN * BKP + frag_off;
    if constexpr (kTwoAcc) {
      if constexpr (!PIPE) {
#pragma unroll
        for (int kk = 0; kk < BK / 8; kk += 2) {
          const f32x4 a0 = *reinterpret_cast<const f32x4*>(As + kk * 8), b0 = *reinterpret_cast<const f32x4*>(Bs + kk * 8);
          const f32x4 a1 = *reinterpret_cast<const f32x4*>(As + kk * 8 + 8), b1 = *reinterpret_cast<const f32x4*>(Bs + kk * 8 + 8);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b0[t], acc[0][0], 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b1[t], acc2, 0, 0, 0);
          }
        }
        return;
      }
      // software-pipelined over pairs of 8-wide k groups: the four ds_read_b128 of pair p+1 are issued before the eight
      // MFMAs of pair p, so their LDS latency hides behind 512 MFMA cycles instead of 64
      constexpr int NP = BK / 16;
      f32x4 fa[2][2], fb[2][2];
      fa[0][0] = *reinterpret_cast<const f32x4*>(As); fb[0][0] = *reinterpret_cast<const f32x4*>(Bs);
      fa[0][1] = *reinterpret_cast<const f32x4*>(As + 8); fb[0][1] = *reinterpret_cast<const f32x4*>(Bs + 8);
#pragma unroll
      for (int pp = 0; pp < NP; ++pp) {
        const int cur_s = pp & 1, nxt_s = cur_s ^ 1;
        if (pp + 1 < NP) {
          fa[nxt_s][0] = *reinterpret_cast<const f32x4*>(As + (pp + 1) * 16); fb[nxt_s][0] = *reinterpret_cast<const f32x4*>(Bs + (pp + 1) * 16);
          fa[nxt_s][1] = *reinterpret_cast<const f32x4*>(As + (pp + 1) * 16 + 8); fb[nxt_s][1] = *reinterpret_cast<const f32x4*>(Bs + (pp + 1) * 16 + 8);
          __builtin_amdgcn_sched_barrier(0);                 // keep the compiler from sinking them back behind the MFMAs
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur_s][0][t], fb[cur_s][0][t], acc[0][0], 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur_s][1][t], fb[cur_s][1][t], acc2, 0, 0, 0);
        }
      }
      return;
    }
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

  // ---- k loop: scalar state (tap, channel offset); the body has no branches ----
  const int csteps = KTAIL ? (p.C + BK - 1) / BK : p.C / BK;
  const int nk_all = ntaps * csteps;
  int kt0 = 0, nk = nk_all;
  if (partial) {                               // split-K: this workgroup reduces k tiles [kt0, kt0 + nk)
    const int per = (nk_all + nslices - 1) / nslices;
    kt0 = kslice * per;
    nk = min(per, nk_all - kt0);
    if (nk < 0) nk = 0;
  }
  int t_nx = kt0 / csteps, c_nx = (kt0 - t_nx * csteps) * BK;       // (tap, c0) of the NEXT tile to fetch
  auto advance = [&]() {                       // (t_nx, c_nx) -> the next k tile; refreshes the per-tap state at a tap boundary
    c_nx += BK;
    if (c_nx >= p.C) { c_nx = 0; ++t_nx; set_tap(t_nx); }        // wave-uniform branch, once per tap
  };
  if constexpr (DMA && NBUF == 3) {
    // three LDS buffers, tile kt+2 issued while tile kt is computed; before the barrier that ends iteration kt every wave
    // waits until ITS pieces of tile kt+1 have landed (all but the 2*PA youngest DMAs = those of tile kt+2), so after the
    // barrier tile kt+1 is complete; buffer (kt+2)%3 was last read in iteration kt-1, which every wave has left.
    static_assert(PA == 1, "vmcnt immediate below");
    if (nk > 0) {
      set_tap(t_nx);
      load_tile(c_nx, 0);
      if (nk > 1) advance();
      load_tile(c_nx, 1);                      // nk == 1: re-fetches tile 0 (never read)
    }
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 2 < nk) advance();              // the last two iterations re-fetch the last tile (never read; keeps the body uniform)
      int nx2 = cur + 2; if (nx2 >= 3) nx2 -= 3;
      load_tile(c_nx, nx2);
      __builtin_amdgcn_sched_barrier(0);
      compute(cur);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      cur = cur == 2 ? 0 : cur + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // drain the spare fetches before the epilogue reuses the LDS
    __syncthreads();
  } else if constexpr (DMA) {
    // two LDS buffers: tile kt+1 streams in while tile kt is computed; drained before the barrier
    if (nk > 0) {
      set_tap(t_nx);
      load_tile(c_nx, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) advance();
      load_tile(c_nx, cur ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      compute(cur);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      cur ^= 1;
    }
    __syncthreads();
  } else if constexpr (PF == 2) {
    f32x4 ra2[PA], rb2[PB];
    auto load_tile2 = [&](int c0) {
#pragma unroll
      for (int i = 0; i < PA; ++i) ra2[i] = buf_load4(ra_src, a_voff[i], (unsigned)c0 * 4u);
#pragma unroll
      for (int i = 0; i < PB; ++i) rb2[i] = buf_load4(rb_src, b_off[i], b_tap + (unsigned)c0 * 4u);
    };
    auto store_tile2 = [&](int buf) {
      float* As = lds[buf];
      float* Bs = As + BM * BKP;
#pragma unroll
      for (int i = 0; i < PA; ++i) *reinterpret_cast<f32x4*>(As + (lrow + i * RPP) * BKP + kc * 4) = ra2[i];
#pragma unroll
      for (int i = 0; i < PB; ++i) *reinterpret_cast<f32x4*>(Bs + (lrow + i * RPP) * BKP + kc * 4) = rb2[i];
    };
    if (nk > 0) {
      set_tap(t_nx);
      load_tile(c_nx, 0);
      store_tile(0);                             // tile 0 -> LDS[0]
      if (nk > 1) advance();
      load_tile(c_nx, 0);                        // tile 1 -> set A, in flight (nk == 1: tile 0 again, never used)
    }
    __syncthreads();
    NNL_TSTAMP(1);
    // pairs of k tiles, then the odd last one outside the loop: with an exit in the MIDDLE of the body (round 4) the compiler kept the
    // accumulators of the two halves in different registers and copied all 32 of them (behind an MFMA drain) once per pair
    int cur = 0, kt = 0;
    for (; kt + 1 < nk; kt += 2) {
      if (kt + 2 < nk) advance();
      load_tile2(c_nx);                          // tile kt+2 -> set B
      __builtin_amdgcn_sched_barrier(0);
      compute(cur);
      __builtin_amdgcn_sched_barrier(0);
      store_tile(cur ^ 1);                       // set A = tile kt+1 (requested an iteration ago)
      __syncthreads();
      cur ^= 1;
      if (kt + 3 < nk) advance();
      load_tile(c_nx, 0);                        // tile kt+3 -> set A (past the end: the last tile again, never used)
      __builtin_amdgcn_sched_barrier(0);
      compute(cur);
      __builtin_amdgcn_sched_barrier(0);
      store_tile2(cur ^ 1);                      // set B = tile kt+2
      __syncthreads();
      cur ^= 1;
    }
    if (kt < nk) {                               // odd count: the last tile sits in LDS[cur]
      compute(cur);
      __syncthreads();
    }
  } else {
  if (nk > 0) {
    set_tap(t_nx);
    load_tile(c_nx, 0);
    store_tile(0);
  }
  __syncthreads();
  NNL_TSTAMP(1);
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // advance to tile kt+1 (the final iteration re-fetches the last tile instead: harmless, keeps the body uniform)
    if (kt + 1 < nk) advance();
    load_tile(c_nx, 0);                        // buffer loads of tile kt+1 go out FIRST ...
    __builtin_amdgcn_sched_barrier(0);         // ... (keep the compiler from sinking them behind the MFMAs to save VGPRs)
    compute(cur);                              // 32 MFMAs per wave cover their latency
    __builtin_amdgcn_sched_barrier(0);
    store_tile(cur ^ 1);                       // vmcnt wait + ds_write only after the MFMAs are issued
    __syncthreads();
    cur ^= 1;
  }
  }

  NNL_TSTAMP(2);
  // ---- epilogue ----
  if constexpr (kTwoAcc) {
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[0][0][e] += acc2[e];
  }
  if constexpr (EPI != 0) {
    // ---- fused LSTM cell ----------------------------------------------------------------------------------------------
    constexpr int LDT = 68;
    float* tl = &lds[0][0];                          // 64 x 68 floats: the k loop is over (it ended with a barrier)
    __shared__ int ticket;
    const int Nlog = p.grid_n * 64;
    const int cl = wn * 32 + (lane & 31);
    if (nslices > 1) {
      // Cross-workgroup hand-over WITHOUT cache-wide fences: a device-scope __threadfence() writes back and invalidates the
      // whole L2 of the XCD (measured: 4x slower steps, W_hh loses its L2 residency).  Instead the slab is written with
      // agent-scope (write-through) atomic stores, every wave drains its stores (vmcnt 0) before the workgroup takes its
      // ticket, and the finishing workgroup reads the slabs with agent-scope loads that bypass its own L2.
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)((long)nslices * p.slab_stride * 4), 0x00020000);
      constexpr int kSc1 = 1 << 4;                   // gfx940+ cache policy bit: device (agent) scope — write-through / L2 bypass
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int rl = wm * 32 + (e & 3) + 8 * (e >> 2) + (lane >> 5) * 4;
        const long off = (long)kslice * p.slab_stride + (long)(m0 + rl) * Nlog + n0 + cl;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[0][0][e]), rs, (m0 + rl < p.M) ? (int)(off * 4) : -1, 0, kSc1);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) ticket = __hip_atomic_fetch_add(&p.lstm.counters[logical], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      if (ticket != nslices - 1) return;             // someone else finishes this tile
      // sum the slices in slice order (deterministic whoever is last): 4 float4 per thread, all loads in flight together
      for (int k4 = 0; k4 < 4; ++k4) {
        const int idx4 = tid + k4 * 256, rl = idx4 >> 4, c4 = (idx4 & 15) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (m0 + rl < p.M) {
          for (int sl0 = 0; sl0 < nslices; sl0 += 8) {
            f32x4 part[8];
#pragma unroll
            for (int sl = 0; sl < 8; ++sl)
              if (sl0 + sl < nslices) part[sl] = buf_load4_pol(rs, (unsigned)(((long)(sl0 + sl) * p.slab_stride + (long)(m0 + rl) * Nlog + n0 + c4) * 4), kSc1);
#pragma unroll
            for (int sl = 0; sl < 8; ++sl)
              if (sl0 + sl < nslices) v += part[sl];
          }
        }
        *reinterpret_cast<f32x4*>(tl + rl * LDT + c4) = v;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int rl = wm * 32 + (e & 3) + 8 * (e >> 2) + (lane >> 5) * 4;
        tl[rl * LDT + cl] = acc[0][0][e];
      }
    }
    __syncthreads();
    const LstmEpi& L = p.lstm;
    const int H = L.H;
    if constexpr (EPI == 1) {
      for (int idx = tid; idx < 64 * 16; idx += 256) {
        const int rl = idx >> 4, j = idx & 15;
        const int b = m0 + rl, u = tile_n * 16 + j;
        if (b >= p.M || u >= H) continue;
        const long g0 = (long)b * 4 * H + u, bu = (long)b * H + u;
        const float gi = nnl_sigmoid(L.gx[g0] + tl[rl * LDT + j]);
        const float gf = nnl_sigmoid(L.gx[g0 + H] + tl[rl * LDT + 16 + j]);
        const float gg = tanhf(L.gx[g0 + 2L * H] + tl[rl * LDT + 32 + j]);
        const float go = nnl_sigmoid(L.gx[g0 + 3L * H] + tl[rl * LDT + 48 + j]);
        const float cn = gf * L.c_prev[bu] + gi * gg;
        const float hn = go * tanhf(cn);
        L.gates[g0] = gi; L.gates[g0 + H] = gf; L.gates[g0 + 2L * H] = gg; L.gates[g0 + 3L * H] = go;
        L.c[bu] = cn;
        L.h[bu] = hn;
        L.hpad[(long)b * L.Hp + u] = hn;
      }
    } else {
      for (int idx = tid; idx < 64 * 64; idx += 256) {
        const int rl = idx >> 6, j = idx & 63;
        const int b = m0 + rl, u = n0 + j;
        if (b >= p.M || u >= H) continue;
        const long g0 = (long)b * 4 * H + u, bu = (long)b * H + u;
        const float gi = L.gates[g0], gf = L.gates[g0 + H], gg = L.gates[g0 + 2L * H], go = L.gates[g0 + 3L * H];
        const float dh = (L.dy ? L.dy[bu] : 0.f) + tl[rl * LDT + j];
        const float tc = tanhf(L.c[bu]);
        const float dcn = L.dc[bu] + dh * go * (1.f - tc * tc);
        float* dg = L.dgates + (long)b * L.Gp;
        dg[u] = dcn * gg * (gi * (1.f - gi));
        dg[H + u] = dcn * L.c_prev[bu] * (gf * (1.f - gf));
        dg[2 * H + u] = dcn * gi * (1.f - gg * gg);
        dg[3 * H + u] = dh * tc * (go * (1.f - go));
        L.dc[bu] = dcn * gf;
      }
    }
    return;
  }
  const int col_l = lane & 31, row_h = (lane >> 5) * 4;
  const bool dense_out = (p.out_stride == 1) && (p.OH == p.P) && (p.OW == p.Q) && (oh0 == 0) && (ow0 == 0);
  if (EPI == 0 && BM == 64 && BN == 64 && p.bal && partial && p.tile_counters != nullptr) {
    // ---- in-kernel fix-up of a split tile (same hand-over as the fused LSTM step: sc1 stores, drain, ticket, sc1 loads) ----
    __shared__ int ticket;
    float* const base = in_tail ? p.tail_out : p.main_out;
    const long sstride = in_tail ? p.tail_slab_stride : p.main_slab_stride;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)((long)nslices * sstride * 4), 0x00020000);
    constexpr int kSc1 = 1 << 4;
    const int cl = n0 + wn * 32 + col_l;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + row_h;
      const long off = (long)kslice * sstride + (long)(row - row0) * p.Nc + cl;
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[0][0][e]), rs, (row < p.M && cl < p.Nc) ? (int)(off * 4) : -1, 0, kSc1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) ticket = __hip_atomic_fetch_add(&p.tile_counters[logical], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (ticket != nslices - 1) return;
    if (tid == 0) __hip_atomic_store(&p.tile_counters[logical], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // zero at rest
    float fs1[4] = {0.f, 0.f, 0.f, 0.f}, fs2[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k4 = 0; k4 < 4; ++k4) {
      const int idx4 = tid + k4 * 256, rl = idx4 >> 4, c4 = n0 + (idx4 & 15) * 4;
      const int row = m0 + rl;
      if (row >= p.M || c4 >= p.Nc) continue;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      for (int sl0 = 0; sl0 < nslices; sl0 += 8) {     // eight slab loads in flight per trip, added in slice order (the loop that
        f32x4 part[8];                                 // followed the first eight used to wait for every single load)
#pragma unroll
        for (int sl = 0; sl < 8; ++sl)
          if (sl0 + sl < nslices) part[sl] = buf_load4_pol(rs, (unsigned)(((long)(sl0 + sl) * sstride + (long)(row - row0) * p.Nc + c4) * 4), kSc1);
#pragma unroll
        for (int sl = 0; sl < 8; ++sl)
          if (sl0 + sl < nslices) v += part[sl];
      }
      const long o = (long)row * p.Nc + c4;
      if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + c4);
      if (p.add) v += *reinterpret_cast<const f32x4*>(p.add + o);
      if (p.relu == 1) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
      else if (p.relu == 2) { v[0] = nnl_sigmoid(v[0]); v[1] = nnl_sigmoid(v[1]); v[2] = nnl_sigmoid(v[2]); v[3] = nnl_sigmoid(v[3]); }
      *reinterpret_cast<f32x4*>(p.y + o) = v;
      if (p.bn_part) {
        const f32x4 pv = *reinterpret_cast<const f32x4*>(p.bn_pivot + c4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[e] - pv[e]; fs1[e] += d; fs2[e] += d * d; }
      }
    }
    if (p.bn_part) {                              // thread t owns columns (t & 15)*4..+3 of rows t>>4, +16, +32, +48
      __syncthreads();
      float* red = &lds[0][0];                      // [16 row lanes][64 cols][2]
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
  if constexpr (EPI == 0 && BM == 64 && BN == 64) {
    if (p.epi4 && dense_out && !p.add_up2 && !partial && p.Nc % 4 == 0) {
      // ---- row-major float4 epilogue: the accumulator tile goes through LDS (free: the k loop ended with a barrier) and every
      // thread finishes four float4 pieces of output rows — 16 lanes cover a 256-B row segment per store instead of 32 lanes
      // writing 128 B of two rows each with 4-byte stores (the 1x1 convolutions of the ResNet-50 body write 4x what they read:
      // their time IS this epilogue), and the addend / bias / BatchNorm pivot come in as float4 too
      constexpr int LDT = 68;
      float* tl = &lds[0][0];                              // 64 x 68 floats
#pragma unroll
      for (int e = 0; e < 16; ++e) tl[(wm * 32 + (e & 3) + 8 * (e >> 2) + row_h) * LDT + wn * 32 + col_l] = acc[0][0][e];
      __syncthreads();
      float fs1[4] = {0.f, 0.f, 0.f, 0.f}, fs2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) {
        const int idx4 = tid + k4 * 256, rl = idx4 >> 4, c4 = n0 + (idx4 & 15) * 4;
        const int row = m0 + rl;
        if (row >= p.M || c4 >= p.Nc) continue;
        f32x4 v = *reinterpret_cast<const f32x4*>(tl + rl * LDT + (idx4 & 15) * 4);
        const long o = (long)row * p.Nc + c4;
        if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + c4);
        if (p.add) v += *reinterpret_cast<const f32x4*>(p.add + o);
        if (p.relu == 1) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
        else if (p.relu == 2) { v[0] = nnl_sigmoid(v[0]); v[1] = nnl_sigmoid(v[1]); v[2] = nnl_sigmoid(v[2]); v[3] = nnl_sigmoid(v[3]); }
        *reinterpret_cast<f32x4*>(p.y + o) = v;
        if (p.bn_part) {
          const f32x4 pv = *reinterpret_cast<const f32x4*>(p.bn_pivot + c4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float d = v[e] - pv[e]; fs1[e] += d; fs2[e] += d * d; }
        }
      }
      if (p.bn_part) {                              // thread t owns columns (t & 15)*4..+3 of rows t>>4, +16, +32, +48
        __syncthreads();
        float* red = &lds[0][0];                      // [16 row lanes][64 cols][2]
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
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn * WTN + j * 32 + col_l;
    const bool cok = col < p.Nc;
    const float bv = (p.bias != nullptr && cok && !partial) ? p.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + row_h;
        if (cok && row < p.M) {
          long pix = row, apix = row;
          if (!dense_out || p.add_up2) {
            const int n = row / PQ;
            const int rem = row - n * PQ;
            const int pp = rem / p.Q;
            const int qq = rem - pp * p.Q;
            const int oh = pp * p.out_stride + oh0, ow = qq * p.out_stride + ow0;
            pix = ((long)n * p.OH + oh) * p.OW + ow;
            apix = p.add_up2 ? ((long)n * (p.OH >> 1) + (oh >> 1)) * (p.OW >> 1) + (ow >> 1) : pix;
          }
          float v = acc[i][j][e] + bv;
          if (!partial) {
            if (p.add) v += p.add[apix * p.Nc + col];
            if (p.relu == 1) v = fmaxf(v, 0.f);
            else if (p.relu == 2) v = nnl_sigmoid(v);                 // ClassificationModel's output activation (retinanet.py:286)
          }
          yout[(pix - row0) * p.Nc + col] = v;
        }
      }
    }
  }
  if (EPI == 0 && BM == 64 && BN == 64 && p.bn_part != nullptr && !partial) {
    // BatchNorm statistics of this tile's 64 rows (lane l holds 16 rows of column l & 31; the two lane halves and the two
    // row waves are combined through LDS, which is free: the k loop ended with a barrier)
    const int col = n0 + wn * 32 + col_l;
    const bool cok = col < p.Nc;
    const float piv = cok ? p.bn_pivot[col] : 0.f, bv = (p.bias != nullptr && cok) ? p.bias[col] : 0.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + row_h;
      if (row < p.M) {
        float v = acc[0][0][e] + bv;
        if (p.relu == 1) v = fmaxf(v, 0.f);
        else if (p.relu == 2) v = nnl_sigmoid(v);
        const float d = v - piv;
        s1 += d; s2 += d * d;
      }
    }
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    float* red = &lds[0][0];                        // [2 row waves][64 cols][2]
    if (lane < 32) { red[((wm * 64) + wn * 32 + lane) * 2] = s1; red[((wm * 64) + wn * 32 + lane) * 2 + 1] = s2; }
    __syncthreads();
    if (tid < 64 && n0 + tid < p.Nc) {
      p.bn_part[((long)tile_m * p.Nc + n0 + tid) * 2 + 0] = red[tid * 2] + red[(64 + tid) * 2];
      p.bn_part[((long)tile_m * p.Nc + n0 + tid) * 2 + 1] = red[tid * 2 + 1] + red[(64 + tid) * 2 + 1];
    }
  }
#ifdef NNL_TAPS_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  NNL_TSTAMP(3);
#endif
}

// One LSTM timestep (defined in conv2d.hip, where the kernel templates are instantiated): epi 1 = forward (q.M = batch,
// q.C = Hp, grid_n = ceil(H/16) gate-gathered tiles), epi 2 = backward (q.Nc = H, q.C = Gp).  q.ksplit k slices per tile
// write slabs q.y[ksplit][M][grid_n*64]; q.lstm.counters must be zero.
int nnl_internal_lstm_step(IgemmTapsParams q, int epi, hipStream_t s);
