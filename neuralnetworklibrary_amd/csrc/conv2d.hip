// K1 — conv2d forward / dgrad / wgrad as implicit GEMMs on the exact-fp32 MFMA (gfx950).
// Replaces the cuDNN convolutions behind the reference's ResNet blocks, FPN and RetinaNet heads
// (Applications/VisionModels/retinanet.py:26-28,43-59,77-97,126-148,187-217,260-295,304,344-348).
// Layout: activations NHWC, filters KRSC (see include/nnl.h).  MFMA-bound: 2*N*P*Q*K*R*S*C flops per pass.
#include "igemm_kernels.h"
#include "igemm_taps.h"
#include "wino.h"
#include "igemm_wgrad.h"
#include "igemm_wgrad2d.h"

#ifdef NNL_TAPS_TIMING
static void* g_wgrad_dbg = nullptr;
extern "C" void* nnl_debug_wgrad_stamps(void) { return g_wgrad_dbg; }       // timing builds only
#endif

namespace {

int check_geom(const nnl_conv_geom_t* g, const char* who) {
  if (!g) return nnl_set_error(NNL_ERR_INVALID_ARG, "%s: null geometry", who);
  if (g->N <= 0 || g->H <= 0 || g->W <= 0 || g->C <= 0 || g->K <= 0 || g->R <= 0 || g->S <= 0 || g->stride <= 0 ||
      g->pad < 0)
    return nnl_set_error(NNL_ERR_INVALID_ARG, "%s: non-positive dimension", who);
  const int P = (g->H + 2 * g->pad - g->R) / g->stride + 1, Q = (g->W + 2 * g->pad - g->S) / g->stride + 1;
  if (P != g->P || Q != g->Q || P <= 0 || Q <= 0)
    return nnl_set_error(NNL_ERR_INVALID_ARG, "%s: P,Q=(%d,%d) do not match the geometry (%d,%d)", who, g->P, g->Q, P, Q);
  if (g->C % 4 != 0)
    return nnl_set_error(NNL_ERR_UNSUPPORTED, "%s: C=%d must be a multiple of 4 (pad the channels; ops.py does)", who, g->C);
  if ((long)g->N * g->H * g->W * g->C >= (1L << 31) || (long)g->N * g->P * g->Q * g->K >= (1L << 31) ||
      (long)g->N * g->P * g->Q >= (1L << 30))
    return nnl_set_error(NNL_ERR_UNSUPPORTED, "%s: tensor too large for 32-bit row indexing", who);
  return NNL_OK;
}

template <int BM, int BN, int WGM, int WGN, int MODE>
int launch_rowk(IgemmRowkParams p, hipStream_t s) {
  p.grid_m = (int)nnl_cdiv(p.M, BM);
  p.grid_n = (int)nnl_cdiv(p.Nc, BN);
  hipLaunchKernelGGL((igemm_rowk_kernel<BM, BN, 16, WGM, WGN, MODE>), dim3(p.grid_m * p.grid_n), dim3(256), 0, s, p);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

template <int MODE>
int dispatch_rowk(const IgemmRowkParams& p, hipStream_t s) {
  // tuning hook (tools/bench_conv.py): NNL_IGEMM_TILE=0..3 forces 128x128 / 128x64 / 64x128 / 64x64
  const int forced = NNL_AB_INT("NNL_IGEMM_TILE", -1);
  switch (forced) {
    case 0: return launch_rowk<128, 128, 2, 2, MODE>(p, s);
    case 1: return launch_rowk<128, 64, 2, 2, MODE>(p, s);
    case 2: return launch_rowk<64, 128, 2, 2, MODE>(p, s);
    case 3: return launch_rowk<64, 64, 2, 2, MODE>(p, s);
    default: break;
  }
  const long b128 = nnl_cdiv(p.M, 128);
  if (p.Nc > 64) {
    if (b128 * nnl_cdiv(p.Nc, 128) >= 400) return launch_rowk<128, 128, 2, 2, MODE>(p, s);
    if (nnl_cdiv(p.M, 64) * nnl_cdiv(p.Nc, 128) >= 400) return launch_rowk<64, 128, 2, 2, MODE>(p, s);
    return launch_rowk<64, 64, 2, 2, MODE>(p, s);
  }
  if (b128 >= 400) return launch_rowk<128, 64, 2, 2, MODE>(p, s);
  return launch_rowk<64, 64, 2, 2, MODE>(p, s);
}


// LDS-DMA staging of the 64x64 tile (igemm_taps.h): an experiment kept behind NNL_IGEMM_DMA (0 off = default, 1 BK=16 launches,
// 2 BK=32 launches, 3 both).  Measured per ResNet-34 layer (bench_conv.py --ab NNL_IGEMM_DMA=0,3): +3.5 % on the 56x56 / C=64
// stage, -1 % on 28x28 / C=128, -7 % with BK=32 (two buffers); inside the full training step the gain on the C=64 stage
// does not show (15.44 ms/step either way), so register staging stays the shipped path.
static bool taps_dma(int bk, const IgemmTapsParams&) {
  const int m = NNL_ENV_INT("NNL_IGEMM_DMA", 0);
  return bk == 16 ? (m & 1) != 0 : (m & 2) != 0;
}

// Two tiles in flight (PF = 2, igemm_taps.h) for the 64x64 tile: NNL_IGEMM_PF2 bit 0 = BK 16 launches (default ON), bit 1 = BK 32
// launches (default off).  Measured per ResNet-34 layer at 64 images (bench_conv.py --ab NNL_IGEMM_PF2=0,1,2,3,
// profiles/r3_pf2_bs64.log): BK 16 (the 28x28 / C = 128 stage) fwd 107.0 -> 111.7 TF/s, dgrad 112.7 -> 116.7; BK 32 needs 148
// VGPRs (three workgroups per CU instead of four) and LOSES 3-8 % on every stride-1 layer (l1 107 -> 104, l4 117 -> 109).
static bool taps_pf2(int bk, const IgemmTapsParams&) {
  const int m = NNL_AB_INT("NNL_IGEMM_PF2", 1);
  return bk == 16 ? (m & 1) != 0 : (m & 2) != 0;
}

template <int BM, int BN, int BK = 16>
int launch_taps(IgemmTapsParams p, hipStream_t s) {
  p.variant = NNL_AB_INT("NNL_IGEMM_VARIANT", 1);   // 1 = pipelined LDS fragment reads (+2-3 % on BK=32)
  p.epi4 = NNL_AB_INT("NNL_IGEMM_EPI4", 1);
  p.grid_m = (int)nnl_cdiv(p.M, BM);
  p.grid_n = (int)nnl_cdiv(p.Nc, BN);
  p.cls_tiles = p.grid_m * p.grid_n;
  const unsigned gx = (unsigned)(p.grid_m * p.grid_n * (p.ncls > 1 ? p.ncls : 1));
  if constexpr (BM == 64 && BN == 64) {
    if (taps_dma(BK, p)) {
      hipLaunchKernelGGL((igemm_taps_kernel<64, 64, BK, 2, 2, false, 0, true>), dim3(gx, p.ksplit > 1 ? p.ksplit : 1), dim3(256), 0, s, p);
      NNL_CHECK_LAUNCH();
      return NNL_OK;
    }
  }
  if constexpr (BM == 64 && BN == 64) {
    if (taps_pf2(BK, p)) {
      hipLaunchKernelGGL((igemm_taps_kernel<64, 64, BK, 2, 2, true, 0, false, 2>), dim3(gx, p.ksplit > 1 ? p.ksplit : 1), dim3(256), 0, s, p);
      NNL_CHECK_LAUNCH();
      return NNL_OK;
    }
  }
  if (BM == 64 && BN == 64 && p.variant == 1)
    hipLaunchKernelGGL((igemm_taps_kernel<BM, BN, BK, 2, 2, true>), dim3(gx, p.ksplit > 1 ? p.ksplit : 1), dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL((igemm_taps_kernel<BM, BN, BK, 2, 2>), dim3(gx, p.ksplit > 1 ? p.ksplit : 1), dim3(256), 0, s, p);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

// one tap, C % 4 == 0 but not a multiple of the k block: the KTAIL instantiation of the 64x64 kernel (igemm_taps.h)
template <int BK>
int launch_taps_ktail(IgemmTapsParams p, hipStream_t s) {
  p.variant = 1;
  p.epi4 = NNL_AB_INT("NNL_IGEMM_EPI4", 1);
  p.grid_m = (int)nnl_cdiv(p.M, 64);
  p.grid_n = (int)nnl_cdiv(p.Nc, 64);
  p.cls_tiles = p.grid_m * p.grid_n;
  hipLaunchKernelGGL((igemm_taps_kernel<64, 64, BK, 2, 2, true, 0, false, 1, true>), dim3((unsigned)(p.grid_m * p.grid_n), 1), dim3(256), 0, s, p);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

// ---- balanced schedule for the 64x64 taps kernel (see IgemmTapsParams::bal) --------------------------------------------
// The plan is a pure function of the GEMM shape, so the *_workspace_bytes() query and the launch agree.
struct BalPlan {
  int on, bk, main_ks, n_main_tiles, tail_slices, tail_row0;
  size_t main_floats, tail_floats;        // workspace: [main slabs][tail slabs]
  int bm;                                 // tile rows: 64 (64x64 tile) or 128 (128x64 tile, BK 16)
  double t_us;                            // predicted time of the chosen plan
};

constexpr int kCUs = 256;
constexpr long kTileCounters = 65536;   // ints in the caller's persistent tile-counter buffer (nnl_conv2d_tile_counters())

BalPlan plan_balance_tile(long M, int Nc, int C, int ntaps, int bm) {
  BalPlan best{};
  best.bm = bm;
  const int e_bal = NNL_ENV_INT("NNL_IGEMM_BALANCE", 1);
  if (e_bal == 0 || Nc % 4 != 0) return best;
  const long gm = nnl_cdiv(M, bm), gn = nnl_cdiv(Nc, 64), T = gm * gn;
  const int e_bk = NNL_AB_INT("NNL_IGEMM_BK32", -1);
  const int bk = (bm == 64 && (e_bk >= 0 ? e_bk : (T < 1200 || C >= 256 || C == 64)) && C % 32 == 0) ? 32 : 16;
  const long I = (long)ntaps * (C / bk);                               // k iterations of a whole tile
  const double c_it = (bk == 32 ? 0.60 : 0.30) * (bm / 64);            // us per k iteration per CU-resident workgroup set (measured ~113 TF/s ceiling)
  const double occ = bm == 128 ? 5 : (bk == 32 ? 4 : 6);               // resident workgroups per CU (LDS- / VGPR-limited)
  auto wave_iters = [&](long blocks, long iters) {                     // busiest CU's iterations for `blocks` equal workgroups
    if (blocks <= 0) return 0.0;
    const long cap = (long)occ * kCUs;                                 // full residency waves, then the remainder on top
    const long full = blocks / cap, rem = blocks - full * cap;         // (measured: 1568 workgroups at occupancy 6 take 7 units)
    return (double)(full * (long)occ + nnl_cdiv(rem, kCUs)) * iters;
  };
  const double plain = wave_iters(T, I) * c_it;
  best.t_us = plain;
  double best_t = plain * (e_bal == 2 ? 1.25 : 0.99);         // need a >= 1 % predicted win (2 = force, for A/B runs)
  const int plan_extra = NNL_AB_INT("NNL_IGEMM_PLAN_EXTRA", 2);                       // k iterations' worth of fix-up cost per sliced workgroup
  const double plan_bw = NNL_AB_INT("NNL_IGEMM_PLAN_BW", 16000) * 1.0e3;                  // slab traffic bandwidth, bytes per us
  const int f_ks = NNL_AB_INT("NNL_IGEMM_PLAN_KS", 0), f_S = NNL_AB_INT("NNL_IGEMM_PLAN_S", 0);   // A/B hooks: force the plan's k slicing
  if (f_ks > 0 || f_S > 0) best_t = 1e300;
  for (int ks = 1; ks <= 4; ks *= 2) {
    if (f_ks > 0 && ks != f_ks) continue;
    if (I / ks < 8) break;
    const long units = T * ks;
    long n_main = ((units / kCUs) * kCUs / ks / gn) * gn;              // main tiles: whole multiples of 256 workgroups, whole tile rows
    if (n_main > T) n_main = T;
    const long tail = T - n_main;
    const long it_main = nnl_cdiv(I, ks);
    static const int kSlices[] = {1, 2, 3, 4, 6, 8, 9, 12, 16, 18, 24, 32, 36, 48};
    for (int S : kSlices) {
      if (tail == 0 && S > 1) break;
      if (S > 1 && I / S < 4) break;
      if (f_S > 0 && tail > 0 && S != f_S) continue;
      const long it_tail = nnl_cdiv(I, S);
      const long tail_blocks = tail * S;
      double t = (wave_iters(n_main * ks, it_main) + wave_iters(tail_blocks, it_tail + (S > 1 ? plan_extra : 0))) * c_it;
      const long row0 = (n_main / gn) * bm < M ? (n_main / gn) * bm : M;
      const double main_b = ks > 1 ? (2.0 * ks + 1) * row0 * Nc * 4 : 0;
      const double tail_b = S > 1 ? (2.0 * S + 1) * (M - row0) * Nc * 4 : 0;
      t += (main_b + tail_b) / plan_bw + (ks > 1 ? 1 : 0) + (S > 1 && tail ? 1 : 0);      // slab traffic (write + re-read) + fix-up latency
      if (t < best_t) {
        best_t = t;
        best.t_us = t;
        best.on = 1; best.bk = bk; best.main_ks = ks; best.n_main_tiles = (int)n_main; best.tail_slices = tail ? S : 1;
        best.tail_row0 = (int)row0;
        best.main_floats = ks > 1 ? (size_t)ks * row0 * Nc : 0;
        best.tail_floats = (S > 1 && tail) ? (size_t)S * (M - row0) * Nc : 0;
      }
    }
  }
  if (best.on && best.main_ks == 1 && best.tail_slices == 1) best.on = 0;
  return best;
}

BalPlan plan_balance(long M, int Nc, int C, int ntaps) {
  // (a 128x64 tile under this schedule was measured 10-12 % slower on the 56x56 and 14x14 stages and removed: profiles/README.md)
  return plan_balance_tile(M, Nc, C, ntaps, 64);
}

size_t balance_workspace_bytes(long M, int Nc, int C, int ntaps) {
  const BalPlan pl = plan_balance(M, Nc, C, ntaps);
  return pl.on ? (pl.main_floats + pl.tail_floats) * sizeof(float) : 0;
}

// out[i] = epilogue( sum_s part[s*slab_stride4 + i] ), fixed order (bitwise reproducible); float4 granularity, Nc % 4 == 0
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ part, long slab_stride4, int nslabs,
                                                           float* __restrict__ out, long n4, const float* __restrict__ bias,
                                                           const float* __restrict__ add, int Nc4, int relu) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f32x4 acc = reinterpret_cast<const f32x4*>(part)[i];
  for (int sl = 1; sl < nslabs; ++sl) acc += reinterpret_cast<const f32x4*>(part)[(long)sl * slab_stride4 + i];
  if (bias) acc += reinterpret_cast<const f32x4*>(bias)[i % Nc4];
  if (add) acc += reinterpret_cast<const f32x4*>(add)[i];
  if (relu == 1) { acc[0] = fmaxf(acc[0], 0.f); acc[1] = fmaxf(acc[1], 0.f); acc[2] = fmaxf(acc[2], 0.f); acc[3] = fmaxf(acc[3], 0.f); }
  else if (relu == 2) { acc[0] = nnl_sigmoid(acc[0]); acc[1] = nnl_sigmoid(acc[1]); acc[2] = nnl_sigmoid(acc[2]); acc[3] = nnl_sigmoid(acc[3]); }
  reinterpret_cast<f32x4*>(out)[i] = acc;
}

int launch_balanced(IgemmTapsParams p, const BalPlan& pl, float* ws, int* counters, hipStream_t s) {
  p.variant = NNL_AB_INT("NNL_IGEMM_VARIANT", 1);   // 1 = pipelined LDS fragment reads (+2-3 % on BK=32)
  p.epi4 = NNL_AB_INT("NNL_IGEMM_EPI4", 1);
  p.grid_m = (int)nnl_cdiv(p.M, pl.bm);
  p.grid_n = (int)nnl_cdiv(p.Nc, 64);
  const int T = p.grid_m * p.grid_n;
  p.bal = 1; p.main_ks = pl.main_ks; p.n_main_tiles = pl.n_main_tiles; p.tail_slices = pl.tail_slices; p.tail_row0 = pl.tail_row0;
  p.main_out = ws; p.main_slab_stride = (long)pl.tail_row0 * p.Nc;
  p.tail_out = ws + pl.main_floats; p.tail_slab_stride = (long)(p.M - pl.tail_row0) * p.Nc;
  p.tile_counters = (pl.bm == 64) ? counters : nullptr;      // in-kernel fix-up of the split tiles (64x64 tile only)
  const unsigned grid = (unsigned)(pl.n_main_tiles * pl.main_ks + (T - pl.n_main_tiles) * pl.tail_slices);
  if (p.ktail && pl.bk == 32)
    hipLaunchKernelGGL((igemm_taps_kernel<64, 64, 32, 2, 2, true, 0, false, 1, true>), dim3(grid), dim3(256), 0, s, p);
  else if (p.ktail)
    hipLaunchKernelGGL((igemm_taps_kernel<64, 64, 16, 2, 2, true, 0, false, 1, true>), dim3(grid), dim3(256), 0, s, p);
  else if (pl.bk == 32 && taps_dma(32, p))
    hipLaunchKernelGGL((igemm_taps_kernel<64, 64, 32, 2, 2, false, 0, true>), dim3(grid), dim3(256), 0, s, p);
  else if (pl.bk != 32 && taps_dma(16, p))
    hipLaunchKernelGGL((igemm_taps_kernel<64, 64, 16, 2, 2, false, 0, true>), dim3(grid), dim3(256), 0, s, p);
  else if (pl.bk == 32 && taps_pf2(32, p))
    hipLaunchKernelGGL((igemm_taps_kernel<64, 64, 32, 2, 2, true, 0, false, 2>), dim3(grid), dim3(256), 0, s, p);
  else if (pl.bk != 32 && taps_pf2(16, p))
    hipLaunchKernelGGL((igemm_taps_kernel<64, 64, 16, 2, 2, true, 0, false, 2>), dim3(grid), dim3(256), 0, s, p);
  else if (pl.bk == 32 && p.variant == 1)
    hipLaunchKernelGGL((igemm_taps_kernel<64, 64, 32, 2, 2, true>), dim3(grid), dim3(256), 0, s, p);
  else if (pl.bk == 32)
    hipLaunchKernelGGL((igemm_taps_kernel<64, 64, 32, 2, 2>), dim3(grid), dim3(256), 0, s, p);
  else if (p.variant == 1)
    hipLaunchKernelGGL((igemm_taps_kernel<64, 64, 16, 2, 2, true>), dim3(grid), dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL((igemm_taps_kernel<64, 64, 16, 2, 2>), dim3(grid), dim3(256), 0, s, p);
  NNL_CHECK_LAUNCH();
  if (p.tile_counters != nullptr) return NNL_OK;             // the kernel reduced its own slabs
  const int Nc4 = p.Nc / 4;
  if (pl.main_ks > 1 && pl.tail_row0 > 0) {
    const long n4 = (long)pl.tail_row0 * Nc4;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)nnl_cdiv(n4, 256)), dim3(256), 0, s, (const float*)p.main_out, n4, pl.main_ks,
                       p.y, n4, p.bias, p.add, Nc4, p.relu);
    NNL_CHECK_LAUNCH();
  }
  if (pl.tail_slices > 1 && p.M > pl.tail_row0) {
    const long n4 = (long)(p.M - pl.tail_row0) * Nc4;
    const long off = (long)pl.tail_row0 * p.Nc;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)nnl_cdiv(n4, 256)), dim3(256), 0, s, (const float*)p.tail_out, n4,
                       pl.tail_slices, p.y + off, n4, p.bias, p.add ? p.add + off : nullptr, Nc4, p.relu);
    NNL_CHECK_LAUNCH();
  }
  return NNL_OK;
}

// Tile choice: estimated time = rounds of resident workgroups x per-workgroup work / per-tile MFMA efficiency.
// Resident workgroups per CU (VGPR/LDS limited) and efficiencies are measured values (tools/bench_conv.py).
int dispatch_taps(const IgemmTapsParams& p_in, hipStream_t s, void* ws = nullptr, size_t ws_bytes = 0, int* counters = nullptr,
                  int* bn_rows = nullptr) {
  IgemmTapsParams p = p_in;
#ifdef NNL_TAPS_TIMING
  p.tile_counters = counters;                       // debug builds: the timestamp area lives behind the counters (igemm_taps.h)
#endif
  if (bn_rows) *bn_rows = 0;
  if (p.ktail) {
    const bool dense = p.out_stride == 1 && p.OH == p.P && p.OW == p.Q && p.oh0 == 0 && p.ow0 == 0 && p.ksplit <= 1;
    float* const bn_part_k = p.bn_part;
    p.bn_part = nullptr;
    if (ws != nullptr && dense) {
      // few, long tiles (1024 x 500 x 1000: 128 workgroups of 32 k steps): the balanced schedule's k slices fill the chip
      const BalPlan pl = plan_balance(p.M, p.Nc, (int)nnl_cdiv(p.C, 32) * 32, 1);
      if (pl.on && ws_bytes >= (pl.main_floats + pl.tail_floats) * sizeof(float)) {
        const long tiles = nnl_cdiv(p.M, pl.bm) * nnl_cdiv(p.Nc, 64);
        int* const cnt = tiles <= kTileCounters ? counters : nullptr;
        if (bn_part_k && cnt != nullptr) {
          p.bn_part = bn_part_k;
          if (bn_rows) *bn_rows = (int)nnl_cdiv(p.M, 64);
        }
        return launch_balanced(p, pl, (float*)ws, cnt, s);
      }
    }
    if (bn_part_k && dense) {
      p.bn_part = bn_part_k;
      if (bn_rows) *bn_rows = (int)nnl_cdiv(p.M, 64);
    }
    return p.C >= 64 ? launch_taps_ktail<32>(p, s) : launch_taps_ktail<16>(p, s);
  }
  const int forced = NNL_AB_INT("NNL_IGEMM_TILE", -1);
  struct Cand { int bm, bn, occ; double eff; };
  static const Cand cands[4] = {{128, 128, 4, 0.90}, {128, 64, 5, 0.90}, {64, 128, 5, 0.90}, {64, 64, 8, 1.00}};   // measured: bench_conv.py, NNL_IGEMM_TILE sweep
  int best = 0;
  if (forced >= 0 && forced < 4) {
    best = forced;
  } else if (p.ntaps == 1 && p.C <= 512 && p.Nc >= 8192 && p.M >= 1024 && p.ncls <= 1) {
    // a Linear onto a huge vocabulary (the AWD-LSTM decoder, Text.py:572: 4480 x 400 -> 47 343): 25 k steps per tile, so the launch is
    // prologue / epilogue / B-re-read bound and the 128 x 128 tile wins (measured, tools/bench_decoder_gemm.py: 1.81 -> 1.49 ms)
    best = 0;
  } else {
    double best_t = 1e300;
    for (int i = 0; i < 4; ++i) {
      const Cand& c = cands[i];
      const long blocks = nnl_cdiv(p.M, c.bm) * nnl_cdiv(p.Nc, c.bn) * (p.ncls > 1 ? p.ncls : 1);
      const long slots = 256L * c.occ;
      const long rounds = nnl_cdiv(blocks, slots);
      // inside one round the CU is shared by min(occ, blocks/256) workgroups: time ~ (workgroups on the busiest CU) x tile work
      const long last = blocks - (rounds - 1) * slots;
      const double per_cu = (double)(rounds - 1) * c.occ + (double)nnl_cdiv(last, 256);
      const double t = per_cu * c.bm * c.bn / c.eff;
      if (t < best_t) { best_t = t; best = i; }
    }
  }
  float* const bn_part = p.bn_part;                 // statistics are produced by the 64x64 kernels only
  p.bn_part = nullptr;
  switch (best) {
    case 0: return launch_taps<128, 128>(p, s);
    case 1: return launch_taps<128, 64>(p, s);
    case 2: return launch_taps<64, 128>(p, s);
    default: {
      const bool dense_out = p.out_stride == 1 && p.OH == p.P && p.OW == p.Q && p.oh0 == 0 && p.ow0 == 0 && p.ksplit <= 1;
      if (ws != nullptr && dense_out) {
        const BalPlan pl = plan_balance(p.M, p.Nc, p.C, p.ntaps);
        if (pl.on && ws_bytes >= (pl.main_floats + pl.tail_floats) * sizeof(float)) {
          const long tiles = nnl_cdiv(p.M, pl.bm) * nnl_cdiv(p.Nc, 64);
          int* const cnt = tiles <= kTileCounters ? counters : nullptr;
          if (bn_part && dense_out && pl.bm == 64 && cnt != nullptr) {      // split tiles are finished in-kernel: stats too
            p.bn_part = bn_part;
            if (bn_rows) *bn_rows = (int)nnl_cdiv(p.M, 64);
          }
          return launch_balanced(p, pl, (float*)ws, cnt, s);
        }
      }
      if (bn_part && dense_out) {
        p.bn_part = bn_part;
        if (bn_rows) *bn_rows = (int)nnl_cdiv(p.M, 64);
      }
      // BK=32 halves the barriers per MFMA at half the occupancy: measured (bench_conv.py --ab NNL_IGEMM_BK32=0,1) +10..20 %
      // on grids of < ~5 workgroups per CU (14x14 / 7x7 stages), -7 % on the 56x56 stage.  NNL_IGEMM_BK32=0/1 overrides.
      const int e_bk = NNL_AB_INT("NNL_IGEMM_BK32", -1);
      const long blocks64 = nnl_cdiv(p.M, 64) * nnl_cdiv(p.Nc, 64) * (p.ncls > 1 ? p.ncls : 1);
      // long k loops (C >= 256) gain from BK=32 on large grids too (RetinaNet heads); so does C = 64 since the prologue / per-tap clean-ups
      // of round 3 (re-measured per layer at 64 images: l1 3x3 0.143 -> 0.138 ms; the C = 128 stage still prefers BK=16: 0.138 vs 0.147)
      const int bk32 = e_bk >= 0 ? e_bk : (blocks64 < 1200 || p.C >= 256 || p.C == 64);
      if (bk32 && p.C % 32 == 0) return launch_taps<64, 64, 32>(p, s);
      return launch_taps<64, 64>(p, s);
    }
  }
}

bool taps_ok(long a_elems, long b_elems, int C, int ntaps) {
  return C % 16 == 0 && ntaps <= IGEMM_MAX_TAPS && a_elems * 4 < (1L << 31) && b_elems * 4 < (1L << 32) - 64;
}

// 1: the tap-table kernel as is; 2: its KTAIL instantiation (one tap, C a multiple of 4 only: NNL_IGEMM_KTAIL=0 sends those shapes
// back to the first-generation kernel); 0: neither
int taps_kind(long a_elems, long b_elems, int C, int ntaps) {
  if (taps_ok(a_elems, b_elems, C, ntaps)) return 1;
  if (ntaps == 1 && C % 4 == 0 && C >= 32 && a_elems * 4 < (1L << 31) && b_elems * 4 < (1L << 32) - 64 && NNL_ENV_INT("NNL_IGEMM_KTAIL", 1) != 0)
    return 2;
  return 0;
}

__global__ void weight_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int K, int RS, int C) {
  // w [K][RS][C] -> wt [C][RS][K]; 32x32 LDS tile transpose over (K, C) for each tap
  __shared__ float tile[32][33];
  const int tap = blockIdx.z;
  const int k0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int k = k0 + i, c = c0 + tx;
    tile[i][tx] = (k < K && c < C) ? w[((long)k * RS + tap) * C + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, k = k0 + tx;
    if (k < K && c < C) wt[((long)c * RS + tap) * K + k] = tile[tx][i];
  }
}

// all conv filters of a model in ONE launch: descriptor d = {w, wt, K, RS, C, first tile}; blockIdx.x -> (tensor, tap, k tile, c tile)
__global__ void weight_transpose_multi_kernel(const nnl_wt_desc_t* __restrict__ desc, const int* __restrict__ tile_tensor) {
  __shared__ float tile[32][33];
  const nnl_wt_desc_t d = desc[tile_tensor[blockIdx.x]];
  int t = (int)blockIdx.x - d.first_tile;
  const int ct = (d.C + 31) / 32, kt = (d.K + 31) / 32;
  const int c0 = (t % ct) * 32; t /= ct;
  const int k0 = (t % kt) * 32;
  const int tap = t / kt;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int k = k0 + i, c = c0 + tx;
    tile[i][tx] = (k < d.K && c < d.C) ? d.w[((long)k * d.RS + tap) * d.C + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, k = k0 + tx;
    if (k < d.K && c < d.C) d.wt[((long)c * d.RS + tap) * d.K + k] = tile[tx][i];
  }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, long n4,
                                                             int splits) {
  // out[i] = sum_s part[s][i] in a FIXED order (bitwise reproducible).  256 threads = 32 float4 columns x 8 split lanes:
  // lane l adds slabs l, l+8, ...; the 8 partial sums are then added in lane order through LDS.
  __shared__ f32x4 red[8][32];
  const int col = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const long i = (long)blockIdx.x * 32 + col;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (i < n4) {
    // lane `sl` adds slabs sl, sl+8, ... IN THAT ORDER; four loads in flight per trip (the plain loop waited for each load)
    int s = sl;
    for (; s + 24 < splits; s += 32) {
      const f32x4 v0 = reinterpret_cast<const f32x4*>(part)[(long)s * n4 + i], v1 = reinterpret_cast<const f32x4*>(part)[(long)(s + 8) * n4 + i];
      const f32x4 v2 = reinterpret_cast<const f32x4*>(part)[(long)(s + 16) * n4 + i], v3 = reinterpret_cast<const f32x4*>(part)[(long)(s + 24) * n4 + i];
      acc += v0; acc += v1; acc += v2; acc += v3;
    }
    for (; s < splits; s += 8) acc += reinterpret_cast<const f32x4*>(part)[(long)s * n4 + i];
  }
  red[sl][col] = acc;
  __syncthreads();
  if (sl == 0 && i < n4) {
    f32x4 t = red[0][col];
#pragma unroll
    for (int l = 1; l < 8; ++l) t += red[l][col];
    reinterpret_cast<f32x4*>(out)[i] = t;
  }
}

// column sums in two fixed-order stages (bitwise reproducible): stage 1 reduces a chunk of rows per block into
// part[chunk][col], stage 2 adds the chunks of one column in order.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ a, float* __restrict__ part, long rows,
                                                              int cols, long rows_per_chunk) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const long r0 = (long)blockIdx.y * rows_per_chunk;
  const long r1 = r0 + rows_per_chunk < rows ? r0 + rows_per_chunk : rows;
  float acc = 0.f;
  if (c < cols) {
    long r = r0 + rl;
    for (; r + 12 < r1; r += 16) {                       // 4 independent loads in flight per lane
      const float v0 = a[r * cols + c], v1 = a[(r + 4) * cols + c], v2 = a[(r + 8) * cols + c], v3 = a[(r + 12) * cols + c];
      acc += (v0 + v1) + (v2 + v3);
    }
    for (; r < r1; r += 4) acc += a[r * cols + c];
  }
  red[rl][threadIdx.x & 63] = acc;
  __syncthreads();
  if (rl == 0 && c < cols)
    part[(long)blockIdx.y * cols + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ReLU backward gate fused with the bias-gradient column sums: g = dy * [y > 0] is written once and summed per column in the same
// pass (the conv + bias + ReLU layers of the RetinaNet heads: two ATen kernels and one pass over dy less per layer)
// ACT 1: ReLU gate g = dy * [y > 0];  ACT 2: sigmoid gate g = dy * y * (1 - y) (y = the sigmoid's OUTPUT: ClassificationModel,
// retinanet.py:286 — torch's sigmoid_backward formula)
template <int ACT>
__device__ __forceinline__ float act_gate(float dy, float y) {
  if (ACT == 1) return y > 0.f ? dy : 0.f;
  return dy * ((1.f - y) * y);
}

template <int ACT>
__global__ __launch_bounds__(256) void act_gate_colsum_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                               float* __restrict__ g, float* __restrict__ part, long rows, int cols,
                                                               long rows_per_chunk) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const long r0 = (long)blockIdx.y * rows_per_chunk;
  const long r1 = r0 + rows_per_chunk < rows ? r0 + rows_per_chunk : rows;
  float acc = 0.f;
  if (c < cols) {
    long r = r0 + rl;
    for (; r + 12 < r1; r += 16) {                       // 8 independent loads in flight per lane
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long o = (r + 4 * u) * cols + c;
        v[u] = act_gate<ACT>(dy[o], y[o]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) g[(r + 4 * u) * cols + c] = v[u];
      acc += (v[0] + v[1]) + (v[2] + v[3]);
    }
    for (; r < r1; r += 4) {
      const long o = r * cols + c;
      const float v = act_gate<ACT>(dy[o], y[o]);
      g[o] = v;
      acc += v;
    }
  }
  red[rl][threadIdx.x & 63] = acc;
  __syncthreads();
  if (rl == 0 && c < cols && part != nullptr)
    part[(long)blockIdx.y * cols + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, float* __restrict__ out, int cols,
                                                            int nchunk) {
  // 256 threads = 16 columns x 16 chunk lanes: lane l adds chunks l, l+16, ... (independent loads), then the 16 partial
  // sums of a column are added in lane order through LDS (fixed order => reproducible)
  __shared__ float red[16][17];
  const int cl = threadIdx.x & 15, kl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float acc = 0.f;
  if (c < cols)
    for (int k = kl; k < nchunk; k += 16) acc += part[(long)k * cols + c];
  red[kl][cl] = acc;
  __syncthreads();
  if (kl == 0 && c < cols) {
    float t = red[0][cl];
#pragma unroll
    for (int l = 1; l < 16; ++l) t += red[l][cl];
    out[c] = t;
  }
}

int colsum_chunks(long rows) {
  long n = nnl_cdiv(rows, 64);
  if (n > 256) n = 256;
  if (n < 1) n = 1;
  return (int)n;
}

struct WgradPlan { int bm, bn, grid_m, grid_n, splits, k_per_split, kg; };

// Tile AND split count are searched together for the shortest predicted launch: all workgroups are resident at once
// (occupancy 4-5), so a launch lasts ceil(blocks/256) x (pixels per split) x (time per workgroup-pixel of the tile); rounding
// k_per_split to 32 and a short last split are accounted for by using the real split length; fewer than 4 workgroups per CU
// cannot hide the load latency; the slab reduce moves (splits+1) x |dw| bytes.  The smaller tiles pay a measured
// efficiency factor (bench_conv.py, NNL_WGRAD_TILE sweep) but give short pixel ranges (1x1/2 shortcut convs: 12544 pixels)
// enough workgroups to fill 256 CUs.
WgradPlan plan_wgrad(int Mc, int Nc, long Kp, int square_bn_divides = 0) {     // != 0: square tiles only, bn must divide it (Winograd columns)
  struct Cand { int bm, bn; double cost; };                               // cost: time per FLOP relative to the 128x128 tile
  static const Cand cands[4] = {{128, 128, 1.00}, {128, 64, 1.08}, {64, 128, 1.08}, {64, 64, 1.10}};
  const int forced = NNL_AB_INT("NNL_WGRAD_TILE", -1);                    // tuning hook: index into cands
  const long max_splits = Kp / 256 > 0 ? Kp / 256 : 1;                    // at least 256 pixels (16 k-steps) per split
  WgradPlan pl{};
  double best_t = 1e300;
  for (int ci = 0; ci < 4; ++ci) {
    const Cand& c = cands[ci];
    if (square_bn_divides != 0) {
      if (c.bm != c.bn || square_bn_divides % c.bn != 0 || (c.bm == 128 && Mc < 128)) continue;
      const int f_w = NNL_AB_INT("NNL_WGRAD_WINO_TILE", -1);          // A/B hook: 0 = 128x128, 3 = 64x64 (when legal)
      if ((f_w == 0 || f_w == 3) && ci != f_w && !(f_w == 0 && (Mc < 128 || square_bn_divides % 128 != 0))) continue;
    } else if (forced >= 0 && forced < 4 ? ci != forced : ((c.bm == 128 && Mc < 128) || (c.bn == 128 && Nc < 128))) continue;
    const long tiles = nnl_cdiv(Mc, c.bm) * nnl_cdiv(Nc, c.bn);
    const double us_per_px = (double)c.bm * c.bn * 2.0 / 441e3 * c.cost;   // one workgroup-pixel at ~113 TF/s / 256 CUs
    const int f_sp = NNL_AB_INT("NNL_WGRAD_SPLITS", 0);                  // A/B hook: force the split count
    for (long sp = 1; sp <= max_splits && (sp == 1 || tiles * sp <= 256 * 5 || f_sp > 0 || NNL_AB_INT("NNL_WGRAD_WGPCU10", 0) > 0); ++sp) {   // unsplit is always a candidate
      if (f_sp > 0 && sp != (f_sp < max_splits ? f_sp : max_splits)) continue;
      const int f_wg = NNL_AB_INT("NNL_WGRAD_WGPCU10", 0);                // A/B hook: workgroups per CU x 10 (e.g. 20 = two per CU)
      if (f_wg > 0) {
        long want = (long)(f_wg * 25.6 / tiles + 0.5);
        if (want < 1) want = 1;
        if (want > max_splits) want = max_splits;
        if (sp != want) continue;
      }
      const long k1 = nnl_cdiv(nnl_cdiv(Kp, sp), 32) * 32;
      const long rs = nnl_cdiv(Kp, k1);
      if (rs != sp) continue;                                             // same plan as a smaller sp
      const long per_cu = nnl_cdiv(tiles * rs, 256);
      const double starve = per_cu < 4 ? pow(4.0 / per_cu, NNL_AB_INT("NNL_WGRAD_STARVE_PCT", 30) * 0.01) : 1.0;   // A/B hook: exponent x100
      const double t = (double)per_cu * k1 * us_per_px * starve + (rs > 1 ? (rs + 1.0) * Mc * Nc * 4 / 4.5e6 + 3 : 0);   // reduce: rs slab reads + one write at ~4.5 TB/s
      if (t < best_t) {
        best_t = t;
        pl.bm = c.bm; pl.bn = c.bn; pl.grid_m = (int)nnl_cdiv(Mc, c.bm); pl.grid_n = (int)nnl_cdiv(Nc, c.bn);
        pl.splits = (int)rs; pl.k_per_split = (int)k1;
      }
    }
  }
  if (pl.bm == 0) {                                                       // more tiles than 5 per CU even unsplit: largest legal tile
    pl.bm = Mc >= 128 ? 128 : 64; pl.bn = (Nc >= 128 && pl.bm == 128) ? 128 : 64;
    if (square_bn_divides != 0 && square_bn_divides % pl.bn != 0) { pl.bm = 64; pl.bn = 64; }
    pl.grid_m = (int)nnl_cdiv(Mc, pl.bm); pl.grid_n = (int)nnl_cdiv(Nc, pl.bn);
    pl.splits = 1; pl.k_per_split = (int)(nnl_cdiv(Kp, 32) * 32);
  }
  // Merge KG neighbouring splits into ONE workgroup of KG wave groups (igemm_wgrad.h): the same waves per CU, the partial sums of
  // the KG pixel ranges meet in LDS, and only every KG-th slab is written / re-read (round 2 measured 1.97x the algorithmic HBM
  // bytes per conv launch, almost all of it wgrad slabs).  Instantiated for the 128x128 (BK 16) and 64x64 (BK 32) tiles; the
  // planner takes it only where it measured as a win (tools/bench_conv.py --ab NNL_WGRAD_KG=1,2,4 at 64 images): the 128x128 tile
  // with >= 16 splits (28 x 28 / 14 x 14 stages: 0.139 -> 0.136 ms and 66 -> 17 MB of slabs per launch).  With few splits the merge
  // unbalances the grid (7 x 7 stage, 7 splits: 0.142 -> 0.187 ms) and the 64x64 tile loses 1-3 %.
  pl.kg = 1;
  const int e_kg = NNL_AB_INT("NNL_WGRAD_KG", -1);                       // A/B hook: 1 = off, 2 / 4 = force (when legal)
  const bool kg_tile = (pl.bm == 128 && pl.bn == 128) || (pl.bm == 64 && pl.bn == 64);
  if (kg_tile && e_kg != 1 && pl.splits >= 2) {
    int kg = (pl.bm == 128 && pl.splits >= 16) ? 4 : 1;
    // Winograd-domain columns (square_bn_divides = 3 * C): the 4-group 128x128 WINO instantiation spills (two staged pixels per
    // operand: 10 VGPRs over), and merged groups measured SLOWER there — RetinaNet FPN level 16 x 64 x 64, C = K = 256: 0.49 ms
    // with one group per workgroup, 0.75 with two, 0.94 with four (direct kernel: 0.66)
    if (square_bn_divides != 0 && pl.bm == 128) kg = 1;
    if (e_kg == 2 || e_kg == 4) kg = pl.splits >= e_kg ? e_kg : 1;
    if (kg > 1) {
      const long sp = nnl_cdiv(pl.splits, kg);
      const long k1 = nnl_cdiv(nnl_cdiv(Kp, sp), 32L * kg) * 32L * kg;    // each group's share stays a multiple of 32 pixels
      pl.kg = kg; pl.k_per_split = (int)k1; pl.splits = (int)nnl_cdiv(Kp, k1);
    }
  }
  return pl;
}

bool wgrad_v2_ok(long a_elems, long b_elems, long Kp) {
  return a_elems * 4 < (1L << 32) - 64 && b_elems * 4 < (1L << 32) - 64 && Kp < (1L << 23);
}

// launches igemm_wgrad_kernel for plan pl; out = dw or the split-K slab workspace
int launch_wgrad_v2(const float* dy, const float* x, float* out, long a_elems, long b_elems, int H, int W, int C, int P, int Q,
                    int R, int S, int stride, int pad, int Mc, int Nc, long Kp, const WgradPlan& pl, hipStream_t s) {
  IgemmWgradParams q{};
  q.a = dy; q.b = x; q.y = out;
  q.a_bytes = (unsigned)(a_elems * 4); q.b_bytes = (unsigned)(b_elems * 4);
  q.H = H; q.W = W; q.C = C; q.P = P; q.Q = Q; q.R = R; q.S = S; q.stride = stride; q.pad = pad;
  q.Mc = Mc; q.Nc = Nc; q.Kp = (int)Kp;
  q.splits = pl.splits; q.k_per_split = pl.k_per_split; q.grid_m = pl.grid_m; q.grid_n = pl.grid_n;
#ifdef NNL_TAPS_TIMING
  { static void* dbg = nullptr; if (!dbg) (void)hipMalloc(&dbg, 3276 * 5 * 8); q.dbg_t = (unsigned long long*)dbg; g_wgrad_dbg = dbg; }
#endif
  {
    // Tile order inside a split: ~128 consecutive tiles run together on one XCD (32 CUs x 4 workgroups) and walk the pixel
    // range in step; per pixel row they touch (distinct row tiles) x bm columns of dy and (distinct column tiles) x bn of x.
    // Pick the order with the smaller footprint (the 47343 x 400 decoder gradient: 16.5k floats per pixel row with row tiles
    // fastest, 4.6k with column tiles fastest).
    const long T = 128, gm = pl.grid_m, gn = pl.grid_n;
    const long fp_m = (T < gm ? T : gm) * pl.bm + nnl_cdiv(T, gm) * pl.bn;
    const long fp_n = nnl_cdiv(T, gn) * pl.bm + (T < gn ? T : gn) * pl.bn;
    { const int e_nf = NNL_AB_INT("NNL_WGRAD_NFAST", -1); q.n_fast = e_nf >= 0 ? e_nf : (fp_n < fp_m ? 1 : 0); }
  }
  const dim3 grid(pl.grid_m * pl.grid_n * pl.splits);
  const int bk32 = NNL_AB_INT("NNL_WGRAD_BK32", 1);                  // 64x64 tile: BK=32 (16 MFMAs per barrier) measured +3 %
  const int pipe = NNL_AB_INT("NNL_WGRAD_PIPE", 1);                 // A/B hook: 1 = software-pipelined fragment reads
  // staging LDS: 2 buffers x BK x (BM + BN) floats per wave group (dynamic: above 64 KB the kernel needs the attribute once)
  auto lds_bytes = [](int bm, int bn, int bk, int kg) { return (size_t)kg * 2 * bk * (bm + bn) * sizeof(float); };
#define NNL_WGRAD_LAUNCH(BM_, BN_, BK_, PIPE_, KG_, PAIR_)                                                                       \
  do {                                                                                                                           \
    const size_t lb = lds_bytes(BM_, BN_, BK_, KG_);                                                                             \
    if (lb > 64 * 1024) {                                                                                                        \
      static bool attr_set = false;                                                                                              \
      if (!attr_set) {                                                                                                           \
        NNL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_wgrad_kernel<BM_, BN_, BK_, 2, 2, PIPE_, KG_, false, PAIR_>),  \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb));                                 \
        attr_set = true;                                                                                                         \
      }                                                                                                                          \
    }                                                                                                                            \
    hipLaunchKernelGGL((igemm_wgrad_kernel<BM_, BN_, BK_, 2, 2, PIPE_, KG_, false, PAIR_>), grid, dim3(256 * KG_), lb, s, q);    \
  } while (0)
  // PAIR staging (igemm_wgrad.h): both 16-byte chunks a thread stages per row must lie in one filter tap
  const bool pair = (R * S == 1 || C % pl.bn == 0) && NNL_AB_INT("NNL_WGRAD_PAIR", 1) != 0;   // (A/B hook: 0 = the one-chunk staging everywhere)
  if (pl.bm == 128 && pl.bn == 128) {
    if (pl.kg == 4) { if (pair) NNL_WGRAD_LAUNCH(128, 128, 16, true, 4, true); else NNL_WGRAD_LAUNCH(128, 128, 16, true, 4, false); }
    else if (pl.kg == 2) NNL_WGRAD_LAUNCH(128, 128, 16, true, 2, false);
    else if (pipe) { if (pair) NNL_WGRAD_LAUNCH(128, 128, 16, true, 1, true); else NNL_WGRAD_LAUNCH(128, 128, 16, true, 1, false); }
    else NNL_WGRAD_LAUNCH(128, 128, 16, false, 1, false);              // BK=32 measured -7 % here
  } else if (pl.bm == 128) {
    NNL_WGRAD_LAUNCH(128, 64, 16, false, 1, false);
  } else if (pl.bn == 128) {
    NNL_WGRAD_LAUNCH(64, 128, 16, true, 1, false);
  } else {
    if (bk32 && pl.k_per_split % 32 == 0) {
      if (pl.kg == 4) { if (pair) NNL_WGRAD_LAUNCH(64, 64, 32, true, 4, true); else NNL_WGRAD_LAUNCH(64, 64, 32, true, 4, false); }
      else if (pl.kg == 2) NNL_WGRAD_LAUNCH(64, 64, 32, true, 2, false);
      else if (pipe) { if (pair) NNL_WGRAD_LAUNCH(64, 64, 32, true, 1, true); else NNL_WGRAD_LAUNCH(64, 64, 32, true, 1, false); }
      else NNL_WGRAD_LAUNCH(64, 64, 32, false, 1, false);
    }
    else if (pipe) NNL_WGRAD_LAUNCH(64, 64, 16, true, 1, false);
    else NNL_WGRAD_LAUNCH(64, 64, 16, false, 1, false);
  }
#undef NNL_WGRAD_LAUNCH
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

// dW [K][3][3][C] from the Winograd-domain slabs dU [splits][K][4][3][C] (igemm_wgrad_kernel<..., WINO>): the slabs are summed in
// index order (bitwise reproducible), then dg0 = dU0 + (dU1 + dU2)/2, dg1 = (dU1 - dU2)/2, dg2 = (dU1 + dU2)/2 + dU3.
__global__ __launch_bounds__(256) void wino_wgrad_finish_kernel(const float* __restrict__ part, float* __restrict__ dw, long n4,
                                                                 int C4, int splits, long slab4) {
  // 256 threads = 32 (k, r, c4) items x 8 split lanes: lane l adds slabs l, l+8, ... IN THAT ORDER for the four positions, the eight
  // partial sums meet in LDS in lane order (fixed order => bitwise reproducible); lane 0 folds the positions back and writes
  __shared__ f32x4 red[8][32][4];
  const int it = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const long i = (long)blockIdx.x * 32 + it;                           // over (k, r, c4)
  f32x4 u0 = {0.f, 0.f, 0.f, 0.f}, u1 = u0, u2 = u0, u3 = u0;
  long c4 = 0, r = 0, k = 0;
  if (i < n4) {
    c4 = i % C4; const long kr = i / C4; r = kr % 3; k = kr / 3;
    const f32x4* src = reinterpret_cast<const f32x4*>(part) + (k * 12 + r) * C4 + c4;
    for (int s = sl; s < splits; s += 8) {
      const f32x4* q = src + s * slab4;
      u0 += q[0]; u1 += q[3L * C4]; u2 += q[6L * C4]; u3 += q[9L * C4];
    }
  }
  red[sl][it][0] = u0; red[sl][it][1] = u1; red[sl][it][2] = u2; red[sl][it][3] = u3;
  __syncthreads();
  if (sl == 0 && i < n4) {
#pragma unroll
    for (int l = 1; l < 8; ++l) { u0 += red[l][it][0]; u1 += red[l][it][1]; u2 += red[l][it][2]; u3 += red[l][it][3]; }
    f32x4* dst = reinterpret_cast<f32x4*>(dw) + (k * 9 + r * 3) * C4 + c4;
    const f32x4 h = 0.5f * (u1 + u2);
    dst[0] = u0 + h;
    dst[C4] = 0.5f * (u1 - u2);
    dst[2L * C4] = h + u3;
  }
}

// the Winograd-domain weight gradient: 3x3 / stride 1 / pad 1, even Q, C % 4 == 0, square tiles whose width divides 3*C
// NNL_WGRAD_WINO: 0 never, 2 wherever it is legal, 1 (default) where it MEASURED faster than the direct kernel
// (tools/bench_conv.py --ab NNL_WGRAD_WINO=0,2 at 8 / 16 / 32 / 64 images and --net r50 --bs 16; profiles/r3_wwg_*.log):
//   C = 128: +6 ... +21 % everywhere (6272 ... 32768 pairs);          C = 64: +5 / +11 % at 100352 / 131072 pairs, -11 / -26 % at
//   50176 / 25088 (12 column tiles of 64 need ~100 splits of the minimum 256 rows: fixed cost per workgroup);
//   C >= 256 (one wave group per workgroup: see plan_wgrad): +19 % at 6272 pairs (14 x 14 stage, 64 images), +35 ... +40 % on the
//   RetinaNet levels (8192 / 32768 pairs: 0.181 -> 0.130 ms, 0.657 -> 0.488 ms), +23 % at C = 512 / 16 x 16.
//   The weight gradient is summed in a different order either way; both are bitwise reproducible run to run.
bool wgrad_wino_ok(const nnl_conv_geom_t* g) {
  const int mode = NNL_ENV_INT("NNL_WGRAD_WINO", 1);
  if (mode == 0) return false;
  if (g->R != 3 || g->S != 3 || g->stride != 1 || g->pad != 1 || g->Q % 2 != 0 || g->C % 64 != 0 || g->K % 4 != 0) return false;
  const long pairs = (long)g->N * g->P * (g->Q / 2);
  if (pairs < 1024 || !wgrad_v2_ok(2 * pairs * g->K, (long)g->N * g->H * g->W * g->C, pairs)) return false;
  if (mode == 2) return true;
  if (g->C == 64) return pairs >= 65536;
  return pairs >= 2048;                                       // (16 x 16 x 16 levels, C = 256 / 512: 0.058 -> 0.047 ms, 0.181 -> 0.147 ms)
}

WgradPlan plan_wgrad_wino(const nnl_conv_geom_t* g) {
  return plan_wgrad(g->K, 12 * g->C, (long)g->N * g->P * (g->Q / 2), 3 * g->C);
}

int launch_wgrad_wino(const float* dy, const float* x, float* slabs, const nnl_conv_geom_t* g, const WgradPlan& pl, hipStream_t s) {
  IgemmWgradParams q{};
  const long pairs = (long)g->N * g->P * (g->Q / 2);
  q.a = dy; q.b = x; q.y = slabs;
  q.a_bytes = (unsigned)(2 * pairs * g->K * 4); q.b_bytes = (unsigned)((long)g->N * g->H * g->W * g->C * 4);
  q.H = g->H; q.W = g->W; q.C = g->C; q.P = g->P; q.Q = g->Q / 2; q.R = 3; q.S = 3; q.stride = 1; q.pad = 1;
  q.Mc = g->K; q.Nc = 12 * g->C; q.Kp = (int)pairs;
  q.splits = pl.splits; q.k_per_split = pl.k_per_split; q.grid_m = pl.grid_m; q.grid_n = pl.grid_n;
  q.n_fast = 1;
  const dim3 grid(pl.grid_m * pl.grid_n * pl.splits);
  auto lds_bytes = [](int bm, int bn, int bk, int kg) { return (size_t)kg * 2 * bk * (bm + bn) * sizeof(float); };
#define NNL_WGRAD_WINO_LAUNCH(BM_, BN_, BK_, KG_)                                                                                \
  do {                                                                                                                           \
    const size_t lb = lds_bytes(BM_, BN_, BK_, KG_);                                                                             \
    if (lb > 64 * 1024) {                                                                                                        \
      static bool attr_set = false;                                                                                              \
      if (!attr_set) {                                                                                                           \
        NNL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_wgrad_kernel<BM_, BN_, BK_, 2, 2, true, KG_, true>), \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb));                                 \
        attr_set = true;                                                                                                         \
      }                                                                                                                          \
    }                                                                                                                            \
    hipLaunchKernelGGL((igemm_wgrad_kernel<BM_, BN_, BK_, 2, 2, true, KG_, true>), grid, dim3(256 * KG_), lb, s, q);             \
  } while (0)
  if (pl.bm == 128) {
    if (pl.kg == 4) NNL_WGRAD_WINO_LAUNCH(128, 128, 16, 4);
    else if (pl.kg == 2) NNL_WGRAD_WINO_LAUNCH(128, 128, 16, 2);
    else NNL_WGRAD_WINO_LAUNCH(128, 128, 16, 1);
  } else if (pl.k_per_split % 32 == 0) {
    if (pl.kg == 4) NNL_WGRAD_WINO_LAUNCH(64, 64, 32, 4);
    else if (pl.kg == 2) NNL_WGRAD_WINO_LAUNCH(64, 64, 32, 2);
    else NNL_WGRAD_WINO_LAUNCH(64, 64, 32, 1);
  } else {
    NNL_WGRAD_WINO_LAUNCH(64, 64, 16, 1);
  }
#undef NNL_WGRAD_WINO_LAUNCH
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

// ---- the weight gradient in the 2-D Winograd F(2x2, 3x3) domain (igemm_wgrad2d.h, round 4) ----
// dW [K][3][3][C] from the slabs dU [splits][K][16][C]: 256 threads = 16 (k, c4) items x 16 positions; thread (item, position) sums its
// position over the slabs in index order (bitwise reproducible), the sixteen sums of an item meet in LDS, and nine of its threads
// apply dW = G^T dU G, G = (1,0,0 / .5,.5,.5 / .5,-.5,.5 / 0,0,1).
__global__ __launch_bounds__(256) void wino2d_wgrad_finish_kernel(const float* __restrict__ part, float* __restrict__ dw, long n_items, int C4,
                                                                   int splits, long slab4) {
  __shared__ f32x4 du[16][16];
  const int it = threadIdx.x >> 4, pos = threadIdx.x & 15;
  const long i = (long)blockIdx.x * 16 + it;                           // over (k, c4)
  const long c4 = i % C4, k = i / C4;
  f32x4 u = {0.f, 0.f, 0.f, 0.f};
  if (i < n_items) {
    const f32x4* src = reinterpret_cast<const f32x4*>(part) + (k * 16 + pos) * C4 + c4;
    for (int s0 = 0; s0 < splits; s0 += 8) {
      f32x4 v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) if (s0 + e < splits) v[e] = src[(long)(s0 + e) * slab4];
#pragma unroll
      for (int e = 0; e < 8; ++e) if (s0 + e < splits) u += v[e];
    }
  }
  du[it][pos] = u;
  __syncthreads();
  if (pos < 9 && i < n_items) {
    const int r = pos / 3, sx = pos - r * 3;
    // G^T column weights of filter tap t over positions 0..3: t = 0: (1, .5, .5, 0), t = 1: (0, .5, -.5, 0), t = 2: (0, .5, .5, 1)
    const float gr[4] = {r == 0 ? 1.f : 0.f, 0.5f, r == 1 ? -0.5f : 0.5f, r == 2 ? 1.f : 0.f};
    const float gs[4] = {sx == 0 ? 1.f : 0.f, 0.5f, sx == 1 ? -0.5f : 0.5f, sx == 2 ? 1.f : 0.f};
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      f32x4 row = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int b = 0; b < 4; ++b) row += gs[b] * du[it][a * 4 + b];
      acc += gr[a] * row;
    }
    reinterpret_cast<f32x4*>(dw)[(k * 9 + pos) * C4 + c4] = acc;
  }
}

// NNL_WGRAD_WINO2D: 0 never, 2 wherever it is legal, 1 (default) where it measured faster than the 1-D domain / the direct kernel
// (tools/bench_conv.py --ab NNL_WGRAD_WINO2D=0,2; profiles/r4_wgrad2d_*.log)
bool wgrad_wino2d_ok(const nnl_conv_geom_t* g) {
  const int mode = NNL_ENV_INT("NNL_WGRAD_WINO2D", 1);
  if (mode == 0 || NNL_ENV_INT("NNL_WGRAD_WINO", 1) == 0) return false;
  if (g->R != 3 || g->S != 3 || g->stride != 1 || g->pad != 1 || g->C % 64 != 0 || g->K % 4 != 0 || g->H < 2 || g->W < 2) return false;
  const long quads = (long)g->N * ((g->H + 1) / 2) * ((g->W + 1) / 2);
  if (quads < 512 || !wgrad_v2_ok((long)g->N * g->H * g->W * g->K, (long)g->N * g->H * g->W * g->C, quads)) return false;
  if (mode == 2) return true;
  return quads >= NNL_AB_INT("NNL_WGRAD_WINO2D_MIN_QUADS", 1024);
}

// Tile / wave-group choice of the 2-D domain, measured (tools/bench_conv.py --ab NNL_WGRAD_WINO_TILE / NNL_WGRAD_KG under NNL_WGRAD_WINO2D=2,
// profiles/r4_wgrad2d_*.log): the 64x64 tile with FOUR wave groups per workgroup (four neighbouring splits meet in LDS: a quarter of the
// slabs) wherever the plan has at least four splits (56^2 / 28^2 / 14^2 stages at 64 images: 0.105 / 0.101 / 0.100 -> 0.094 / 0.093 /
// 0.092 ms); the 128x128 tile where the 64x64 plan has fewer (7^2 stage: 0.127 -> 0.114 ms).  NNL_WGRAD_WINO2D_RULE=0: the generic planner.
WgradPlan plan_wgrad_wino2d(const nnl_conv_geom_t* g) {
  const long quads = (long)g->N * ((g->H + 1) / 2) * ((g->W + 1) / 2);
  WgradPlan pl = plan_wgrad(g->K, 16 * g->C, quads, g->C);
  if (NNL_AB_INT("NNL_WGRAD_WINO2D_RULE", 1) == 0 || NNL_AB_INT("NNL_WGRAD_WINO_TILE", -1) >= 0 || NNL_AB_INT("NNL_WGRAD_KG", -1) >= 0) return pl;
  auto with_tile = [&](int bt, int splits_target) {
    WgradPlan q{};
    q.bm = q.bn = bt; q.grid_m = (int)nnl_cdiv(g->K, bt); q.grid_n = (int)nnl_cdiv(16L * g->C, bt); q.kg = 1;
    const long tiles = (long)q.grid_m * q.grid_n;
    long sp = splits_target > 0 ? splits_target : (256L * 4 + tiles - 1) / tiles;            // ~4 workgroups per CU
    const long max_sp = quads / 256 > 0 ? quads / 256 : 1;
    if (sp > max_sp) sp = max_sp;
    if (sp < 1) sp = 1;
    const long k1 = nnl_cdiv(nnl_cdiv(quads, sp), 32) * 32;
    q.k_per_split = (int)k1; q.splits = (int)nnl_cdiv(quads, k1);
    return q;
  };
  const bool big = g->K >= 128 && g->C % 128 == 0;
  // big problems on wide layers (RetinaNet heads / FPN on P3: 16384 quads x 256 x 256): the 128x128 tile's operand reuse wins (0.430 -> 0.374 ms)
  if (big && (double)quads * g->K * g->C >= NNL_AB_INT("NNL_WGRAD_WINO2D_BIG_E6", 500) * 1e6) return pl.bm == 128 ? pl : with_tile(128, 0);
  WgradPlan p64 = pl.bm == 64 ? pl : with_tile(64, 0);
  if (p64.splits >= 4) {                                                // four neighbouring splits -> one workgroup of four wave groups
    const long sp = nnl_cdiv(p64.splits, 4);
    const long k1 = nnl_cdiv(nnl_cdiv(quads, sp), 128L) * 128L;          // each group's share stays a multiple of 32 quads
    if (k1 / 4 >= NNL_AB_INT("NNL_WGRAD_WINO2D_MIN_GROUP", 700)) {      // (shorter shares: prologue / group reduction dominate — 28^2 at 32 images: -23 %)
      p64.kg = 4; p64.k_per_split = (int)k1; p64.splits = (int)nnl_cdiv(quads, k1);
      return p64;
    }
    return pl;
  }
  if (big) return pl.bm == 128 ? pl : with_tile(128, 0);
  return p64;
}

int launch_wgrad_wino2d(const float* dy, const float* x, float* slabs, const nnl_conv_geom_t* g, const WgradPlan& pl, hipStream_t s) {
  IgemmWgrad2dParams q{};
  q.a = dy; q.b = x; q.y = slabs;
  q.a_bytes = (unsigned)((long)g->N * g->H * g->W * g->K * 4); q.b_bytes = (unsigned)((long)g->N * g->H * g->W * g->C * 4);
  q.H = g->H; q.W = g->W; q.C = g->C; q.H2 = (g->H + 1) / 2; q.W2 = (g->W + 1) / 2;
  q.Mc = g->K; q.Nc = 16 * g->C; q.Kp = (int)((long)g->N * q.H2 * q.W2);
  q.splits = pl.splits; q.k_per_split = pl.k_per_split; q.grid_m = pl.grid_m; q.grid_n = pl.grid_n;
  q.n_fast = 1;
  const dim3 grid(pl.grid_m * pl.grid_n * pl.splits);
  auto lds_bytes = [](int bt, int bk, int kg) { return (size_t)kg * 2 * bk * 2 * bt * sizeof(float); };
#define NNL_WGRAD2D_LAUNCH(BT_, BK_, KG_)                                                                                        \
  do {                                                                                                                           \
    const size_t lb = lds_bytes(BT_, BK_, KG_);                                                                                  \
    if (lb > 64 * 1024) {                                                                                                        \
      static bool attr_set = false;                                                                                              \
      if (!attr_set) {                                                                                                           \
        NNL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_wgrad2d_kernel<BT_, BK_, true, KG_>),             \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb));                                 \
        attr_set = true;                                                                                                         \
      }                                                                                                                          \
    }                                                                                                                            \
    hipLaunchKernelGGL((igemm_wgrad2d_kernel<BT_, BK_, true, KG_>), grid, dim3(256 * KG_), lb, s, q);                            \
  } while (0)
  if (pl.bm == 128) {
    if (pl.kg == 2) NNL_WGRAD2D_LAUNCH(128, 16, 2);
    else NNL_WGRAD2D_LAUNCH(128, 16, 1);
  } else if (pl.k_per_split % 32 == 0) {
    if (pl.kg == 4) NNL_WGRAD2D_LAUNCH(64, 32, 4);
    else if (pl.kg == 2) NNL_WGRAD2D_LAUNCH(64, 32, 2);
    else NNL_WGRAD2D_LAUNCH(64, 32, 1);
  } else {
    NNL_WGRAD2D_LAUNCH(64, 16, 1);
  }
#undef NNL_WGRAD2D_LAUNCH
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

}  // namespace

int nnl_internal_gemm_nt(const float* a, const float* b, float* y, const float* bias, const float* add, int M, int N,
                         int K, int relu, hipStream_t s) {
  if (M <= 0 || N <= 0 || K <= 0 || K % 4 != 0) return nnl_set_error(NNL_ERR_INVALID_ARG, "gemm_nt: bad sizes");
  IgemmRowkParams p{};
  p.a = a; p.b = b; p.y = y; p.bias = bias; p.add = add;
  p.N = M; p.H = 1; p.W = 1; p.C = K; p.P = 1; p.Q = 1; p.R = 1; p.S = 1; p.stride = 1; p.pad = 0;
  p.M = M; p.Nc = N; p.Kg = K; p.relu = relu;
  if (taps_ok((long)M * K, (long)N * K, K, 1)) {
    IgemmTapsParams q{};
    q.a = a; q.b = b; q.y = y; q.bias = bias; q.add = add;
    q.a_bytes = (unsigned)((long)M * K * 4); q.b_bytes = (unsigned)((long)N * K * 4);
    q.H = 1; q.W = 1; q.C = K; q.P = 1; q.Q = 1; q.in_stride = 1; q.ih0 = 0; q.iw0 = 0;
    q.OH = 1; q.OW = 1; q.out_stride = 1; q.oh0 = 0; q.ow0 = 0;
    q.M = M; q.Nc = N; q.b_row_stride = K; q.relu = relu; q.ntaps = 1;
    q.tap_dh[0] = 0; q.tap_dw[0] = 0; q.tap_aoff[0] = 0; q.tap_woff[0] = 0;
    q.tap_affine = 1; q.tap_R = 1; q.tap_S = 1; q.tap_dstep = 1;
    return dispatch_taps(q, s);
  }
  return dispatch_rowk<IGEMM_MODE_FWD>(p, s);
}

// y_slabs[s][M][N] = partial products over the s-th range of K (s < splits): for skinny GEMMs (M = batch) whose tile grid
// cannot fill the chip; the consumer adds the slabs in index order (deterministic).  K % 32 == 0.
int nnl_internal_gemm_nt_splitk(const float* a, const float* b, float* y_slabs, int M, int N, int K, int splits, hipStream_t s) {
  if (M <= 0 || N <= 0 || K <= 0 || K % 32 != 0 || splits < 1 || !taps_ok((long)M * K, (long)N * K, K, 1))
    return nnl_set_error(NNL_ERR_INVALID_ARG, "gemm_nt_splitk: bad sizes M=%d N=%d K=%d", M, N, K);
  IgemmTapsParams q{};
  q.a = a; q.b = b; q.y = y_slabs; q.bias = nullptr; q.add = nullptr;
  q.a_bytes = (unsigned)((long)M * K * 4); q.b_bytes = (unsigned)((long)N * K * 4);
  q.H = 1; q.W = 1; q.C = K; q.P = 1; q.Q = 1; q.in_stride = 1; q.ih0 = 0; q.iw0 = 0;
  q.OH = 1; q.OW = 1; q.out_stride = 1; q.oh0 = 0; q.ow0 = 0;
  q.M = M; q.Nc = N; q.b_row_stride = K; q.relu = 0; q.ntaps = 1;
  q.tap_dh[0] = 0; q.tap_dw[0] = 0; q.tap_aoff[0] = 0; q.tap_woff[0] = 0;
  q.tap_affine = 1; q.tap_R = 1; q.tap_S = 1; q.tap_dstep = 1;
  q.ksplit = splits; q.slab_stride = (long)M * N;
  return launch_taps<64, 64, 32>(q, s);
}

int nnl_internal_lstm_step(IgemmTapsParams q, int epi, hipStream_t s) {
  if ((epi != 1 && epi != 2) || q.M <= 0 || q.C % 32 != 0 || q.lstm.H <= 0 || q.lstm.counters == nullptr)
    return nnl_set_error(NNL_ERR_INVALID_ARG, "lstm_step: bad arguments");
  q.grid_m = (int)nnl_cdiv(q.M, 64);
  q.grid_n = epi == 1 ? (int)nnl_cdiv(q.lstm.H, 16) : (int)nnl_cdiv(q.lstm.H, 64);
  q.bal = 0;
  const dim3 grid(q.grid_m * q.grid_n, q.ksplit > 1 ? q.ksplit : 1), block(256);
  if (epi == 1)
    hipLaunchKernelGGL((igemm_taps_kernel<64, 64, 32, 2, 2, true, 1>), grid, block, 0, s, q);
  else
    hipLaunchKernelGGL((igemm_taps_kernel<64, 64, 32, 2, 2, true, 2>), grid, block, 0, s, q);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

size_t nnl_internal_gemm_tn_workspace_bytes(int Mc, int Nc, long Kp) {
  const WgradPlan pl = plan_wgrad(Mc, Nc, Kp);
  return pl.splits > 1 ? (size_t)pl.splits * Mc * Nc * sizeof(float) : 0;
}

int nnl_internal_gemm_tn(const float* a, const float* b, float* y, int Mc, int Nc, long Kp, void* ws, size_t ws_bytes,
                         hipStream_t s) {
  if (Mc <= 0 || Nc <= 0 || Kp <= 0 || Mc % 4 != 0 || Nc % 4 != 0) return nnl_set_error(NNL_ERR_INVALID_ARG, "gemm_tn: bad sizes");
  IgemmKmajorParams p{};
  p.a = a; p.b = b;
  p.N = (int)Kp; p.H = 1; p.W = 1; p.C = Nc; p.P = 1; p.Q = 1; p.R = 1; p.S = 1; p.stride = 1; p.pad = 0;
  p.Mc = Mc; p.Nc = Nc; p.Kp = Kp;
  const WgradPlan pl = plan_wgrad(Mc, Nc, Kp);
  p.grid_m = pl.grid_m; p.grid_n = pl.grid_n; p.splits = pl.splits; p.k_per_split = pl.k_per_split;
  const size_t need = nnl_internal_gemm_tn_workspace_bytes(Mc, Nc, Kp);
  if (need > 0 && (ws == nullptr || ws_bytes < need)) return nnl_set_error(NNL_ERR_WORKSPACE, "gemm_tn: workspace too small");
  p.y = pl.splits > 1 ? (float*)ws : y;
  if (wgrad_v2_ok(Kp * Mc, Kp * Nc, Kp)) {
    int st = launch_wgrad_v2(a, b, p.y, Kp * Mc, Kp * Nc, 1, 1, Nc, 1, 1, 1, 1, 1, 0, Mc, Nc, Kp, pl, s);
    if (st) return st;
  } else {
    const dim3 grid(pl.grid_m * pl.grid_n * pl.splits), block(256);
    if (pl.bm == 128 && pl.bn == 128)
      hipLaunchKernelGGL((igemm_kmajor_kernel<128, 128, 16, 2, 2>), grid, block, 0, s, p);
    else if (pl.bm == 128)
      hipLaunchKernelGGL((igemm_kmajor_kernel<128, 64, 16, 2, 2>), grid, block, 0, s, p);
    else if (pl.bn == 128)
      hipLaunchKernelGGL((igemm_kmajor_kernel<64, 128, 16, 2, 2>), grid, block, 0, s, p);
    else
      hipLaunchKernelGGL((igemm_kmajor_kernel<64, 64, 16, 2, 2>), grid, block, 0, s, p);
    NNL_CHECK_LAUNCH();
  }
  if (pl.splits > 1) {
    const long n4 = (long)Mc * Nc / 4;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)nnl_cdiv(n4, 32)), dim3(256), 0, s, (const float*)ws, y, n4, pl.splits);
    NNL_CHECK_LAUNCH();
  }
  return NNL_OK;
}

// Which kernel serves a 3x3 / stride 1 / pad 1 problem (in = [N][H][W][Cin], Nc output channels)?  NNL_CONV_WINO: 0 never the
// Winograd kernel, 2 always (where it applies), 1 (default) by predicted time.  Both planners share one cost model (us per k
// iteration of a CU's resident workgroups); measured against it (tools/bench_conv.py --ab NNL_CONV_WINO=0,2 at 8 / 16 / 32 / 64
// images, profiles/r3_wino_*.log) the Winograd iteration costs 1.18x (BK 16) / 1.08x (BK 32) the direct one (two A loads and an add per
// staged element, single accumulator chains: folded into nnl_wino_plan_time_us), both launches carry ~6 us the model does not see, and the filter transform adds a launch (3 us)
// plus 21 * Cin * Nc * 4 bytes of traffic.  A 3 % margin keeps coin-flip cases on the direct kernel.
// Returns 0 (direct kernel), 1 (1-D F(2,3), wino.hip) or 2 (2-D F(2x2,3x3), wino2.hip).  The 2-D kernel issues 2.25x fewer multiplies
// on a quarter of the direct kernel's tiles with four-pixel slabs and five loads per k step; its planner predicts absolute launch
// times from a model fitted to forced-schedule sweeps (wino2.hip: w2_cost) and it is taken where that prediction beats the 1-D
// kernel's — and only where the 1-D kernel already beats the direct one: the model was not fitted on the tiny grids the direct kernel
// keeps (profiles/r3_wino2d_*.log: -13 ... -20 % against the 1-D kernel per ResNet-34 stage at 64 images, -3 ... -9 % at 32).
// NNL_CONV_WINO2=0 turns it off; NNL_CONV_WINO=3 forces it wherever legal.  (Round 4's spatially staged 2-D kernel — raw input rows by
// LDS-DMA, 4x less traffic but issue-bound: 12.83 vs 12.59 ms per step, profiles/r4_wino2s_* — was removed in round 5.)
static int wino_mode(int N, int H, int W, int Cin, int Nc, int R, int S, int stride, int pad) {
  if (!nnl_wino_ok(N, H, W, Cin, Nc, R, S, stride, pad)) return 0;
  const int e = NNL_ENV_INT("NNL_CONV_WINO", 1);
  const bool two_ok = nnl_wino2_ok(N, H, W, Cin, Nc, R, S, stride, pad);
  if (e >= 3) return two_ok ? 2 : 1;
  if (e == 2) return 1;
  const double t_d = plan_balance((long)N * H * W, Nc, Cin, 9).t_us + 6.0;
  const double t_w = nnl_wino_plan_time_us(N, H, W, Cin, Nc) + 6.0 + 3.0 + 21.0 * Cin * Nc * 4.0 / 4.0e6;
  const bool two_on = two_ok && NNL_ENV_INT("NNL_CONV_WINO2", 1) != 0;
  if (!(t_w < 0.97 * t_d)) {
    // small grids (round 5): the 1-D kernel loses to the direct one, but the 2-D kernel's POSITION-SPLIT plan (wino2.hip: one position per
    // workgroup, 16-way natural split) may still beat both — its cost model is calibrated on exactly these shapes (8 - 32 images)
    if (two_on && nnl_wino2_plan_is_pos(N, H, W, Cin, Nc) && nnl_wino2_plan_time_us(N, H, W, Cin, Nc) < 0.9 * t_d) return 2;
    return 0;
  }
  // the 2-D kernel: 10 % predicted margin over the 1-D one (at 32 images they are within 5 % either way)
  if (two_on && nnl_wino2_plan_time_us(N, H, W, Cin, Nc) < 0.9 * t_w) return 2;
  return 1;
}
static size_t wino_mode_workspace(int mode, int N, int H, int W, int Cin, int Nc) {
  return mode == 2 ? nnl_wino2_workspace_bytes(N, H, W, Cin, Nc) : nnl_wino_workspace_bytes(N, H, W, Cin, Nc);
}
static int wino_mode_launch(int mode, const WinoProblem& q, void* ws, size_t ws_bytes, int* counters, long n_counters, hipStream_t s) {
  nnl_prof_exec_frac(mode >= 2 ? 1.0 / 2.25 : 1.0 / 1.5);                  // multiplies issued per algorithmic multiply (bench.py: roofline.executed_frac)
  return mode == 2 ? nnl_wino2_launch(q, ws, ws_bytes, counters, n_counters, s) : nnl_wino_launch(q, ws, ws_bytes, counters, n_counters, s);
}

// debug / tuning: the planners' predicted launch times (us) for a 3x3 / stride 1 / pad 1 problem: out[0] direct, out[1] Winograd 1-D, out[2] 2-D,
// out[3] = -1 (the slot of round 4's spatially staged kernel, removed); out must hold FOUR doubles
extern "C" int nnl_debug_conv_plan_times(int N, int H, int W, int Cin, int Nc, double* out) {
  out[0] = plan_balance((long)N * H * W, Nc, Cin, 9).t_us;
  out[1] = nnl_wino_plan_time_us(N, H, W, Cin, Nc);
  out[2] = nnl_wino2_plan_time_us(N, H, W, Cin, Nc);
  out[3] = -1.0;
  return wino_mode(N, H, W, Cin, Nc, 3, 3, 1, 1);
}

// (the larger of the direct kernel's slabs and the Winograd path's transformed filter + slabs: either may run, see wino.h)
extern "C" size_t nnl_conv2d_fwd_workspace_bytes(const nnl_conv_geom_t* g) {
  if (!g || check_geom(g, "conv2d_fwd_workspace_bytes")) return 0;
  const long a_elems = (long)g->N * g->H * g->W * g->C, b_elems = (long)g->K * g->R * g->S * g->C;
  const int tk = taps_kind(a_elems, b_elems, g->C, g->R * g->S);
  if (!tk) return 0;
  if (tk == 2) return balance_workspace_bytes((long)g->N * g->P * g->Q, g->K, (int)nnl_cdiv(g->C, 32) * 32, 1);
  size_t b = balance_workspace_bytes((long)g->N * g->P * g->Q, g->K, g->C, g->R * g->S);
  if (const int wm = wino_mode(g->N, g->H, g->W, g->C, g->K, g->R, g->S, g->stride, g->pad)) {
    const size_t wb = wino_mode_workspace(wm, g->N, g->H, g->W, g->C, g->K);
    if (wb > b) b = wb;
  }
  return b;
}

extern "C" size_t nnl_conv2d_dgrad_workspace_bytes(const nnl_conv_geom_t* g) {
  if (!g || check_geom(g, "conv2d_dgrad_workspace_bytes") || g->stride != 1) return 0;
  const long a_elems = (long)g->N * g->P * g->Q * g->K, b_elems = (long)g->C * g->R * g->S * g->K;
  const int tk = taps_kind(a_elems, b_elems, g->K, g->R * g->S);
  if (!tk) return 0;
  if (tk == 2) return balance_workspace_bytes((long)g->N * g->H * g->W, g->C, (int)nnl_cdiv(g->K, 32) * 32, 1);
  size_t b = balance_workspace_bytes((long)g->N * g->H * g->W, g->C, g->K, g->R * g->S);
  if (const int wm = wino_mode(g->N, g->P, g->Q, g->K, g->C, g->R, g->S, g->stride, g->pad)) {
    const size_t wb = wino_mode_workspace(wm, g->N, g->P, g->Q, g->K, g->C);
    if (wb > b) b = wb;
  }
  return b;
}

extern "C" int64_t nnl_conv2d_tile_counters(void) { return kTileCounters; }

extern "C" int nnl_conv2d_wino_preferred(const nnl_conv_geom_t* g, int dgrad) {
  if (!g) return 0;
  return dgrad ? wino_mode(g->N, g->P, g->Q, g->K, g->C, g->R, g->S, g->stride, g->pad)
               : wino_mode(g->N, g->H, g->W, g->C, g->K, g->R, g->S, g->stride, g->pad);
}

extern "C" int nnl_conv2d_fwd(const float* x, const float* w, const float* bias, float* y, const nnl_conv_geom_t* g,
                              int relu, void* workspace, size_t workspace_bytes, int32_t* tile_counters, float* bn_partials,
                              const float* bn_pivot, int32_t* bn_rows, void* stream) {
  return nnl_conv2d_fwd_pre(x, w, bias, y, g, relu, workspace, workspace_bytes, tile_counters, bn_partials, bn_pivot, bn_rows, nullptr, stream);
}

extern "C" int nnl_conv2d_fwd_pre(const float* x, const float* w, const float* bias, float* y, const nnl_conv_geom_t* g,
                                  int relu, void* workspace, size_t workspace_bytes, int32_t* tile_counters, float* bn_partials,
                                  const float* bn_pivot, int32_t* bn_rows, const float* u, void* stream) {
  if (bn_rows) *bn_rows = 0;
  int st = check_geom(g, "conv2d_fwd");
  if (st) return st;
  NNL_CHECK_ARG(x && w && y, "conv2d_fwd: null pointer");
  NNL_CHECK_ARG(relu >= 0 && relu <= 2, "conv2d_fwd: relu (activation) must be 0 none, 1 ReLU or 2 sigmoid");
  hipStream_t s = (hipStream_t)stream;
  IgemmRowkParams p{};
  p.a = x; p.b = w; p.y = y; p.bias = bias;
  p.N = g->N; p.H = g->H; p.W = g->W; p.C = g->C; p.P = g->P; p.Q = g->Q;
  p.R = g->R; p.S = g->S; p.stride = g->stride; p.pad = g->pad;
  p.M = g->N * g->P * g->Q; p.Nc = g->K; p.Kg = g->R * g->S * g->C; p.relu = relu;
  NnlProfScope prof(NNL_PROF_CONV_FWD, s, 2.0 * p.M * (double)p.Nc * p.Kg);
  const long a_elems = (long)g->N * g->H * g->W * g->C, b_elems = (long)g->K * g->R * g->S * g->C;
  const int wmode = (relu != 2 && workspace != nullptr) ? wino_mode(g->N, g->H, g->W, g->C, g->K, g->R, g->S, g->stride, g->pad) : 0;
  if (wmode && workspace_bytes >= wino_mode_workspace(wmode, g->N, g->H, g->W, g->C, g->K)) {
    // 3x3 / stride 1 / pad 1: fused Winograd — 1-D F(2,3) (wino.hip, 1.5x fewer MFMA k-steps per output) or 2-D F(2x2,3x3) (wino2.hip, 2.25x)
    WinoProblem wq{};
    wq.in = x; wq.filt = w; wq.out = y; wq.bias = bias; wq.add = nullptr;
    wq.N = g->N; wq.H = g->H; wq.W = g->W; wq.Cin = g->C; wq.Nc = g->K; wq.relu = relu; wq.flip = 0;
    const bool stats = bn_partials && bn_pivot && bn_rows;
    wq.bn_part = stats ? bn_partials : nullptr; wq.bn_pivot = bn_pivot;
    wq.u_pre = u;
    st = wino_mode_launch(wmode, wq, workspace, workspace_bytes, tile_counters, kTileCounters, s);
    if (st == NNL_OK && stats) *bn_rows = wmode >= 2 ? nnl_wino2_bn_rows(g->N, g->H, g->W) : nnl_wino_bn_rows(g->N, g->H, g->W);
    return st;
  }
  if (const int tk = taps_kind(a_elems, b_elems, g->C, g->R * g->S)) {
    IgemmTapsParams q{};
    q.ktail = tk == 2;
    q.a = x; q.b = w; q.y = y; q.bias = bias; q.add = nullptr;
    q.a_bytes = (unsigned)(a_elems * 4); q.b_bytes = (unsigned)(b_elems * 4);
    q.H = g->H; q.W = g->W; q.C = g->C; q.P = g->P; q.Q = g->Q;
    q.in_stride = g->stride; q.ih0 = -g->pad; q.iw0 = -g->pad;
    q.OH = g->P; q.OW = g->Q; q.out_stride = 1; q.oh0 = 0; q.ow0 = 0;
    q.M = p.M; q.Nc = g->K; q.b_row_stride = g->R * g->S * g->C; q.relu = relu;
    q.ntaps = g->R * g->S;
    for (int r = 0; r < g->R; ++r)
      for (int ss = 0; ss < g->S; ++ss) {
        const int t = r * g->S + ss;
        q.tap_dh[t] = (signed char)r; q.tap_dw[t] = (signed char)ss;
        q.tap_aoff[t] = (r * g->W + ss) * g->C; q.tap_woff[t] = t * g->C;
      }
    q.tap_affine = NNL_AB_INT("NNL_IGEMM_AFFINE", 1); q.tap_R = g->R; q.tap_S = g->S; q.tap_dh0 = 0; q.tap_dw0 = 0; q.tap_dstep = 1;
    q.bn_part = (bn_partials && bn_pivot && bn_rows) ? bn_partials : nullptr; q.bn_pivot = bn_pivot;
    return dispatch_taps(q, s, workspace, workspace_bytes, tile_counters, bn_rows);
  }
  if (relu == 2) return nnl_set_error(NNL_ERR_UNSUPPORTED, "conv2d_fwd: the sigmoid epilogue needs C %% 16 == 0 (C=%d)", g->C);
  return dispatch_rowk<IGEMM_MODE_FWD>(p, s);
}

/* y = conv(x, w) + bias + upsample_nearest_x2(small): the FPN top-down merge `P5_upsampled + P4_1(C4)` (reference retinanet.py:
 * 131-141) in the lateral convolution's epilogue — the upsampled tensor is never materialised and the add is not a separate pass. */
extern "C" int nnl_conv2d_fwd_add_up2(const float* x, const float* w, const float* bias, const float* small, float* y,
                                      const nnl_conv_geom_t* g, void* stream) {
  int st = check_geom(g, "conv2d_fwd_add_up2");
  if (st) return st;
  NNL_CHECK_ARG(x && w && y && small, "conv2d_fwd_add_up2: null pointer");
  NNL_CHECK_ARG(g->P % 2 == 0 && g->Q % 2 == 0, "conv2d_fwd_add_up2: the output (%d x %d) must be twice the small map", g->P, g->Q);
  const long a_elems = (long)g->N * g->H * g->W * g->C, b_elems = (long)g->K * g->R * g->S * g->C;
  if (!taps_ok(a_elems, b_elems, g->C, g->R * g->S))
    return nnl_set_error(NNL_ERR_UNSUPPORTED, "conv2d_fwd_add_up2: needs C %% 16 == 0 (C=%d)", g->C);
  hipStream_t s = (hipStream_t)stream;
  const long M = (long)g->N * g->P * g->Q;
  NnlProfScope prof(NNL_PROF_CONV_FWD, s, 2.0 * M * (double)g->K * g->R * g->S * g->C);
  IgemmTapsParams q{};
  q.a = x; q.b = w; q.y = y; q.bias = bias; q.add = small; q.add_up2 = 1;
  q.a_bytes = (unsigned)(a_elems * 4); q.b_bytes = (unsigned)(b_elems * 4);
  q.H = g->H; q.W = g->W; q.C = g->C; q.P = g->P; q.Q = g->Q;
  q.in_stride = g->stride; q.ih0 = -g->pad; q.iw0 = -g->pad;
  q.OH = g->P; q.OW = g->Q; q.out_stride = 1; q.oh0 = 0; q.ow0 = 0;
  q.M = (int)M; q.Nc = g->K; q.b_row_stride = g->R * g->S * g->C; q.relu = 0;
  q.ntaps = g->R * g->S;
  for (int r = 0; r < g->R; ++r)
    for (int ss = 0; ss < g->S; ++ss) {
      const int t = r * g->S + ss;
      q.tap_dh[t] = (signed char)r; q.tap_dw[t] = (signed char)ss;
      q.tap_aoff[t] = (r * g->W + ss) * g->C; q.tap_woff[t] = t * g->C;
    }
  q.tap_affine = NNL_AB_INT("NNL_IGEMM_AFFINE", 1); q.tap_R = g->R; q.tap_S = g->S; q.tap_dh0 = 0; q.tap_dw0 = 0; q.tap_dstep = 1;
  return dispatch_taps(q, s);            // no workspace: the plain grid (the split-tile fix-up path reads a same-shape addend only)
}

extern "C" int nnl_conv2d_weight_transpose(const float* w, float* wt, int K, int R, int S, int C, void* stream) {
  NNL_CHECK_ARG(w && wt && K > 0 && R > 0 && S > 0 && C > 0, "conv2d_weight_transpose: bad argument");
  hipStream_t s = (hipStream_t)stream;
  NnlProfScope prof(NNL_PROF_ELEMENTWISE, s, 8.0 * K * R * S * C);
  hipLaunchKernelGGL(weight_transpose_kernel, dim3((unsigned)nnl_cdiv(C, 32), (unsigned)nnl_cdiv(K, 32), R * S), dim3(256), 0, s, w,
                     wt, K, R * S, C);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

extern "C" int nnl_conv2d_weight_transpose_multi(const nnl_wt_desc_t* desc, const int32_t* tile_tensor, int64_t n_tiles,
                                                 double total_elems, void* stream) {
  NNL_CHECK_ARG(desc && tile_tensor && n_tiles > 0 && n_tiles < (1L << 31), "conv2d_weight_transpose_multi: bad argument");
  hipStream_t s = (hipStream_t)stream;
  NnlProfScope prof(NNL_PROF_ELEMENTWISE, s, 8.0 * total_elems);
  hipLaunchKernelGGL(weight_transpose_multi_kernel, dim3((unsigned)n_tiles), dim3(256), 0, s, desc, tile_tensor);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

extern "C" int nnl_conv2d_dgrad(const float* dy, const float* wt, float* dx, const nnl_conv_geom_t* g, const float* addend,
                                void* workspace, size_t workspace_bytes, int32_t* tile_counters, void* stream) {
  return nnl_conv2d_dgrad_pre(dy, wt, dx, g, addend, workspace, workspace_bytes, tile_counters, nullptr, stream);
}

extern "C" int nnl_conv2d_dgrad_pre(const float* dy, const float* wt, float* dx, const nnl_conv_geom_t* g, const float* addend,
                                    void* workspace, size_t workspace_bytes, int32_t* tile_counters, const float* u, void* stream) {
  int st = check_geom(g, "conv2d_dgrad");
  if (st) return st;
  NNL_CHECK_ARG(dy && wt && dx, "conv2d_dgrad: null pointer");
  NNL_CHECK_ARG(g->K % 4 == 0, "conv2d_dgrad: K=%d must be a multiple of 4", g->K);
  {
    const bool taps = taps_ok((long)g->N * g->P * g->Q * g->K, (long)g->C * g->R * g->S * g->K, g->K, g->R * g->S);
    // stride 2: the addend is applied by the class that owns each dx pixel, so every output-parity class needs a tap (3x3, pad 1)
    const bool s2_full = g->stride == 2 && g->R == 3 && g->S == 3 && g->pad == 1;
    if (addend != nullptr && !(taps && (g->stride == 1 || s2_full)))
      return nnl_set_error(NNL_ERR_UNSUPPORTED, "conv2d_dgrad: the fused addend needs K %% 16 == 0 and stride 1 (or a 3x3 / pad 1 filter at stride 2)");
  }
  hipStream_t s = (hipStream_t)stream;
  IgemmRowkParams p{};
  p.a = dy; p.b = wt; p.y = dx; p.bias = nullptr;
  p.N = g->N; p.H = g->P; p.W = g->Q; p.C = g->K;          // A source = dy [N][P][Q][K]
  p.P = g->H; p.Q = g->W;                                  // GEMM rows enumerate dx pixels
  p.R = g->R; p.S = g->S; p.stride = g->stride; p.pad = g->pad;
  p.M = g->N * g->H * g->W; p.Nc = g->C; p.Kg = g->R * g->S * g->K; p.relu = 0;
  NnlProfScope prof(NNL_PROF_CONV_DGRAD, s, 2.0 * g->N * (double)g->P * g->Q * g->K * g->R * g->S * g->C);
  const long a_elems = (long)g->N * g->P * g->Q * g->K, b_elems = (long)g->C * g->R * g->S * g->K;
  const int wmode = workspace != nullptr ? wino_mode(g->N, g->P, g->Q, g->K, g->C, g->R, g->S, g->stride, g->pad) : 0;
  if (wmode && workspace_bytes >= wino_mode_workspace(wmode, g->N, g->P, g->Q, g->K, g->C)) {
    // stride-1 dgrad of a 3x3 / pad 1 filter = the same convolution over dy with the flipped, transposed filter (wino.hip)
    WinoProblem wq{};
    wq.in = dy; wq.filt = wt; wq.out = dx; wq.bias = nullptr; wq.add = addend;
    wq.N = g->N; wq.H = g->P; wq.W = g->Q; wq.Cin = g->K; wq.Nc = g->C; wq.relu = 0; wq.flip = 1;
    wq.u_pre = u;
    return wino_mode_launch(wmode, wq, workspace, workspace_bytes, tile_counters, kTileCounters, s);
  }
  const int tk = taps_kind(a_elems, b_elems, g->K, g->R * g->S);
  if (tk && (g->stride == 1 || g->stride == 2)) {
    IgemmTapsParams q{};
    q.ktail = tk == 2;
    q.a = dy; q.b = wt; q.y = dx; q.bias = nullptr; q.add = addend;
    q.a_bytes = (unsigned)(a_elems * 4); q.b_bytes = (unsigned)(b_elems * 4);
    q.H = g->P; q.W = g->Q; q.C = g->K;                     // gathered tensor = dy [N][P][Q][K]
    q.in_stride = 1; q.ih0 = 0; q.iw0 = 0;
    q.OH = g->H; q.OW = g->W; q.Nc = g->C; q.b_row_stride = g->R * g->S * g->K; q.relu = 0;
    const int st2 = g->stride;
    // one launch per output-parity class (a single class when stride == 1): dx pixel (st*hh + ph, st*ww + pw) receives
    // exactly the taps r with (ph + pad - r) % st == 0, from dy pixel hh + (ph + pad - r)/st
    struct Cls { int ph, pw, P, Q, nt; signed char dh[IGEMM_MAX_TAPS], dw[IGEMM_MAX_TAPS]; int aoff[IGEMM_MAX_TAPS], woff[IGEMM_MAX_TAPS]; };
    Cls cls[4];
    int ncls = 0;
    bool need_zero = false;
    for (int ph = 0; ph < st2; ++ph)
      for (int pw = 0; pw < st2; ++pw) {
        Cls c{};
        c.ph = ph; c.pw = pw;
        c.P = (g->H - ph + st2 - 1) / st2; c.Q = (g->W - pw + st2 - 1) / st2;
        for (int r = 0; r < g->R; ++r) {
          if ((ph + g->pad - r) % st2 != 0) continue;
          for (int ss = 0; ss < g->S; ++ss) {
            if ((pw + g->pad - ss) % st2 != 0) continue;
            const int dh = (ph + g->pad - r) / st2, dw = (pw + g->pad - ss) / st2;
            c.dh[c.nt] = (signed char)dh; c.dw[c.nt] = (signed char)dw;
            c.aoff[c.nt] = (dh * g->Q + dw) * g->K; c.woff[c.nt] = (r * g->S + ss) * g->K;
            ++c.nt;
          }
        }
        if (c.P <= 0 || c.Q <= 0) continue;
        if (c.nt == 0) { need_zero = true; continue; }      // no filter tap reaches this parity class: its dx pixels are zero
        cls[ncls++] = c;
      }
    if (need_zero) NNL_CHECK_HIP(hipMemsetAsync(dx, 0, sizeof(float) * g->N * g->H * g->W * g->C, s));
    auto fill = [&](IgemmTapsParams& c, const Cls& k, int at) {
      for (int t = 0; t < k.nt; ++t) {
        c.tap_dh[at + t] = k.dh[t]; c.tap_dw[at + t] = k.dw[t]; c.tap_aoff[at + t] = k.aoff[t]; c.tap_woff[at + t] = k.woff[t];
      }
    };
    // all classes in ONE launch when they have the same row count (even H, W): longest classes first
    bool merged = ncls > 1 && NNL_AB_INT("NNL_DGRAD_MERGE", 1) != 0;
    for (int i = 1; i < ncls && merged; ++i) merged = cls[i].P == cls[0].P && cls[i].Q == cls[0].Q;
    if (merged) {
      for (int i = 1; i < ncls; ++i)                        // insertion sort by decreasing tap count (<= 4 entries)
        for (int j = i; j > 0 && cls[j].nt > cls[j - 1].nt; --j) { const Cls t = cls[j]; cls[j] = cls[j - 1]; cls[j - 1] = t; }
      IgemmTapsParams c = q;
      c.P = cls[0].P; c.Q = cls[0].Q; c.M = g->N * c.P * c.Q;
      c.out_stride = st2; c.oh0 = cls[0].ph; c.ow0 = cls[0].pw;
      c.ncls = ncls;
      int at = 0;
      for (int i = 0; i < ncls; ++i) {
        c.cls_tap0[i] = at; c.cls_ntaps[i] = cls[i].nt; c.cls_oh0[i] = cls[i].ph; c.cls_ow0[i] = cls[i].pw;
        fill(c, cls[i], at);
        at += cls[i].nt;
      }
      c.ntaps = cls[0].nt;
      return dispatch_taps(c, s, nullptr, 0, tile_counters);
    }
    for (int i = 0; i < ncls; ++i) {
      IgemmTapsParams c = q;
      c.P = cls[i].P; c.Q = cls[i].Q; c.M = g->N * c.P * c.Q;
      c.out_stride = st2; c.oh0 = cls[i].ph; c.ow0 = cls[i].pw;
      fill(c, cls[i], 0);
      c.ntaps = cls[i].nt;
      if (st2 == 1 && cls[i].nt == g->R * g->S) {          // the full raster, r-major: (dh, dw) = (pad - r, pad - s), woff = t*K
        c.tap_affine = NNL_AB_INT("NNL_IGEMM_AFFINE", 1); c.tap_R = g->R; c.tap_S = g->S; c.tap_dh0 = g->pad; c.tap_dw0 = g->pad; c.tap_dstep = -1;
      }
      int st = dispatch_taps(c, s, st2 == 1 ? workspace : nullptr, workspace_bytes, tile_counters);
      if (st) return st;
    }
    return NNL_OK;
  }
  return dispatch_rowk<IGEMM_MODE_DGRAD>(p, s);
}

extern "C" size_t nnl_conv2d_wgrad_workspace_bytes(const nnl_conv_geom_t* g) {
  if (!g || g->K <= 0 || g->C <= 0) return 0;
  if (wgrad_wino2d_ok(g))                                   // the 2-D Winograd-domain slabs [splits][K][16*C] (always: dU is folded to dW from them)
    return (size_t)plan_wgrad_wino2d(g).splits * g->K * 16 * g->C * sizeof(float);
  if (wgrad_wino_ok(g))                                     // the Winograd-domain slabs [splits][K][12*C] (always: dU is folded to dW from them)
    return (size_t)plan_wgrad_wino(g).splits * g->K * 12 * g->C * sizeof(float);
  const WgradPlan pl = plan_wgrad(g->K, g->R * g->S * g->C, (long)g->N * g->P * g->Q);
  return pl.splits > 1 ? (size_t)pl.splits * g->K * g->R * g->S * g->C * sizeof(float) : 0;
}

extern "C" int nnl_conv2d_wgrad(const float* x, const float* dy, float* dw, const nnl_conv_geom_t* g, void* workspace,
                                size_t workspace_bytes, void* stream) {
  int st = check_geom(g, "conv2d_wgrad");
  if (st) return st;
  NNL_CHECK_ARG(x && dy && dw, "conv2d_wgrad: null pointer");
  NNL_CHECK_ARG(g->K % 4 == 0, "conv2d_wgrad: K=%d must be a multiple of 4", g->K);
  hipStream_t s = (hipStream_t)stream;
  IgemmKmajorParams p{};
  p.a = dy; p.b = x;
  p.N = g->N; p.H = g->H; p.W = g->W; p.C = g->C; p.P = g->P; p.Q = g->Q;
  p.R = g->R; p.S = g->S; p.stride = g->stride; p.pad = g->pad;
  p.Mc = g->K; p.Nc = g->R * g->S * g->C; p.Kp = (long)g->N * g->P * g->Q;
  const WgradPlan pl = plan_wgrad(p.Mc, p.Nc, p.Kp);
  p.grid_m = pl.grid_m; p.grid_n = pl.grid_n; p.splits = pl.splits; p.k_per_split = pl.k_per_split;
  const size_t need = nnl_conv2d_wgrad_workspace_bytes(g);
  if (need > 0 && (workspace == nullptr || workspace_bytes < need))
    return nnl_set_error(NNL_ERR_WORKSPACE, "conv2d_wgrad: workspace %zu B < required %zu B", workspace_bytes, need);
  p.y = pl.splits > 1 ? (float*)workspace : dw;
  NnlProfScope prof(NNL_PROF_CONV_WGRAD, s, 2.0 * p.Kp * (double)p.Mc * p.Nc);
  if (wgrad_wino2d_ok(g)) {                                 // 2-D Winograd-domain weight gradient (igemm_wgrad2d.h) + the fold-back reduce
    const WgradPlan wp = plan_wgrad_wino2d(g);
    nnl_prof_exec_frac(1.0 / 2.25);
    int st2 = launch_wgrad_wino2d(dy, x, (float*)workspace, g, wp, s);
    if (st2) return st2;
    const long n_items = (long)g->K * (g->C / 4);
    hipLaunchKernelGGL(wino2d_wgrad_finish_kernel, dim3((unsigned)nnl_cdiv(n_items, 16L)), dim3(256), 0, s, (const float*)workspace, dw, n_items,
                       g->C / 4, wp.splits, (long)g->K * 16 * (g->C / 4));
    NNL_CHECK_LAUNCH();
    return NNL_OK;
  }
  if (wgrad_wino_ok(g)) {                                   // Winograd-domain weight gradient (igemm_wgrad.h, WINO) + the fold-back reduce
    const WgradPlan wp = plan_wgrad_wino(g);
    nnl_prof_exec_frac(1.0 / 1.5);
    int st2 = launch_wgrad_wino(dy, x, (float*)workspace, g, wp, s);
    if (st2) return st2;
    const long n4 = (long)g->K * 3 * (g->C / 4);
    hipLaunchKernelGGL(wino_wgrad_finish_kernel, dim3((unsigned)nnl_cdiv(n4, 32L)), dim3(256), 0, s, (const float*)workspace, dw, n4,
                       g->C / 4, wp.splits, (long)g->K * 12 * (g->C / 4));
    NNL_CHECK_LAUNCH();
    return NNL_OK;
  }
  const long a_elems = p.Kp * g->K, b_elems = (long)g->N * g->H * g->W * g->C;
  if (wgrad_v2_ok(a_elems, b_elems, p.Kp)) {
    int st2 = launch_wgrad_v2(dy, x, p.y, a_elems, b_elems, g->H, g->W, g->C, g->P, g->Q, g->R, g->S, g->stride, g->pad, p.Mc,
                              p.Nc, p.Kp, pl, s);
    if (st2) return st2;
  } else {
    const dim3 grid(pl.grid_m * pl.grid_n * pl.splits), block(256);
    if (pl.bm == 128 && pl.bn == 128)
      hipLaunchKernelGGL((igemm_kmajor_kernel<128, 128, 16, 2, 2>), grid, block, 0, s, p);
    else if (pl.bm == 128)
      hipLaunchKernelGGL((igemm_kmajor_kernel<128, 64, 16, 2, 2>), grid, block, 0, s, p);
    else if (pl.bn == 128)
      hipLaunchKernelGGL((igemm_kmajor_kernel<64, 128, 16, 2, 2>), grid, block, 0, s, p);
    else
      hipLaunchKernelGGL((igemm_kmajor_kernel<64, 64, 16, 2, 2>), grid, block, 0, s, p);
    NNL_CHECK_LAUNCH();
  }
  if (pl.splits > 1) {
    const long n4 = (long)p.Mc * p.Nc / 4;                 // Nc = R*S*C with C % 4 == 0
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)nnl_cdiv(n4, 32)), dim3(256), 0, s, (const float*)workspace, dw, n4, pl.splits);
    NNL_CHECK_LAUNCH();
  }
  return NNL_OK;
}

extern "C" size_t nnl_colsum_workspace_bytes(int64_t rows, int64_t cols) {
  if (rows <= 0 || cols <= 0) return 0;
  return (size_t)colsum_chunks(rows) * cols * sizeof(float);
}

extern "C" int nnl_act_gate_colsum(const float* dy, const float* y, float* g, float* colsum, int64_t rows, int64_t cols, int act,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  NNL_CHECK_ARG(dy && y && g && rows > 0 && cols > 0 && cols < (1L << 30), "act_gate_colsum: bad argument");
  NNL_CHECK_ARG(act == 1 || act == 2, "act_gate_colsum: act must be 1 (ReLU) or 2 (sigmoid)");
  if (colsum != nullptr && (workspace == nullptr || workspace_bytes < nnl_colsum_workspace_bytes(rows, cols)))
    return nnl_set_error(NNL_ERR_WORKSPACE, "act_gate_colsum: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  NnlProfScope prof(NNL_PROF_ELEMENTWISE, s, 12.0 * rows * cols);
  const int nchunk = colsum_chunks(rows);
  const long rpc = nnl_cdiv(nnl_cdiv(rows, nchunk), 4) * 4;
  const dim3 grid((unsigned)nnl_cdiv(cols, 64), nchunk);
  float* part = colsum ? (float*)workspace : nullptr;
  if (act == 1)
    hipLaunchKernelGGL(act_gate_colsum_kernel<1>, grid, dim3(256), 0, s, dy, y, g, part, (long)rows, (int)cols, rpc);
  else
    hipLaunchKernelGGL(act_gate_colsum_kernel<2>, grid, dim3(256), 0, s, dy, y, g, part, (long)rows, (int)cols, rpc);
  NNL_CHECK_LAUNCH();
  if (colsum != nullptr) {
    hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)nnl_cdiv(cols, 16)), dim3(256), 0, s, (const float*)workspace, colsum,
                       (int)cols, nchunk);
    NNL_CHECK_LAUNCH();
  }
  return NNL_OK;
}

extern "C" int nnl_relu_gate_colsum(const float* dy, const float* y, float* g, float* colsum, int64_t rows, int64_t cols,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  return nnl_act_gate_colsum(dy, y, g, colsum, rows, cols, 1, workspace, workspace_bytes, stream);
}

// out = src[0] + src[1] + ... + src[n - 1] (in that order: bitwise reproducible), n <= 8 dense tensors of `numel` floats
namespace {
struct SumN { const float* src[8]; };
__global__ __launch_bounds__(256) void sum_tensors_kernel(SumN p, int n, float* __restrict__ out, long numel) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < numel) {
    f32x4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) if (k < n) v[k] = *reinterpret_cast<const f32x4*>(p.src[k] + i);
    f32x4 acc = v[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) if (k < n) acc += v[k];
    *reinterpret_cast<f32x4*>(out + i) = acc;
  } else {
    for (long j = i; j < numel; ++j) {
      float acc = p.src[0][j];
      for (int k = 1; k < n; ++k) acc += p.src[k][j];
      out[j] = acc;
    }
  }
}
}  // namespace

extern "C" int nnl_sum_tensors(const float* const* src, int n, float* out, int64_t numel, void* stream) {
  NNL_CHECK_ARG(src && out && n >= 1 && n <= 8 && numel > 0, "sum_tensors: 1 to 8 tensors of numel > 0 elements");
  SumN p{};
  for (int k = 0; k < n; ++k) {
    NNL_CHECK_ARG(src[k] != nullptr && ((uintptr_t)src[k] & 15) == 0, "sum_tensors: null or unaligned source");
    p.src[k] = src[k];
  }
  NNL_CHECK_ARG(((uintptr_t)out & 15) == 0, "sum_tensors: unaligned destination");
  hipStream_t s = (hipStream_t)stream;
  NnlProfScope prof(NNL_PROF_ELEMENTWISE, s, 4.0 * (n + 1) * numel);
  hipLaunchKernelGGL(sum_tensors_kernel, dim3((unsigned)nnl_cdiv(nnl_cdiv(numel, 4L), 256L)), dim3(256), 0, s, p, n, out, (long)numel);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

extern "C" int nnl_colsum(const float* a, float* out, int64_t rows, int64_t cols, void* workspace, size_t workspace_bytes,
                          void* stream) {
  NNL_CHECK_ARG(a && out && rows > 0 && cols > 0 && cols < (1L << 30), "colsum: bad argument");
  if (workspace == nullptr || workspace_bytes < nnl_colsum_workspace_bytes(rows, cols))
    return nnl_set_error(NNL_ERR_WORKSPACE, "colsum: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  NnlProfScope prof(NNL_PROF_ELEMENTWISE, s, 4.0 * rows * cols);
  const int nchunk = colsum_chunks(rows);
  const long rpc = nnl_cdiv(nnl_cdiv(rows, nchunk), 4) * 4;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)nnl_cdiv(cols, 64), nchunk), dim3(256), 0, s, a, (float*)workspace,
                     (long)rows, (int)cols, rpc);
  NNL_CHECK_LAUNCH();
  hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)nnl_cdiv(cols, 16)), dim3(256), 0, s, (const float*)workspace, out, (int)cols,
                     nchunk);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}
