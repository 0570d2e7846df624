// K5 — persistent FORWARD recurrence of one LSTM layer on a 2-D partition of W_hh with the barrier-free granule exchange of
// lstm_bptt2.hip (round 4): one cooperative launch runs all T timesteps of WeightDropLSTM1.forward -> nn.LSTM (cuDNN's persistent
// RNN in the reference; Applications/Text.py:495-513, :535-551).  The first persistent forward (lstm_persist.hip) partitions only
// the gate columns: every workgroup streams ALL of h_{t-1} (294 KB) per step and the grid meets at a barrier — 17-19 us per step at
// H = 1150, 12 of them the k loop at the L2 -> CU bandwidth.
//
// Per step:  pre[b][g H + j] = gx_t[b][g H + j] + sum_k h_{t-1}[b][k] W_hh[g H + j][k]  (gates g = i, f, g, o; k < H), then the cell.
//
// Partition.  KG x NG workgroups (<= 256, co-resident).  Column group ng owns the units [ng Us, ng Us + Us) with all four gate
// columns (local column = gate * Us + unit); workgroup (kg, ng) keeps the [Ks = Kp/KG] x [4 Us] block of W_hh in LDS for all
// timesteps (H = 1150: 4 x 64 workgroups, 288 x 72 -> 92 KB).  A step has two phases, both per STREAM (16 batch rows; four
// independent streams, two waves each — see lstm_bptt2.hip):
//   A  partial[kg][b][col] over the workgroup's k slice: v_mfma_f32_16x16x4_f32, the A operand = the h_{t-1} granules of the
//      stream's rows (two 16-B agent-scope loads per 16 k), B from the [k/4][col][4] LDS image; the two k halves of a stream meet
//      in LDS and the first wave publishes the partial granules.
//   B  the 16 x Us (batch, unit) elements of the stream in column group ng are split over its KG workgroups; each adds the KG
//      partials of its four gate columns in kg order (fixed => bitwise reproducible), applies the cell (c kept in a register for
//      the whole sequence), writes y / cy / the activated gates (plain stores: read after the launch) and publishes h_t.
// Exchange: 8-byte granules {value, tag = step}, agent-scope store / polled agent-scope loads, 4 rotating slots, bounded polls
// (*err = 2 on time-out; nobody blocks anybody) — exactly the scheme of lstm_bptt2.hip.  The exchange buffers must be ZERO on entry.
#include "nnl_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
typedef int i32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kNH = 2;              // k splits per stream = waves per SIMD
constexpr int kBlock = 256 * kNH;  // 4 streams x kNH waves
constexpr int kRows = 64;          // batch rows = 4 streams x 16
constexpr int kPollLimit = 1 << 20;
constexpr int kMaxNT = 6;
constexpr int kSlots = 4;

struct Fwd2 {
  const float* gx;      // [T][B][4H]
  const float* w;       // [4H][Kp]  W_hh, k padded with zeros
  const float* h0;      // [B][H]
  const float* c0;      // [B][H]
  float* y;             // [T][B][H]
  float* cy;            // [T][B][H]
  float* gates;         // [T][B][4H] activated i, f, g, o (saved for backward)
  u64* xp;              // [kSlots][NG][KG][64][16 NT] partial granules, zero on entry
  u64* xt;              // [kSlots][64][Kp] h granules, zero on entry
  int* err;
  int T, B, H, Kp, KG, NG, Ks, Us, NWG, chunk;
  int dbg;              // timing experiments only (NNL_LSTM_FWD2_DBG; results invalid): 1 no k loop, 2 polls accept any tag
};

__device__ __forceinline__ u64 pack(float v, unsigned tag) { return ((u64)tag << 32) | (u64)__float_as_uint(v); }
__device__ __forceinline__ float val_of(u64 g) { return __uint_as_float((unsigned)g); }
__device__ __forceinline__ unsigned tag_of(u64 g) { return (unsigned)(g >> 32); }
__device__ __forceinline__ void st_granule(u64* p, float v, unsigned tag) {
  __hip_atomic_store(p, pack(v, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 ld_granule(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

template <int NT>     // column tiles of 16 per workgroup (4 Us <= 16 NT)
__global__ __launch_bounds__(kBlock) void lstm_fwd2_kernel(Fwd2 p) {
  extern __shared__ float lds[];                          // W block [Ks/4][16 NT][4], then the hand-over buffers [4][kNH-1][4 NT][64]
  __shared__ int s_pair[4][kNH];
  constexpr int Nsp = 16 * NT;
  constexpr int CH = 3;                                   // k groups (16 k each) per register chunk: two chunks in flight
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = wave >> 2;                                // k part of the stream
  const int m = (wave + h) & 3;                           // stream = batch rows 16 m .. 16 m + 15 (waves of one SIMD: different streams)
  const int H = p.H, B = p.B, Kp = p.Kp, KG = p.KG, NG = p.NG, Ks = p.Ks, Us = p.Us;
  int kg, ng;
  if (KG % 8 == 0 && p.NWG % 8 == 0) {                    // the NG workgroups of one k slice on one XCD
    const int xcd = blockIdx.x & 7, r = blockIdx.x >> 3;
    kg = xcd * (KG / 8) + r / NG;
    ng = r % NG;
  } else {
    kg = blockIdx.x / NG;
    ng = blockIdx.x % NG;
  }
  const int u0 = ng * Us;
  {
    const int kq_n = Ks / 4;
    for (int i = tid; i < Nsp * kq_n; i += kBlock) {
      const int col = i / kq_n, kq = i - col * kq_n;      // consecutive threads: consecutive 16-B pieces of one W row
      const int gate = col / Us, uu = col - gate * Us;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (col < 4 * Us && u0 + uu < H) v = *reinterpret_cast<const f32x4*>(p.w + ((long)gate * H + u0 + uu) * Kp + (long)kg * Ks + 4 * kq);
      *reinterpret_cast<f32x4*>(lds + ((long)kq * Nsp + col) * 4) = v;
    }
    if (tid < 4 * kNH) s_pair[tid / kNH][tid % kNH] = 0;
  }
  float* pairbuf = lds + (long)Ks * Nsp + (long)m * (kNH - 1) * (4 * NT * 64);
  // the (batch, unit) element of stream m this thread owns for the whole sequence
  const int idx = lane + 64 * h;
  const int e = kg * p.chunk + idx;
  const bool valid = idx < p.chunk && e < 16 * Us;
  const int eb = 16 * m + (valid ? e / Us : 0);
  const int euu = valid ? e % Us : 0;
  const int ej = u0 + euu;
  const bool eok = valid && eb < B && ej < H;
  const unsigned o = (unsigned)(eb * H + ej);             // 32-bit per-lane offsets against wave-uniform bases
  const unsigned og = (unsigned)(eb * 4 * H + ej);
  const unsigned ox = (unsigned)(eb * Kp + ej), op = (unsigned)(eb * Nsp + euu);
  float c_state = eok ? p.c0[o] : 0.f;
  __syncthreads();
  const long BH = (long)B * H, BG = (long)B * 4 * H;
  const long tile = (long)kRows * Nsp;                    // one workgroup's partial block (granules)
  const long xt_slot = (long)kRows * Kp;
  const int row = lane & 15, qk = lane >> 4;
  const bool rok = 16 * m + row < B;
  const int ngrp = Ks / 16;
  const int ga = h * ngrp / kNH, gb = (h + 1) * ngrp / kNH;      // this wave's k groups
  const bool any_tag = (p.dbg & 2) != 0;
  int timed_out = 0;
  if (eok) st_granule(p.xt + ox, p.h0[o], 1u);            // step 0: publish h0 (slot 0, tag 1)
  for (int s = 1; s <= p.T; ++s) {                        // step s computes timestep t = s-1 from h_{t-1} (published with tag s)
    const int t = s - 1;
    // this step's input projections: nobody else's results, requested before the k loop
    float gxv[4] = {0.f, 0.f, 0.f, 0.f};
    if (eok) {
#pragma unroll
      for (int g = 0; g < 4; ++g) gxv[g] = (p.gx + t * BG + (long)g * H)[og];
    }
    const unsigned tag = (unsigned)s;
    u64* xp_s = p.xp + ((long)(s % kSlots) * NG + ng) * KG * tile;
    {
      // ---- phase A: this wave's k groups of h_{t-1} W_hh^T for the stream's 16 rows ---------------------------------------
      f32x4 acc[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
      const long kb = (long)kg * Ks + 4 * qk;             // this lane's first k
      const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(p.xt + (long)((s - 1) % kSlots) * xt_slot, 0, (int)(xt_slot * 8), 0x00020000);
      const unsigned aoff = (unsigned)(((long)(16 * m + (rok ? row : 0)) * Kp + kb) * 8);
      const float* bp = lds + ((long)qk * Nsp + row) * 4;
      i32x4 cur[CH][2], nxt[CH][2];
      auto live = [&](int g) { return rok && g < gb && kb + 16 * g < H; };           // (k >= H: nobody writes those granules, they stay zero)
      auto fetch = [&](i32x4 (&dst)[CH][2], int g0) {
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          const unsigned off = live(g0 + j) ? aoff + (unsigned)(16 * (g0 + j)) * 8 : 0xFFFFFFFFu;      // out of range: zeros
          dst[j][0] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)off, 0, 16));
          dst[j][1] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(off == 0xFFFFFFFFu ? off : off + 16), 0, 16));
        }
      };
      // H need not be a multiple of 4: the last live group of a lane may hold pad granules (k >= H), checked one by one
      auto bad4 = [&](const i32x4 (&c)[2], int g) {
        const long k0 = kb + 16 * g;
        return ((unsigned)c[0][1] < tag && k0 < H) || ((unsigned)c[0][3] < tag && k0 + 1 < H) ||
               ((unsigned)c[1][1] < tag && k0 + 2 < H) || ((unsigned)c[1][3] < tag && k0 + 3 < H);
      };
      auto stale = [&](const i32x4 (&c)[CH][2], int g0) {
        bool bad = false;
#pragma unroll
        for (int j = 0; j < CH; ++j) bad |= live(g0 + j) && bad4(c[j], g0 + j);
        return __builtin_amdgcn_ballot_w64(bad && !any_tag) != 0;
      };
      const int gend = (p.dbg & 1) ? ga : gb;
      auto lds_b = [&](f32x4 (&dst)[NT], int g) {
        const float* bq = bp + (long)(4 * g) * Nsp * 4;
#pragma unroll
        for (int n = 0; n < NT; ++n) dst[n] = *reinterpret_cast<const f32x4*>(bq + 16 * n * 4);
      };
      auto mfma_group = [&](const i32x4 (&a2)[2], const f32x4 (&bfr)[NT]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float a = __int_as_float(a2[i >> 1][2 * (i & 1)]);
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bfr[n][i], acc[n], 0, 0, 0);
        }
      };
      int g0 = ga;                                         // full chunks: no condition inside the unrolled body (see lstm_bptt2.hip)
      if (g0 + CH <= gend) fetch(cur, g0);
      for (; g0 + CH <= gend; g0 += CH) {
        if (g0 + 2 * CH <= gend) fetch(nxt, g0 + CH);
        for (int tries = 0; stale(cur, g0); ++tries) {      // wave-uniform: the whole chunk is requested again
          if (tries > kPollLimit || timed_out) { timed_out = 1; break; }
          __builtin_amdgcn_s_sleep(1);
          fetch(cur, g0);
        }
        f32x4 bf[2][NT];
        lds_b(bf[0], g0);
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          if (j + 1 < CH) lds_b(bf[(j + 1) & 1], g0 + j + 1);
          mfma_group(cur[j], bf[j & 1]);
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) { cur[j][0] = nxt[j][0]; cur[j][1] = nxt[j][1]; }
      }
      for (; g0 < gend; ++g0) {                            // the ragged rest, one group at a time
        i32x4 one[2];
        const bool lv = live(g0);
        auto fetch1 = [&]() {
          const unsigned off = lv ? aoff + (unsigned)(16 * g0) * 8 : 0xFFFFFFFFu;
          one[0] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)off, 0, 16));
          one[1] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(off == 0xFFFFFFFFu ? off : off + 16), 0, 16));
        };
        fetch1();
        f32x4 b1[NT];
        lds_b(b1, g0);
        for (int tries = 0; __builtin_amdgcn_ballot_w64(lv && bad4(one, g0) && !any_tag) != 0; ++tries) {
          if (tries > kPollLimit || timed_out) { timed_out = 1; break; }
          __builtin_amdgcn_s_sleep(1);
          fetch1();
        }
        mfma_group(one, b1);
      }
      if (h) {
        // the other k parts: hand the sums to the stream's first wave through LDS (same lane layout on both sides)
        float* pb = pairbuf + (long)(h - 1) * (4 * NT * 64);
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int v = 0; v < 4; ++v) pb[(4 * n + v) * 64 + lane] = acc[n][v];
        __hip_atomic_store(&s_pair[m][h], s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else {
        auto partners_ready = [&]() {
          bool r = true;
#pragma unroll
          for (int hh = 1; hh < kNH; ++hh) r &= __hip_atomic_load(&s_pair[m][hh], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= s;
          return r;
        };
        for (int tries = 0; !partners_ready(); ++tries) {       // (a stream without valid batch rows waits for nobody: its partners may be steps ahead, all sides hold zeros)
          if (tries > kPollLimit) { timed_out = 1; break; }      // (cannot happen: the partners' own polls are bounded)
          __builtin_amdgcn_s_sleep(1);
        }
        u64* pp = xp_s + (long)kg * tile;
        const unsigned opp = (unsigned)((16 * m + 4 * qk) * Nsp + row);
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            float sum = acc[n][v];                                     // parts added in k order: fixed => bitwise reproducible
#pragma unroll
            for (int hh = 1; hh < kNH; ++hh) sum += pairbuf[(long)(hh - 1) * (4 * NT * 64) + (4 * n + v) * 64 + lane];
            st_granule(pp + (v * Nsp + 16 * n) + opp, sum, tag);
          }
      }
    }
    // ---- phase B: the four pre-activations of the owned element = gx + the KG partials in kg order, then the cell ----------
    float pre[4] = {gxv[0], gxv[1], gxv[2], gxv[3]};
    for (int k0 = 0; k0 < KG; k0 += 4) {                   // 4 kg x 4 gates = 16 granules per batch (whole wave together)
      u64 v[4][4];
      auto fetch16 = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            v[j][g] = (eok && k0 + j < KG) ? ld_granule(xp_s + (long)(k0 + j) * tile + g * Us + op) : pack(0.f, tag);
      };
      auto stale16 = [&]() {
        bool bad = false;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) bad |= tag_of(v[j][g]) < tag;
        return __builtin_amdgcn_ballot_w64(bad && !any_tag) != 0;
      };
      fetch16();
      for (int tries = 0; stale16(); ++tries) {
        if (tries > kPollLimit || timed_out) { timed_out = 1; break; }
        __builtin_amdgcn_s_sleep(1);
        fetch16();
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) pre[g] += val_of(v[j][g]);         // (absent kg: +0.f)
    }
    if (eok) {
      const float gi = sigmoidf_(pre[0]), gf = sigmoidf_(pre[1]), gg = tanhf(pre[2]), go = sigmoidf_(pre[3]);
      const float c = gf * c_state + gi * gg;
      const float hv = go * tanhf(c);
      c_state = c;
      if (s < p.T) st_granule(p.xt + (long)(s % kSlots) * xt_slot + ox, hv, tag + 1);      // what the other workgroups wait for goes first
      (p.y + t * BH)[o] = hv;
      (p.cy + t * BH)[o] = c;
      float* gt = p.gates + t * BG;
      gt[og] = gi; (gt + H)[og] = gf; (gt + 2 * H)[og] = gg; (gt + 3 * H)[og] = go;
    }
  }
  if (timed_out && lane == 0) __hip_atomic_store(p.err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct PlanF { int KG, NG, Ks, Us, NT, chunk; size_t lds; bool ok; };

// Partition by a small cost model (us per step): MFMA time of the k loop, the operand traffic, the partial traffic / round trips.
PlanF planf(long B, long H, long Kp) {
  PlanF best{};
  best.ok = false;
  if (B < 1 || B > kRows || H < 1 || Kp < H || Kp % 16 != 0) return best;
  const int fkg = NNL_ENV_INT("NNL_LSTM_FWD2_KG", 0), fng = NNL_ENV_INT("NNL_LSTM_FWD2_NG", 0);
  double best_cost = 1e30;
  for (int NG = 1; NG <= 256; ++NG) {
    if (fng > 0 && NG != fng) continue;
    const int Us = (int)nnl_cdiv(H, NG);
    if ((long)Us * (NG - 1) >= H) continue;               // an empty column group
    const int NT = (int)nnl_cdiv(4L * Us, 16);
    if (NT > kMaxNT) continue;
    for (int KG = 1; KG * NG <= 256; ++KG) {
      if (fkg > 0 && KG != fkg) continue;
      if (Kp % (16 * KG) != 0) continue;
      const int Ks = (int)(Kp / KG);
      const size_t lds = ((size_t)Ks * 16 * NT + 4u * (kNH - 1) * 4 * NT * 64) * sizeof(float);
      if (lds > 156 * 1024) continue;
      const int chunk = (int)nnl_cdiv(16L * Us, KG);          // elements of one stream per workgroup
      if (chunk > 64 * kNH) continue;
      const double wgs = (double)KG * NG;
      const double mfma = (Ks / 16.0) * NT * 4 * 32 / 2400.0;
      const double operand = wgs * kRows * Ks * 8.0 / 6.7e6;
      const double parts = wgs * kRows * 16.0 * NT * 8.0 * 2 / 16.0e6;
      const double batches = (double)nnl_cdiv(KG, 4);       // serialized partial round trips of phase B
      const double cost = mfma + 0.5 * operand + 1.5 * batches + parts + (wgs < 128 ? 1.0 : 0.0);
      if (cost < best_cost) {
        best_cost = cost;
        best = PlanF{KG, NG, Ks, Us, NT, chunk, lds, true};
      }
    }
  }
  return best;
}

template <int NT>
hipError_t launchf(Fwd2& p, size_t lds, hipStream_t s) {
  auto kernel = lstm_fwd2_kernel<NT>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  void* args[] = {&p};
  return hipLaunchCooperativeKernel(reinterpret_cast<const void*>(kernel), dim3(p.NWG), dim3(kBlock), args, (unsigned)lds, s);
}

size_t xp_granules(const PlanF& pl) { return (size_t)kSlots * pl.NG * pl.KG * kRows * 16 * pl.NT; }
size_t xt_granules(long Kp) { return (size_t)kSlots * kRows * Kp; }

}  // namespace

// ---- entry points used by lstm.hip ----------------------------------------------------------------------------------
bool nnl_lstm_fwd2_ok(long B, long H, long Kp) { return planf(B, H, Kp).ok; }

// workspace (floats): the two granule exchanges (8 bytes per granule)
size_t nnl_lstm_fwd2_ws_floats(long T, long B, long H, long Kp) {
  const PlanF pl = planf(B, H, Kp);
  if (!pl.ok) return 0;
  return 2 * (xp_granules(pl) + xt_granules(Kp)) + 16;
}

// for tools / tests: the partition the planner picks ([KG, NG, Ks, Us, NT]); 0 when the shape does not fit
extern "C" int nnl_debug_lstm_fwd2_plan(int64_t B, int64_t H, int32_t* out5) {
  const PlanF pl = planf(B, H, nnl_cdiv(H, 32) * 32);
  if (!pl.ok) return 0;
  out5[0] = pl.KG; out5[1] = pl.NG; out5[2] = pl.Ks; out5[3] = pl.Us; out5[4] = pl.NT;
  return 1;
}

// returns hipSuccess when the cooperative launch was issued; any other value: nothing was launched, take another path
hipError_t nnl_lstm_fwd2(const float* gx, const float* w_hh_pad, const float* h0, const float* c0, float* y, float* cy, float* gates,
                         long T, long B, long H, long Kp, float* ws, int* err, hipStream_t s) {
  const PlanF pl = planf(B, H, Kp);
  if (!pl.ok || (reinterpret_cast<uintptr_t>(ws) & 7) != 0) return hipErrorInvalidValue;
  Fwd2 p{};
  p.gx = gx; p.w = w_hh_pad; p.h0 = h0; p.c0 = c0; p.y = y; p.cy = cy; p.gates = gates;
  p.xp = reinterpret_cast<u64*>(ws);
  p.xt = p.xp + xp_granules(pl);
  p.err = err;
  p.T = (int)T; p.B = (int)B; p.H = (int)H; p.Kp = (int)Kp;
  p.dbg = NNL_ENV_INT("NNL_LSTM_FWD2_DBG", 0);
  p.KG = pl.KG; p.NG = pl.NG; p.Ks = pl.Ks; p.Us = pl.Us; p.NWG = pl.KG * pl.NG; p.chunk = pl.chunk;
  hipError_t e = hipMemsetAsync(p.xp, 0, sizeof(u64) * (xp_granules(pl) + xt_granules(Kp)), s);     // tag 0 = nothing yet
  if (e != hipSuccess) return e;
  switch (pl.NT) {
    case 1: return launchf<1>(p, pl.lds, s);
    case 2: return launchf<2>(p, pl.lds, s);
    case 3: return launchf<3>(p, pl.lds, s);
    case 4: return launchf<4>(p, pl.lds, s);
    case 5: return launchf<5>(p, pl.lds, s);
    default: return launchf<6>(p, pl.lds, s);
  }
}
