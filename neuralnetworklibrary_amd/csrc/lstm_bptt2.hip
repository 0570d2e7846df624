// K5 — persistent BPTT of one LSTM layer on a 2-D PARTITION of W_hh (round 4): one cooperative launch runs all T backward
// timesteps of WeightDropLSTM1.forward -> nn.LSTM (cuDNN's persistent RNN in the reference; Applications/Text.py:495-513,
// :535-551).  It replaces the per-timestep pair of lstm.hip (split-K GEMM + cell kernel, 20 us per step at H = 1150) and the
// first persistent BPTT (lstm_persist.hip: every workgroup streams ALL of dgates_{t+1}, 1.2 MB per step — 42 us).
//
// Per step:  dh_t[b][j] = dy_t[b][j] + sum_k dgates_{t+1}[b][k] * W_hh[k][j]   (k < 4H, j < H),  then the pointwise cell backward.
//
// Partition.  KG x NG workgroups (<= 256, one per CU, co-resident: cooperative launch).  Workgroup (kg, ng) keeps the
// [Ks = Gp/KG] x [Ns = ceil(H/NG)] block of W_hh in LDS for all timesteps (H = 1150: 16 x 16 workgroups, 288 x 72 -> 92 KB) —
// W_hh leaves HBM once per layer.  A step has two phases:
//   A  partial[kg][b][j] = sum_{k in slice kg} dgates_{t+1}[b][k] W[k][j]  for the 64 batch rows and the Ns columns of ng:
//      v_mfma_f32_16x16x4_f32, wave w = batch rows 16w..16w+15, NT = ceil(Ns/16) column tiles.  A lane holds 4 consecutive k of
//      its row: the MFMA's k slots are only a summation label, so slot q of MFMA i is k = 16g + 4q + i and one ds_read_b128 of the
//      [k/4][col][4] LDS image feeds the matching B operands of four MFMAs.  64 x Ks values per workgroup and step instead of
//      the whole 64 x 4H.
//   B  the 64 x Ns (batch, unit) elements of column slice ng are split over its KG workgroups; each element adds the KG partials
//      in kg order (fixed order => bitwise reproducible), runs the cell backward with dc kept in a register for the whole
//      sequence, and writes its four gate gradients: to the tape dgates[t] (plain stores: read by the weight-gradient GEMMs
//      after the launch) and to the exchange.
// Exchange without barriers.  A first version met at two counters per step (drain the write-through stores, one agent-scope
// atomic, poll, then load the data): six serialized device-scope round trips, ~10 of its 27 us per step.  Now every exchanged
// value travels as an 8-byte GRANULE {value, tag} written by ONE agent-scope store and polled by the consumer with agent-scope
// loads until the tag equals the step it waits for: the data is its own flag — no drain, no counter, no barrier; a hand-over costs
// one store -> load latency.  Granules of different steps rotate through 4 slots: a workgroup that is in phase A of step s + 2
// implies every workgroup has finished phase B of step s (every unit's four gate columns feed some k slice of every column
// group), so a slot is never overwritten while somebody may still read it.  The one exception are workgroups that wait for
// nothing (a k slice that lies entirely in the zero padding, no valid batch rows): they may run many steps ahead — but everything
// they ever publish is zero, so a consumer accepts any tag >= the step it waits for (an older tag: keep polling).
// Polls are bounded: on time-out the kernel sets *err = 2 and carries on to the end (no workgroup ever blocks another: the grid
// always drains).  The exchange buffers must be ZERO on entry (tag 0 = nothing yet; steps count from 1).
// Workgroups that share a k slice poll the same granules: they are placed on one XCD (blocks b and b + 8 share an XCD).
#include "nnl_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
typedef int i32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kNH = 2;              // k splits per stream = waves per SIMD
constexpr int kBlock = 256 * kNH;  // 4 streams x kNH waves
constexpr int kRows = 64;          // batch rows = 4 waves x 16
constexpr int kPollLimit = 1 << 20;
constexpr int kMaxNT = 6;
constexpr int kSlots = 4;

struct Bptt2 {
  const float* dy;      // [T][B][H] or null
  const float* dhT;     // [B][H] or null
  const float* dcT;     // [B][H] or null
  const float* gates;   // [T][B][4H] activated gates of the forward
  const float* cy;      // [T][B][H]
  const float* c0;      // [B][H]
  const float* wt;      // [>=H][Gp]  W_hh^T, k (= gate column) padded with zeros
  float* dgates;        // [T][B][Gp] the tape (pad columns zero on entry, never written)
  u64* xp;              // [kSlots][NG][KG][64][16 NT] partial granules, zero on entry
  u64* xt;              // [kSlots][64][Gp] gate-gradient granules, zero on entry
  float* dh0;           // [B][H]
  float* dc0;           // [B][H]
  int* err;
  int T, B, H, Gp, KG, NG, Ks, Ns, NWG, chunk;
  int dbg;              // timing experiments only (NNL_LSTM_BPTT2_DBG; results invalid): 1 no k loop, 2 polls accept any tag, 4 staggered start of streams 2 / 3, 8 raised wave priority in the k loop
};

__device__ __forceinline__ u64 pack(float v, unsigned tag) { return ((u64)tag << 32) | (u64)__float_as_uint(v); }
__device__ __forceinline__ float val_of(u64 g) { return __uint_as_float((unsigned)g); }
__device__ __forceinline__ unsigned tag_of(u64 g) { return (unsigned)(g >> 32); }
__device__ __forceinline__ void st_granule(u64* p, float v, unsigned tag) {
  __hip_atomic_store(p, pack(v, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 ld_granule(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int NT, int kPB>     // column tiles of 16 per workgroup; partial granules polled per batch in phase B
__global__ __launch_bounds__(kBlock) void lstm_bptt2_kernel(Bptt2 p) {
  extern __shared__ float lds[];                          // W block [Ks/4][16 NT][4], then the pair buffers [4][4 NT][64]
  __shared__ int s_pair[4][kNH];
  __shared__ int s_done[4];                                // last step whose pair buffers the stream's first wave has consumed
  constexpr int Nsp = 16 * NT;
  constexpr int CH = 3;                                   // k groups (16 k each) per register chunk: two chunks in flight
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Four independent STREAMS (the batch-row tiles m = 0..3: the recurrence never mixes batch rows), kNH waves each: wave (m, h)
  // takes k part h of the stream's phase A and a share of its phase-B elements.  Waves w, w + 4, w + 8, w + 12 share a SIMD: they
  // belong to four different streams, so while a stream waits for its exchange (two store -> load hand-overs, ~6 us per step) the
  // others have the matrix pipe.
  const int h = wave >> 2;
  const int m = (wave + h) & 3;
  const int H = p.H, B = p.B, Gp = p.Gp, KG = p.KG, NG = p.NG, Ks = p.Ks, Ns = p.Ns;
  int kg, ng;
  if (KG % 8 == 0 && p.NWG % 8 == 0) {                    // the NG workgroups of one k slice on one XCD
    const int xcd = blockIdx.x & 7, r = blockIdx.x >> 3;
    kg = xcd * (KG / 8) + r / NG;
    ng = r % NG;
  } else {
    kg = blockIdx.x / NG;
    ng = blockIdx.x % NG;
  }
  {
    const int kq_n = Ks / 4;
    for (int i = tid; i < Nsp * kq_n; i += kBlock) {
      const int col = i / kq_n, kq = i - col * kq_n;      // consecutive threads: consecutive 16-B pieces of one W^T row
      const int j = ng * Ns + col;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (col < Ns && j < H) v = *reinterpret_cast<const f32x4*>(p.wt + (long)j * Gp + (long)kg * Ks + 4 * kq);
      *reinterpret_cast<f32x4*>(lds + ((long)kq * Nsp + col) * 4) = v;
    }
    if (tid < 4 * kNH) s_pair[tid / kNH][tid % kNH] = 0;
    if (tid < 4) s_done[tid] = 0;
  }
  float* pairbuf = lds + (long)Ks * Nsp + (long)m * (kNH - 1) * (4 * NT * 64);      // [kNH-1][4 NT][64] of this stream
  // the (batch, unit) element of stream m this thread owns for the whole sequence: the 16 Ns elements of the stream's rows in
  // column slice ng are split over the KG workgroups
  const int idx = lane + 64 * h;
  const int e = kg * p.chunk + idx;
  const bool valid = idx < p.chunk && e < 16 * Ns;
  const int eb = 16 * m + (valid ? e / Ns : 0);
  const int ejj = valid ? e % Ns : 0;
  const int ej = ng * Ns + ejj;
  const bool eok = valid && eb < B && ej < H;
  float dc_state = (eok && p.dcT) ? p.dcT[(long)eb * H + ej] : 0.f;
  __syncthreads();
  const long BH = (long)B * H, BG = (long)B * 4 * H;
  const long tile = (long)kRows * Nsp;                    // one workgroup's partial block (granules)
  const long xt_slot = (long)kRows * Gp;
  const int row = lane & 15, qk = lane >> 4;
  const bool rok = 16 * m + row < B;
  const int ngrp = Ks / 16;
  const int ga = h * ngrp / kNH, gb = (h + 1) * ngrp / kNH;      // this wave's k groups
  const bool any_tag = (p.dbg & 2) != 0;
  int timed_out = 0;
  const unsigned o = (unsigned)(eb * H + ej);                      // 32-bit per-lane offsets against wave-uniform bases (saddr addressing:
  const unsigned og = (unsigned)(eb * 4 * H + ej);                 // one VGPR per address instead of two per pointer)
  const unsigned ox = (unsigned)(eb * Gp + ej), op = (unsigned)(eb * Nsp + ejj);
  if ((p.dbg & 4) && m >= 2) {                            // experiment: start streams 2, 3 half a step later
    __builtin_amdgcn_s_sleep(127);
    __builtin_amdgcn_s_sleep(127);
  }
  for (int s = 0; s <= p.T; ++s) {                        // step s handles timestep t = T-1-s; s = T: only dh0 / dc0
    const int t = p.T - 1 - s;
    // operands of this step's cell: nobody else's results, requested before the k loop
    float dyv = 0.f, cv = 0.f, cpv = 0.f, gv[4] = {0.f, 0.f, 0.f, 0.f};
    if (t >= 0 && eok) {
      dyv = p.dy ? (p.dy + t * BH)[o] : 0.f;
      cv = (p.cy + t * BH)[o];
      cpv = t == 0 ? p.c0[o] : (p.cy + (t - 1) * BH)[o];
#pragma unroll
      for (int g = 0; g < 4; ++g) gv[g] = (p.gates + t * BG + (long)g * H)[og];
    }
    const unsigned tag = (unsigned)s;                     // what this step's phases wait for (s >= 1 where anything is awaited)
    u64* xp_s = p.xp + ((long)(s % kSlots) * NG + ng) * KG * tile;
    if (s >= 1) {
      // ---- phase A: this wave's k groups of dgates_{t+1} W_hh for the stream's 16 rows; the operand granules were written in
      // phase B of step s-1
      f32x4 acc[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
      const long kb = (long)kg * Ks + 4 * qk;             // this lane's first k
      // two 16-B agent-scope (sc1) buffer loads per k group: the lane's four consecutive granules (each 8-B half is one granule,
      // written by one store: a torn 16-B read can only mix two whole granules, and each carries its own tag)
      const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(p.xt + (long)((s - 1) % kSlots) * xt_slot, 0, (int)(xt_slot * 8), 0x00020000);
      const unsigned aoff = (unsigned)(((long)(16 * m + (rok ? row : 0)) * Gp + kb) * 8);
      const float* bp = lds + ((long)qk * Nsp + row) * 4;
      i32x4 cur[CH][2], nxt[CH][2];
      auto live = [&](int g) { return rok && g < gb && kb + 16 * g < 4 * H; };       // (pad k: nobody writes those granules)
      auto fetch = [&](i32x4 (&dst)[CH][2], int g0) {
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          const unsigned off = live(g0 + j) ? aoff + (unsigned)(16 * (g0 + j)) * 8 : 0xFFFFFFFFu;      // out of range: zeros
          dst[j][0] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)off, 0, 16));
          dst[j][1] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(off == 0xFFFFFFFFu ? off : off + 16), 0, 16));
        }
      };
      auto stale = [&](const i32x4 (&c)[CH][2], int g0) {
        bool bad = false;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          const bool b4 = (unsigned)c[j][0][1] < tag || (unsigned)c[j][0][3] < tag || (unsigned)c[j][1][1] < tag || (unsigned)c[j][1][3] < tag;
          bad |= b4 && live(g0 + j);
        }
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
      // full chunks of CH groups: no condition inside the unrolled body (a predicate there made the compiler move the accumulators
      // between AGPRs and VGPRs around every group: two pipe drains and 40 moves per 20 MFMAs)
      int g0 = ga;
      if (p.dbg & 8) __builtin_amdgcn_s_setprio(3);        // experiment: the wave that has started its k loop keeps the matrix pipe
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
        auto stale1 = [&]() {
          const bool b4 = (unsigned)one[0][1] < tag || (unsigned)one[0][3] < tag || (unsigned)one[1][1] < tag || (unsigned)one[1][3] < tag;
          return __builtin_amdgcn_ballot_w64(b4 && lv && !any_tag) != 0;
        };
        fetch1();
        f32x4 b1[NT];
        lds_b(b1, g0);
        for (int tries = 0; stale1(); ++tries) {
          if (tries > kPollLimit || timed_out) { timed_out = 1; break; }
          __builtin_amdgcn_s_sleep(1);
          fetch1();
        }
        mfma_group(one, b1);
      }
      if (p.dbg & 8) __builtin_amdgcn_s_setprio(0);
      if (h) {
        // the other k parts: hand the sums to the stream's first wave through LDS (same lane layout on both sides)
        float* pb = pairbuf + (long)(h - 1) * (4 * NT * 64);
        // the single pair buffer is re-written every step: not before the first wave has read step s - 1's sums (with
        // chunk <= 64 this wave owns no phase-B element, so nothing else orders it behind that read)
        for (int tries = 0; __hip_atomic_load(&s_done[m], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < s - 1; ++tries) {
          if (tries > kPollLimit) { timed_out = 1; break; }
          __builtin_amdgcn_s_sleep(1);
        }
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
        for (int tries = 0; !partners_ready(); ++tries) {      // (a stream without valid batch rows waits for nobody: its partners may be steps ahead, all sides hold zeros)
          if (tries > kPollLimit) { timed_out = 1; break; }      // (cannot happen: the partner's own polls are bounded)
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
        __hip_atomic_store(&s_done[m], s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);      // the pair buffers may be re-written
      }
    }
    // ---- phase B: complete dh_t of the owned element, cell backward ---------------------------------------------------
    float dh = 0.f;
    if (s >= 1) {
      // the KG partials of this element, in kg order; polled in batches of 16 — one round trip at KG = 16 (whole wave together: the
      // trip counts are uniform)
      for (int k0 = 0; k0 < KG; k0 += kPB) {
        u64 v[kPB];
        auto fetch8 = [&]() {
#pragma unroll
          for (int j = 0; j < kPB; ++j) v[j] = (eok && k0 + j < KG) ? ld_granule(xp_s + (long)(k0 + j) * tile + op) : pack(0.f, tag);
        };
        auto stale8 = [&]() {
          bool bad = false;
#pragma unroll
          for (int j = 0; j < kPB; ++j) bad |= tag_of(v[j]) < tag;
          return __builtin_amdgcn_ballot_w64(bad && !any_tag) != 0;
        };
        fetch8();
        for (int tries = 0; stale8(); ++tries) {
          if (tries > kPollLimit || timed_out) { timed_out = 1; break; }
          __builtin_amdgcn_s_sleep(1);
          fetch8();
        }
#pragma unroll
        for (int j = 0; j < kPB; ++j) dh += val_of(v[j]);     // (absent kg: +0.f)
      }
    } else if (eok && p.dhT) {
      dh = p.dhT[o];
    }
    if (eok) {
      if (t < 0) {
        p.dh0[o] = dh;
        p.dc0[o] = dc_state;
      } else {
        dh += dyv;
        const float gi = gv[0], gf = gv[1], gg = gv[2], go = gv[3];
        const float tc = tanhf(cv);
        const float dcn = dc_state + dh * go * (1.f - tc * tc);
        const float d0 = dcn * gg * (gi * (1.f - gi)), d1 = dcn * cpv * (gf * (1.f - gf));
        const float d2 = dcn * gi * (1.f - gg * gg), d3 = dh * tc * (go * (1.f - go));
        u64* xg = p.xt + (long)(s % kSlots) * xt_slot;            // what the other workgroups wait for goes first
        st_granule(xg + ox, d0, tag + 1);
        st_granule(xg + H + ox, d1, tag + 1);
        st_granule(xg + 2 * H + ox, d2, tag + 1);
        st_granule(xg + 3 * H + ox, d3, tag + 1);
        float* dg = p.dgates + (long)t * B * Gp;
        dg[ox] = d0; (dg + H)[ox] = d1; (dg + 2 * H)[ox] = d2; (dg + 3 * H)[ox] = d3;
        dc_state = dcn * gf;
      }
    }
  }
  if (timed_out && lane == 0) __hip_atomic_store(p.err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct Plan2 { int KG, NG, Ks, Ns, NT, chunk; size_t lds; bool ok; };

// Partition by a small cost model (us per step): MFMA time of the k loop, the operand traffic through the L2s, the partial traffic.
Plan2 plan2(long B, long H, long Gp) {
  Plan2 best{};
  best.ok = false;
  if (B < 1 || B > kRows || H < 1 || Gp < 4 * H || Gp % 16 != 0) return best;
  const int fkg = NNL_AB_INT("NNL_LSTM_BPTT2_KG", 0), fng = NNL_AB_INT("NNL_LSTM_BPTT2_NG", 0);
  double best_cost = 1e30;
  for (int NG = 1; NG <= 256; ++NG) {
    if (fng > 0 && NG != fng) continue;
    const int Ns = (int)nnl_cdiv(H, NG);
    if ((long)Ns * (NG - 1) >= H) continue;               // an empty column slice
    const int NT = (int)nnl_cdiv(Ns, 16);
    if (NT > kMaxNT) continue;
    for (int KG = 1; KG * NG <= 256; ++KG) {
      if (fkg > 0 && KG != fkg) continue;
      if (Gp % (16 * KG) != 0) continue;
      const int Ks = (int)(Gp / KG);
      const size_t lds = ((size_t)Ks * 16 * NT + 4u * (kNH - 1) * 4 * NT * 64) * sizeof(float);      // W block + the streams' hand-over buffers
      if (lds > 156 * 1024) continue;
      const int chunk = (int)nnl_cdiv(16L * Ns, KG);          // elements of one stream (16 batch rows) per workgroup
      if (chunk > 64 * kNH) continue;
      const double wgs = (double)KG * NG;
      const double mfma = (Ks / 16.0) * NT * 4 * 32 / 2400.0;
      const double operand = wgs * kRows * Ks * 8.0 / 6.7e6;
      const double parts = wgs * kRows * 16.0 * NT * 8.0 * 2 / 16.0e6;
      // (measured at H = 1150, us per step: (16,16) 15.3, (16,15) 15.5, (12,18) 16.4, (8,24) 17.2, (18,12) 17.6, (9,24) 18.3)
      const double batches = (double)nnl_cdiv(KG, 16);      // serialized partial round trips of phase B
      const double xcd_local = (KG % 8 == 0 && ((long)KG * NG) % 8 == 0) ? 0.0 : 1.0;      // k slices not kept on one XCD
      const double cost = mfma + 0.5 * operand + 2.0 * batches + xcd_local + parts;
      if (cost < best_cost) {
        best_cost = cost;
        best = Plan2{KG, NG, Ks, Ns, NT, chunk, lds, true};
      }
    }
  }
  return best;
}

template <int NT, int PB>
hipError_t launch2(Bptt2& p, size_t lds, hipStream_t s) {
  auto kernel = lstm_bptt2_kernel<NT, PB>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  void* args[] = {&p};
  return hipLaunchCooperativeKernel(reinterpret_cast<const void*>(kernel), dim3(p.NWG), dim3(kBlock), args, (unsigned)lds, s);
}

size_t xp_granules(const Plan2& pl) { return (size_t)kSlots * pl.NG * pl.KG * kRows * 16 * pl.NT; }
size_t xt_granules(long Gp) { return (size_t)kSlots * kRows * Gp; }

}  // namespace

// ---- entry points used by lstm.hip ----------------------------------------------------------------------------------
bool nnl_lstm_bptt2_ok(long B, long H, long Gp) { return plan2(B, H, Gp).ok; }

// workspace (floats): the two granule exchanges (8 bytes per granule)
size_t nnl_lstm_bptt2_ws_floats(long T, long B, long H, long Gp) {
  const Plan2 pl = plan2(B, H, Gp);
  if (!pl.ok) return 0;
  return 2 * (xp_granules(pl) + xt_granules(Gp)) + 16;
}

// for tools / tests: the partition the planner picks ([KG, NG, Ks, Ns, NT]); 0 when the shape does not fit
extern "C" int nnl_debug_lstm_bptt2_plan(int64_t B, int64_t H, int32_t* out5) {
  const Plan2 pl = plan2(B, H, nnl_cdiv(4 * H, 32) * 32);
  if (!pl.ok) return 0;
  out5[0] = pl.KG; out5[1] = pl.NG; out5[2] = pl.Ks; out5[3] = pl.Ns; out5[4] = pl.NT;
  return 1;
}

// returns hipSuccess when the cooperative launch was issued; any other value: nothing was launched, take another path
hipError_t nnl_lstm_bptt2(const float* dy, const float* dhT, const float* dcT, const float* gates, const float* cy, const float* c0,
                          const float* w_hh_t_pad, float* dgates_pad, float* dh0, float* dc0, long T, long B, long H, long Gp,
                          float* ws, int* err, hipStream_t s) {
  const Plan2 pl = plan2(B, H, Gp);
  if (!pl.ok || (reinterpret_cast<uintptr_t>(ws) & 7) != 0) return hipErrorInvalidValue;
  Bptt2 p{};
  p.dy = dy; p.dhT = dhT; p.dcT = dcT; p.gates = gates; p.cy = cy; p.c0 = c0; p.wt = w_hh_t_pad;
  p.dgates = dgates_pad; p.dh0 = dh0; p.dc0 = dc0;
  p.xp = reinterpret_cast<u64*>(ws);
  p.xt = p.xp + xp_granules(pl);
  p.err = err;
  p.T = (int)T; p.B = (int)B; p.H = (int)H; p.Gp = (int)Gp;
  p.dbg = NNL_AB_INT("NNL_LSTM_BPTT2_DBG", 0);
  p.KG = pl.KG; p.NG = pl.NG; p.Ks = pl.Ks; p.Ns = pl.Ns; p.NWG = pl.KG * pl.NG; p.chunk = pl.chunk;
  hipError_t e = hipMemsetAsync(p.xp, 0, sizeof(u64) * (xp_granules(pl) + xt_granules(Gp)), s);     // tag 0 = nothing yet
  if (e != hipSuccess) return e;
  // partial polls of phase B in batches of 8 or 16 granules (measured: KG = 16 -> 2 x 8: 15.9 vs 16.3 us per step at H = 1150;
  // KG = 10 -> one batch of 16: 9.5 vs 10.1 at H = 400)
  const int pb_env = NNL_AB_INT("NNL_LSTM_BPTT2_PB", 0);
  const int pb = pb_env == 8 || pb_env == 16 ? pb_env : (pl.KG % 8 == 0 ? 8 : 16);
#define NNL_BPTT2_CASE(N) case N: return pb == 8 ? launch2<N, 8>(p, pl.lds, s) : launch2<N, 16>(p, pl.lds, s)
  switch (pl.NT) {
    NNL_BPTT2_CASE(1);
    NNL_BPTT2_CASE(2);
    NNL_BPTT2_CASE(3);
    NNL_BPTT2_CASE(4);
    NNL_BPTT2_CASE(5);
    default: return pb == 8 ? launch2<6, 8>(p, pl.lds, s) : launch2<6, 16>(p, pl.lds, s);
  }
#undef NNL_BPTT2_CASE
}
