// k-loop probe: the 64x64-tile fp32 MFMA main loop of the conv kernels, staged three ways, on random data.
//   REG : buffer_load_dwordx4 -> VGPR -> ds_write_b128 (padded rows) -> barrier            (what igemm_taps_kernel does today)
//   DMA : buffer_load_dwordx4 ... lds (LDS-DMA) into an unpadded, source-swizzled image -> vmcnt(0) + barrier
//   DMA3: same with three LDS buffers and one tile in flight across the barrier (counted vmcnt, raw s_barrier)
// hipcc --offload-arch=gfx950 -O3 tools/kloop_probe.hip -o tools/kloop_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// source: [rows][256] floats per "image"; a tile iteration reads 64 rows x BK floats of A and of B at channel offset c0
template <int BK, int MODE, bool KSPLIT = false>   // MODE 0 REG, 1 DMA (2 buffers), 2 DMA3 (3 buffers); KSPLIT: each wave owns the whole
__global__ __launch_bounds__(256) void kloop(  // 64x64 tile over a quarter of BK (4 accumulators, half the fragment reads)
float* out, const float* __restrict__ src, int src_bytes, int iters) {
  constexpr int KC = BK / 4;                        // 16-B chunks per row
  constexpr int BKP = MODE == 0 ? BK + 4 : BK;      // padded rows only for the register-staged image
  constexpr int NBUF = MODE == 2 ? 3 : 2;
  constexpr int TILE = 128 * BKP;                   // floats per buffer (A 64 rows + B 64 rows)
  __shared__ __attribute__((aligned(16))) float lds[NBUF * TILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, src_bytes, 0x00020000);
  const unsigned blk_base = (unsigned)(blockIdx.x % 61) * 128u * 1024u;     // 128 rows x 1 KB per block, L2-resident set

  f32x16 acc, acc2, acc3, acc4;
  for (int e = 0; e < 16; ++e) { acc[e] = 0.f; acc2[e] = 0.f; acc3[e] = 0.f; acc4[e] = 0.f; }

  // ---- fragment read addresses ----
  const int frow = lane & 31, fk = lane >> 5;       // row within the wave's 32, k half
  auto frag_ptr = [&](int buf, int is_b, int kk /* 8-wide k group */) -> const float* {
    const int row = (is_b ? 64 + wn * 32 : wm * 32) + frow;
    const int chunk = kk * 2 + fk;                  // 16-B chunk index within the row
    if (MODE == 0) return lds + buf * TILE + row * BKP + chunk * 4;
    const int sw = BK == 16 ? (row >> 2) & 3 : (row >> 1) & 7;
    return lds + buf * TILE + row * BK + ((chunk ^ sw) * 4);
  };

  // ---- staging addresses ----
  // REG: thread -> (row = tid / KC (+ 256/KC per pass), chunk = tid % KC)
  constexpr int RPP = 256 / KC, PASSES = 64 / RPP;
  f32x4 ra[PASSES], rb[PASSES];
  auto issue = [&](int it, int buf) {
    const unsigned c0 = (unsigned)((it * BK) & 255) * 4u;
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < PASSES; ++i) {
        const unsigned row = tid / KC + i * RPP, ch = tid % KC;
        ra[i] = buf_load4(rs, blk_base + row * 1024u + ch * 16u, c0);
        rb[i] = buf_load4(rs, blk_base + (64u + row) * 1024u + ch * 16u, c0);
      }
    } else {
      // DMA: one wave-instruction fills 1 KB = 64 slots of 16 B, lane-linear; slot s of the A (or B) image = (row = s / KC, chunk' = s % KC)
      constexpr int PIECES = 64 * KC / 64 / 4;      // wave-instructions per wave per operand (BK16: 1, BK32: 2)
#pragma unroll
      for (int pc = 0; pc < PIECES; ++pc) {
        const int s = (wave * PIECES + pc) * 64 + lane;
        const int row = s / KC, chp = s % KC;
        const int sw = BK == 16 ? (row >> 2) & 3 : (row >> 1) & 7;
        const unsigned ch = (unsigned)(chp ^ sw);
        float* dstA = lds + buf * TILE + (wave * PIECES + pc) * 256;
        float* dstB = dstA + 64 * BK;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDSP(dstA), 16, blk_base + (unsigned)row * 1024u + ch * 16u, c0, 0, 0);
        // B rows 64..127: their swizzle uses the B-local row (row) as well (frag_ptr uses the global row index 64+..: same low bits)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDSP(dstB), 16, blk_base + (unsigned)(64 + row) * 1024u + ch * 16u, c0, 0, 0);
      }
    }
  };
  auto commit = [&](int buf) {                       // REG only: registers -> LDS
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
      const int row = tid / KC + i * RPP, ch = tid % KC;
      *reinterpret_cast<f32x4*>(lds + buf * TILE + row * BKP + ch * 4) = ra[i];
      *reinterpret_cast<f32x4*>(lds + buf * TILE + (64 + row) * BKP + ch * 4) = rb[i];
    }
  };
  auto ks_ptr = [&](int buf, int row) -> const float* {
    const int chunk = wave * 2 + fk;
    if (MODE == 0) return lds + buf * TILE + row * BKP + chunk * 4;
    const int sw = (row >> 1) & 7;
    return lds + buf * TILE + row * BK + ((chunk ^ sw) * 4);
  };
  auto compute = [&](int buf) {
    if (KSPLIT) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(ks_ptr(buf, frow)), a1 = *reinterpret_cast<const f32x4*>(ks_ptr(buf, 32 + frow));
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(ks_ptr(buf, 64 + frow)), b1 = *reinterpret_cast<const f32x4*>(ks_ptr(buf, 96 + frow));
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b0[t], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b1[t], acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b0[t], acc3, 0, 0, 0);
        acc4 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b1[t], acc4, 0, 0, 0);
      }
      return;
    }
#pragma unroll
    for (int kk = 0; kk < BK / 8; kk += 2) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(frag_ptr(buf, 0, kk)), b0 = *reinterpret_cast<const f32x4*>(frag_ptr(buf, 1, kk));
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(frag_ptr(buf, 0, kk + 1)), b1 = *reinterpret_cast<const f32x4*>(frag_ptr(buf, 1, kk + 1));
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b0[t], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b1[t], acc2, 0, 0, 0);
      }
    }
  };

  if (MODE == 0) {
    issue(0, 0); commit(0); __syncthreads();
    int cur = 0;
    for (int it = 0; it < iters; ++it) {
      issue(it + 1, cur ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      compute(cur);
      __builtin_amdgcn_sched_barrier(0);
      commit(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
  } else if (MODE == 1) {
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int cur = 0;
    for (int it = 0; it < iters; ++it) {
      issue(it + 1, cur ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      compute(cur);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      cur ^= 1;
    }
  } else {
    constexpr int PER = 2 * (64 * KC / 64 / 4);     // DMA instructions per tile per wave
    issue(0, 0); issue(1, 1);
    if (PER == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int cur = 0;
    for (int it = 0; it < iters; ++it) {
      int nx2 = cur + 2; if (nx2 >= 3) nx2 -= 3;
      issue(it + 2, nx2);                            // buffer nx2 was read in iteration it-1: every wave passed the barrier since
      __builtin_amdgcn_sched_barrier(0);
      compute(cur);
      __builtin_amdgcn_sched_barrier(0);
      if (PER == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // tile it+1 landed (own pieces)
      __builtin_amdgcn_s_barrier();
      cur = cur + 1 == 3 ? 0 : cur + 1;
    }
  }
  // result: sum + a checksum cell per block so the variants can be compared for equality
  float s = 0.f;
  for (int e = 0; e < 16; ++e) s += acc[e] + acc2[e] + acc3[e] + acc4[e];
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) atomicAdd(out + blockIdx.x, s);
}

// XB: the register-staged BK32 loop with the MFMA block split AROUND the barrier.  The fragments of the first 16 k of tile i+1 are
// read right after the barrier of iteration i while the last four MFMAs of tile i (operands already in registers) are still to be
// issued, and the LDS writes of tile i+1 sit between the two MFMA blocks — a wave always has independent MFMAs queued across its
// ds_write -> barrier -> ds_read turnaround.  Same arithmetic as kloop<32, 0> (same accumulation order per accumulator).
__global__ __launch_bounds__(256) void kloop_xb(float* out, const float* __restrict__ src, int src_bytes, int iters) {
  constexpr int BK = 32, KC = 8, BKP = 36, TILE = 128 * BKP;
  __shared__ __attribute__((aligned(16))) float lds[2 * TILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, src_bytes, 0x00020000);
  const unsigned blk_base = (unsigned)(blockIdx.x % 61) * 128u * 1024u;
  f32x16 acc, acc2;
  for (int e = 0; e < 16; ++e) { acc[e] = 0.f; acc2[e] = 0.f; }
  const int frow = lane & 31, fk = lane >> 5;
  const float* fa = lds + (wm * 32 + frow) * BKP + fk * 4;            // + buf * TILE + kk * 8
  const float* fb = lds + (64 + wn * 32 + frow) * BKP + fk * 4;
  constexpr int RPP = 256 / KC, PASSES = 64 / RPP;                    // 32 rows per pass, 2 passes
  f32x4 ra[PASSES], rb[PASSES];
  auto issue = [&](int it) {
    const unsigned c0 = (unsigned)((it * BK) & 255) * 4u;
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
      const unsigned row = tid / KC + i * RPP, ch = tid % KC;
      ra[i] = buf_load4(rs, blk_base + row * 1024u + ch * 16u, c0);
      rb[i] = buf_load4(rs, blk_base + (64u + row) * 1024u + ch * 16u, c0);
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
      const int row = tid / KC + i * RPP, ch = tid % KC;
      *reinterpret_cast<f32x4*>(lds + buf * TILE + row * BKP + ch * 4) = ra[i];
      *reinterpret_cast<f32x4*>(lds + buf * TILE + (64 + row) * BKP + ch * 4) = rb[i];
    }
  };
  f32x4 a0, b0, a1, b1, a2, b2, a3, b3;                               // k groups 0,1 (R0) and 2,3 (R1)
  auto read_h0 = [&](int buf) {
    a0 = *reinterpret_cast<const f32x4*>(fa + buf * TILE); b0 = *reinterpret_cast<const f32x4*>(fb + buf * TILE);
    a1 = *reinterpret_cast<const f32x4*>(fa + buf * TILE + 8); b1 = *reinterpret_cast<const f32x4*>(fb + buf * TILE + 8);
  };
  auto read_h1 = [&](int buf) {
    a2 = *reinterpret_cast<const f32x4*>(fa + buf * TILE + 16); b2 = *reinterpret_cast<const f32x4*>(fb + buf * TILE + 16);
    a3 = *reinterpret_cast<const f32x4*>(fa + buf * TILE + 24); b3 = *reinterpret_cast<const f32x4*>(fb + buf * TILE + 24);
  };
  issue(0); commit(0); __syncthreads();
  read_h0(0);
  int cur = 0;
  for (int it = 0; it < iters; ++it) {
    issue(it + 1);
    read_h1(cur);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b0[t], acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b1[t], acc2, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    commit(cur ^ 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[t], b2[t], acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a3[t], b3[t], acc2, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    read_h0(cur ^ 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 2; t < 4; ++t) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[t], b2[t], acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a3[t], b3[t], acc2, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    cur ^= 1;
  }
  float sum = 0.f;
  for (int e = 0; e < 16; ++e) sum += acc[e] + acc2[e];
  for (int o = 32; o; o >>= 1) sum += __shfl_xor(sum, o);
  if (lane == 0) atomicAdd(out + blockIdx.x, sum);
}

int main(int argc, char** argv) {
  const int src_floats = 64 * 128 * 256 + 4096;
  std::vector<float> h(src_floats);
  unsigned x = 12345u;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (float)(int)(x >> 16) * (1.f / 65536.f) - 0.5f; }
  float *src, *out; hipMalloc(&src, src_floats * 4); hipMalloc(&out, 4 << 20);
  hipMemcpy(src, h.data(), src_floats * 4, hipMemcpyHostToDevice);
  auto run = [&](const char* name, auto kern, int g) {
    const int iters = 4000, blocks = 256 * g;
    hipMemset(out, 0, 4 << 20);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, src, src_floats * 4, 9);   // 9 iterations: checksum run
    hipDeviceSynchronize();
    float chk[2]; hipMemcpy(chk, out, 8, hipMemcpyDeviceToHost);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, src, src_floats * 4, iters);
    hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, src, src_floats * 4, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    return printf("%-28s %d wg/CU  %7.3f ms  checksum %.6e %.6e\n", name, g, ms, chk[0], chk[1]), ms;
  };
  auto tf = [&](float ms, int bk, int g) { printf("      -> %6.1f TFLOP/s\n", 2.0 * 2048 * (bk / 2) * 4000.0 * 4 * 256 * g / (ms * 1e-3) / 1e12); };
  for (int g : {4, 5, 6}) {
    tf(run("BK16 REG", kloop<16, 0>, g), 16, g);
    tf(run("BK16 DMA (2 buf)", kloop<16, 1>, g), 16, g);
    tf(run("BK16 DMA3 (3 buf)", kloop<16, 2>, g), 16, g);
  }
  for (int g : {1, 2, 3, 4}) {
    tf(run("BK32 REG cross-barrier MFMA split", kloop_xb, g), 32, g);
    tf(run("BK32 REG", kloop<32, 0>, g), 32, g);
    tf(run("BK32 REG wave-k-split", kloop<32, 0, true>, g), 32, g);
    tf(run("BK32 DMA wave-k-split", kloop<32, 1, true>, g), 32, g);
  }
  for (int g : {4, 5}) {
    tf(run("BK32 DMA (2 buf)", kloop<32, 1>, g), 32, g);
  }
  tf(run("BK32 DMA3 (3 buf)", kloop<32, 2>, 3), 32, 3);
  return 0;
}
