// Micro-benchmark: what does the fp32 MFMA pipe of one MI355X sustain, as a function of waves per SIMD, accumulators per
// wave (dependent chain length) and whether the operands come from ds_read_b128?  Build: hipcc --offload-arch=gfx950 -O3
// tools/mfma_probe.hip -o /tmp/mfma_probe ; run on the GPU box.  Output: one line per variant with TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool LDS, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void probe32(float* out, int iters, float seed) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 64 * 36];
  const int lane = threadIdx.x & 63;
  if (LDS) for (int i = threadIdx.x; i < 2 * 64 * 36; i += blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + (unsigned)blockIdx.x * 40503u; h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    lds[i] = seed > 1.5f ? (float)(int)(h & 0xFFFF) * (1.f / 65536.f) - 0.5f : seed * (float)(i & 7);   // seed 2: random operands
  }
  __syncthreads();
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a) for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  f32x4 fa = {seed, seed * 2, seed * 3, seed * 4}, fb = {seed, -seed, seed, -seed};
  const float* As = lds + (lane & 31) * 36 + (lane >> 5) * 4;
  const float* Bs = As + 64 * 36;
  for (int it = 0; it < iters; ++it) {
    // one "k group pair": 8 MFMAs (like BK=16 of the conv kernel), optionally fed by 4 ds_read_b128
    f32x4 a0 = fa, b0 = fb, a1 = fa, b1 = fb;
    if (LDS) {
      const int o = (it & 1) * 16;
      a0 = *reinterpret_cast<const f32x4*>(As + o); b0 = *reinterpret_cast<const f32x4*>(Bs + o);
      a1 = *reinterpret_cast<const f32x4*>(As + o + 8); b1 = *reinterpret_cast<const f32x4*>(Bs + o + 8);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b0[t], acc[0], 0, 0, 0);
      acc[NACC > 1 ? 1 : 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b1[t], acc[NACC > 1 ? 1 : 0], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int a = 0; a < NACC; ++a) for (int e = 0; e < 16; ++e) s += acc[a][e];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

// 16x16x4 f32: 8 passes (32 cycles), 4 accumulator VGPRs
template <int NACC, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void probe16(float* out, int iters, float seed) {
  f32x4 acc[NACC];
  for (int a = 0; a < NACC; ++a) for (int e = 0; e < 4; ++e) acc[a][e] = 0.f;
  float fa = seed, fb = -seed;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 16; ++t)     // 16 x (16*16*4) = 8 x (32*32*2) FMAs: same work per iteration as probe32
      acc[t % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc[t % NACC], 0, 0, 0);
  }
  float s = 0.f;
  for (int a = 0; a < NACC; ++a) for (int e = 0; e < 4; ++e) s += acc[a][e];
  if (s == 12345.678f) out[threadIdx.x] = s;
}


// The conv kernel's k loop, ingredient by ingredient (64x64 tile, 4 waves, BK=16, double-buffered LDS):
//   STAGE 1: ds_read_b128 x4 + 8 MFMA                      (as probe32<2,true>)
//   STAGE 2: + 2 ds_write_b128 of register data + s_barrier per iteration
//   STAGE 3: + 2 buffer_load_dwordx4 per thread per iteration from an L2-resident array feeding those writes
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int STAGE>
__global__ __launch_bounds__(256) void probe_loop(float* out, const float* __restrict__ src, int iters, float seed) {
  constexpr int BKP = 20;
  __shared__ __attribute__((aligned(16))) float lds[2][128 * BKP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  for (int i = tid; i < 2 * 128 * BKP; i += 256) {
    unsigned h = (unsigned)i * 2654435761u + (unsigned)blockIdx.x * 40503u; h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    (&lds[0][0])[i] = (float)(int)(h & 0xFFFF) * (1.f / 65536.f) - 0.5f;
  }
  __syncthreads();
  f32x16 acc, acc2;
  for (int e = 0; e < 16; ++e) { acc[e] = 0.f; acc2[e] = 0.f; }
  const int kc = tid & 3, lrow = tid >> 2;
  const int frag = (lane & 31) * BKP + (lane >> 5) * 4;
  f32x4 ra = {seed, seed, seed, seed}, rb = ra;
  const f32x4* g = reinterpret_cast<const f32x4*>(src) + (blockIdx.x & 63) * 4096 + tid;
  int cur = 0;
  for (int it = 0; it < iters; ++it) {
    if (STAGE >= 3) { ra = g[(it & 7) * 256]; rb = g[2048 + (it & 7) * 256]; }
    __builtin_amdgcn_sched_barrier(0);
    const float* As = lds[cur] + wm * 32 * BKP + frag;
    const float* Bs = lds[cur] + 64 * BKP + wn * 32 * BKP + frag;
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(As), b0 = *reinterpret_cast<const f32x4*>(Bs);
    const f32x4 a1 = *reinterpret_cast<const f32x4*>(As + 8), b1 = *reinterpret_cast<const f32x4*>(Bs + 8);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b0[t], acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b1[t], acc2, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (STAGE >= 2) {
      float* Aw = lds[cur ^ 1];
      *reinterpret_cast<f32x4*>(Aw + lrow * BKP + kc * 4) = ra;
      *reinterpret_cast<f32x4*>(Aw + (64 + lrow) * BKP + kc * 4) = rb;
      __syncthreads();
      cur ^= 1;
    }
  }
  float s = 0.f;
  for (int e = 0; e < 16; ++e) s += acc[e] + acc2[e];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

// Shape comparison on RANDOM operands read from LDS (the chip lowers its clock under MFMA load; the clock it holds can depend
// on the shape): one wave computes a 32x32 output tile per k-group of 8 either as 4 x v_mfma_f32_32x32x2 (k pairs) or as
// 2x2 tiles of v_mfma_f32_16x16x4 (2 k-quads): the same FMAs and the same LDS bytes (4 ds_read_b128 per 8 k? see below).
template <int SHAPE>   // 32 or 16
__global__ __launch_bounds__(256) void probe_shape(float* out, int iters, float seed) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 64 * 36];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 2 * 64 * 36; i += blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + (unsigned)blockIdx.x * 40503u; h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    lds[i] = (float)(int)(h & 0xFFFF) * (1.f / 65536.f) - 0.5f;
  }
  __syncthreads();
  float s = 0.f;
  if (SHAPE == 32) {
    f32x16 acc0, acc1;
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
    const float* As = lds + (lane & 31) * 36 + (lane >> 5) * 4;
    const float* Bs = As + 64 * 36;
    for (int it = 0; it < iters; ++it) {          // 16 k per iteration: 8 MFMA 32x32x2
      const int o = (it & 1) * 16;
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(As + o), b0 = *reinterpret_cast<const f32x4*>(Bs + o);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(As + o + 8), b1 = *reinterpret_cast<const f32x4*>(Bs + o + 8);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b0[t], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b1[t], acc1, 0, 0, 0);
      }
    }
    for (int e = 0; e < 16; ++e) s += acc0[e] + acc1[e];
  } else {
    // 16x16x4: lane l holds A[row l&15][k = l>>4] -> with a b128 read of 4 consecutive k the lane feeds 4 MFMAs whose k sets
    // are {t, 4+t, 8+t, 12+t}: rows (l&15) + 16*i, k chunk (l>>4)*4 of a 16-wide k group: 2 A reads + 2 B reads per 16 k
    f32x4 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
    const float* As = lds + (lane & 15) * 36 + (lane >> 4) * 4;
    const float* Bs = As + 64 * 36;
    for (int it = 0; it < iters; ++it) {          // 16 k per iteration: 4 k-quads x 2x2 tiles = 16 MFMA 16x16x4
      const int o = (it & 1) * 16;
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(As + o), a1 = *reinterpret_cast<const f32x4*>(As + 16 * 36 + o);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(Bs + o), b1 = *reinterpret_cast<const f32x4*>(Bs + 16 * 36 + o);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[t], b0[t], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[t], b1[t], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[t], b0[t], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[t], b1[t], acc[1][1], 0, 0, 0);
      }
    }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 4; ++e) s += acc[i][j][e];
  }
  if (s == 12345.678f) out[threadIdx.x] = s;
}

template <typename K>
void run(const char* name, K kern, int waves, int wg_per_cu, float* out, float seed = 1.0f) {
  const int iters = 20000, blocks = 256 * wg_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(waves * 64), 0, 0, out, 100, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(waves * 64), 0, 0, out, iters, seed);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = 2.0 * 2048 * 8 * (double)iters * waves * blocks;
  printf("%-44s waves/SIMD=%4.1f  %7.3f ms  %6.1f TFLOP/s\n", name, waves * wg_per_cu / 4.0, ms, flop / (ms * 1e-3) / 1e12);
}

int main() {
  float* out; hipMalloc(&out, 1 << 20);
#define R32(NACC, LDS, W, G) run("32x32x2 acc=" #NACC " lds=" #LDS " wg=" #W "w x" #G, probe32<NACC, LDS, W>, W, G, out)
  R32(1, false, 4, 1); R32(1, false, 4, 2); R32(1, false, 4, 4); R32(1, false, 4, 6);
  R32(2, false, 4, 1); R32(2, false, 4, 2); R32(2, false, 4, 4); R32(2, false, 4, 6);
  R32(1, true, 4, 1); R32(1, true, 4, 2); R32(1, true, 4, 4); R32(1, true, 4, 6);
  R32(2, true, 4, 1); R32(2, true, 4, 2); R32(2, true, 4, 4); R32(2, true, 4, 6);
#define R16(NACC, W, G) run("16x16x4 acc=" #NACC " wg=" #W "w x" #G, probe16<NACC, W>, W, G, out)
  R16(1, 4, 1); R16(2, 4, 1); R16(4, 4, 1); R16(1, 4, 4); R16(2, 4, 4); R16(4, 4, 4);
  run("32x32x2 acc=2 lds RANDOM data wg=4w x4", probe32<2, true, 4>, 4, 4, out, 2.0f);
  run("32x32x2 acc=2 lds RANDOM data wg=4w x6", probe32<2, true, 4>, 4, 6, out, 2.0f);
  for (int rep = 0; rep < 2; ++rep) {
    run("SHAPE 32x32x2 random LDS operands wg=4w x4", probe_shape<32>, 4, 4, out);
    run("SHAPE 16x16x4 (2x2) random LDS operands x4", probe_shape<16>, 4, 4, out);
    run("SHAPE 32x32x2 random LDS operands wg=4w x6", probe_shape<32>, 4, 6, out);
    run("SHAPE 16x16x4 (2x2) random LDS operands x6", probe_shape<16>, 4, 6, out);
  }
  float* src; hipMalloc(&src, 64 * 4096 * 16); hipMemset(src, 0, 64 * 4096 * 16);
  for (int g = 4; g <= 6; g += 2) {
    const int iters = 20000, blocks = 256 * g;
    for (int st = 1; st <= 3; ++st) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      auto launch = [&](int n) {
        if (st == 1) hipLaunchKernelGGL(probe_loop<1>, dim3(blocks), dim3(256), 0, 0, out, src, n, 1.0f);
        if (st == 2) hipLaunchKernelGGL(probe_loop<2>, dim3(blocks), dim3(256), 0, 0, out, src, n, 1.0f);
        if (st == 3) hipLaunchKernelGGL(probe_loop<3>, dim3(blocks), dim3(256), 0, 0, out, src, n, 1.0f);
      };
      launch(100); hipDeviceSynchronize();
      hipEventRecord(e0); launch(iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("k-loop stage %d (random LDS data), %d wg/CU:  %7.3f ms  %6.1f TFLOP/s\n", st, g, ms, 2.0 * 2048 * 8 * (double)iters * 4 * blocks / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
