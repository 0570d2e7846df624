// Micro-benchmark: do VALU instructions of one wave overlap with the fp32 MFMA chain of ANOTHER wave on the same SIMD of an MI355X?
// One workgroup of 8 waves per CU (two per SIMD).  `mfma_mask` / `valu_mask` pick (by wave index bit) which waves run a chain of dependent
// v_mfma_f32_32x32x2_f32 (16 per iteration: one k step of the 2-D Winograd kernel) and which run `nv` dependent-free v_pk_fma_f32 per
// iteration (its input transform is ~60); the other waves exit at once.  If the two overlap, time(both) = max; if the SIMD issues them
// exclusively, time(both) = sum.  Build: hipcc --offload-arch=gfx950 -O3 tools/coissue_probe.hip -o tools/coissue_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NV, int KIND, bool BF16 = false, int NOPS = 0>
__global__ __launch_bounds__(512) void probe(float* out, int iters, float seed, unsigned mfma_mask, unsigned valu_mask) {
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  float s = 0.f;
  if ((mfma_mask >> wave) & 1) {
    f32x16 acc, acc2;
    for (int e = 0; e < 16; ++e) { acc[e] = 0.f; acc2[e] = 0.f; }
    if (BF16) {                                            // the bf16 matrix core: 16 dependent v_mfma_f32_32x32x16_bf16
      bf16x8 a, b;
      for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(seed + e); b[e] = (__bf16)(seed - e); }
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 16; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
      }
    } else {
      float a = seed, b = -seed;
      float sv[4] = {seed, seed + 1.f, seed + 2.f, seed + 3.f};
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          if (NOPS == 100 && (t & 1)) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc2, 0, 0, 0);   // two independent chains
          else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
          // NOPS x 16 idle cycles of THIS wave behind each MFMA (64 cycles in the pipe): does the wave waiting at issue for its
          // dependent MFMA hold the SIMD's issue port against the other wave?
          if (NOPS < 0) {                                  // -NOPS independent v_fma_f32 of THIS wave behind each MFMA
#pragma unroll
            for (int q = 0; q < -NOPS; ++q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(sv[q & 3]) : "v"(a), "v"(b));
          }
          if (NOPS >= 1 && NOPS < 100) asm volatile("s_nop 15");
          if (NOPS >= 2 && NOPS < 100) asm volatile("s_nop 15");
          if (NOPS >= 3 && NOPS < 100) asm volatile("s_nop 15");
          if (NOPS >= 4 && NOPS < 100) asm volatile("s_nop 15");
        }
      }
    }
    for (int e = 0; e < 16; ++e) s += acc[e] + acc2[e];
    if (!BF16 && NOPS < 0) s += 1.f;
  } else if ((valu_mask >> wave) & 1) {
    if (KIND == 0) {                                       // v_pk_fma_f32
      f32x2 r[8];
      for (int e = 0; e < 8; ++e) r[e] = f32x2{seed * e, seed};
      const f32x2 m = {seed, 1.f - seed};
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < NV; ++t) r[t & 7] = __builtin_elementwise_fma(r[t & 7], m, r[(t + 3) & 7]);
      }
      for (int e = 0; e < 8; ++e) s += r[e][0] + r[e][1];
    } else if (KIND == 1) {                                // v_fma_f32
      float r[8];
      for (int e = 0; e < 8; ++e) r[e] = seed * e + threadIdx.x;
      const float m = 1.f - seed;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < NV; ++t) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[t & 7]) : "v"(m), "v"(r[(t + 3) & 7]));   // (asm: the compiler packs plain fmaf pairs into v_pk_fma_f32)
      }
      for (int e = 0; e < 8; ++e) s += r[e];
    } else {                                               // integer: v_add3_u32 / v_xor
      unsigned r[8];
      for (int e = 0; e < 8; ++e) r[e] = (unsigned)(seed * e) + threadIdx.x;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < NV; ++t) r[t & 7] = (r[t & 7] ^ r[(t + 3) & 7]) + (unsigned)it;
      }
      for (int e = 0; e < 8; ++e) s += (float)r[e];
    }
  }
  if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int NV, int KIND, bool BF16 = false, int NOPS = 0>
float run(float* out, int iters, unsigned mm, unsigned vm) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((probe<NV, KIND, BF16, NOPS>), dim3(256), dim3(512), 0, 0, out, iters, 0.f, mm, vm);
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((probe<NV, KIND, BF16, NOPS>), dim3(256), dim3(512), 0, 0, out, iters, 0.f, mm, vm);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms / 5 * 1e3f / iters;     // us per iteration
}

int main() {
  float* out;
  hipMalloc(&out, 4096);
  const int iters = 4000;
  struct { const char* name; unsigned mm, vm; } cases[] = {
      {"mfma on waves 0-3 only", 0x0F, 0}, {"mfma on waves 0,2,4,6 (two per SIMD if wave i -> SIMD i % 4)", 0x55, 0},
      {"mfma on waves 0,1,4,5", 0x33, 0}, {"mfma on all 8 waves", 0xFF, 0},
      {"valu on waves 4-7 only", 0, 0xF0}, {"valu on all 8 waves", 0, 0xFF},
      {"mfma on waves 0-3 + valu on waves 4-7", 0x0F, 0xF0}, {"mfma on waves 0,1,4,5 + valu on waves 2,3,6,7", 0x33, 0xCC}};
  static const char* kinds[] = {"v_pk_fma_f32", "v_fma_f32", "integer xor + add"};
  for (int kind = 0; kind < 3; ++kind) {
    printf("VALU kind: %s\n", kinds[kind]);
    for (auto& c : cases) {
      if (kind > 0 && c.vm == 0) continue;
      float t[3];
      if (kind == 0) { t[0] = run<32, 0>(out, iters, c.mm, c.vm); t[1] = run<64, 0>(out, iters, c.mm, c.vm); t[2] = run<128, 0>(out, iters, c.mm, c.vm); }
      if (kind == 1) { t[0] = run<32, 1>(out, iters, c.mm, c.vm); t[1] = run<64, 1>(out, iters, c.mm, c.vm); t[2] = run<128, 1>(out, iters, c.mm, c.vm); }
      if (kind == 2) { t[0] = run<32, 2>(out, iters, c.mm, c.vm); t[1] = run<64, 2>(out, iters, c.mm, c.vm); t[2] = run<128, 2>(out, iters, c.mm, c.vm); }
      printf("  %-68s  nv=32: %.3f us/iter   nv=64: %.3f   nv=128: %.3f\n", c.name, t[0], t[1], t[2]);
    }
  }
  printf("fp32 MFMA with n x s_nop 15 behind each (VALU kind v_fma_f32, 64 per iteration): [mfma on waves 0-3 only] / [+ valu on waves 4-7]\n");
  printf("  n=0: %.3f / %.3f   n=1: %.3f / %.3f   n=2: %.3f / %.3f   n=3: %.3f / %.3f   n=4: %.3f / %.3f\n",
         run<64, 1, false, 0>(out, iters, 0x0F, 0), run<64, 1, false, 0>(out, iters, 0x0F, 0xF0), run<64, 1, false, 1>(out, iters, 0x0F, 0),
         run<64, 1, false, 1>(out, iters, 0x0F, 0xF0), run<64, 1, false, 2>(out, iters, 0x0F, 0), run<64, 1, false, 2>(out, iters, 0x0F, 0xF0),
         run<64, 1, false, 3>(out, iters, 0x0F, 0), run<64, 1, false, 3>(out, iters, 0x0F, 0xF0), run<64, 1, false, 4>(out, iters, 0x0F, 0),
         run<64, 1, false, 4>(out, iters, 0x0F, 0xF0));
  printf("fp32 MFMA with n independent v_fma_f32 of the SAME wave behind each: [mfma on waves 0-3 only] / [+ 64 v_fma_f32 per iteration on waves 4-7]\n");
  printf("  n=0: %.3f / %.3f   n=4: %.3f / %.3f   n=8: %.3f / %.3f   n=12: %.3f / %.3f   n=16: %.3f / %.3f\n",
         run<64, 1, false, 0>(out, iters, 0x0F, 0), run<64, 1, false, 0>(out, iters, 0x0F, 0xF0), run<64, 1, false, -4>(out, iters, 0x0F, 0),
         run<64, 1, false, -4>(out, iters, 0x0F, 0xF0), run<64, 1, false, -8>(out, iters, 0x0F, 0), run<64, 1, false, -8>(out, iters, 0x0F, 0xF0),
         run<64, 1, false, -12>(out, iters, 0x0F, 0), run<64, 1, false, -12>(out, iters, 0x0F, 0xF0), run<64, 1, false, -16>(out, iters, 0x0F, 0),
         run<64, 1, false, -16>(out, iters, 0x0F, 0xF0));
  printf("fp32 MFMA on TWO independent accumulators alternating: [mfma on waves 0-3 only] / [+ 64 v_fma_f32 per iteration on waves 4-7] / [+ 128]\n");
  printf("  %.3f / %.3f / %.3f\n", run<64, 1, false, 100>(out, iters, 0x0F, 0), run<64, 1, false, 100>(out, iters, 0x0F, 0xF0),
         run<128, 1, false, 100>(out, iters, 0x0F, 0xF0));
  printf("MFMA kind: v_mfma_f32_32x32x16_bf16 (the bf16 matrix core), VALU kind v_fma_f32\n");
  for (auto& c : cases) {
    const float t0 = run<32, 1, true>(out, iters, c.mm, c.vm), t1 = run<64, 1, true>(out, iters, c.mm, c.vm), t2 = run<128, 1, true>(out, iters, c.mm, c.vm);
    printf("  %-68s  nv=32: %.3f us/iter   nv=64: %.3f   nv=128: %.3f\n", c.name, t0, t1, t2);
  }
  return 0;
}
