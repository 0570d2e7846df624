// Ceiling probe for an fp32-ACCURATE GEMM inner loop on the bf16 matrix cores: every fp32 operand is split into three bf16
// terms (hi + mid + lo = 24 mantissa bits) and a product keeps the six most significant cross terms (hh, hm, mh, hl, lh, mm:
// relative error ~3e-7, below the fp32 accumulation error itself — numerics check in profiles/README.md).  One k-block of 16 then
// costs 6 x v_mfma_f32_32x32x16_bf16 (6 x 32 cycles... vs 8 x 64 cycles of v_mfma_f32_32x32x2_f32) per 32x32 output tile.
// This probe only measures what the LDS -> MFMA part of such a loop sustains (operands pre-split and resident in LDS, random
// data, no global loads): FP32-EQUIVALENT TFLOP/s = 2*M*N*K / time.
// hipcc --offload-arch=gfx950 -O3 tools/splitbf16_probe.hip -o tools/splitbf16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// LDS image per split: [rows][16] bf16 = 32 B per row; fragment (row r, k half h) = 16 B at r*32 + h*16
template <int TW, int TERMS>     // TW: wave tile = (32*TW) x (32*TW); TERMS 3 or 6
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
  constexpr int ROWS = 64 * TW;                                // block tile rows (= cols): 2x2 waves
  __shared__ __attribute__((aligned(16))) unsigned short lds[2][3][2 * ROWS * 16];   // [buf][split][A rows then B rows][16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  for (int i = tid; i < 2 * 3 * 2 * ROWS * 16; i += 256) {
    unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u; h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    (&lds[0][0][0])[i] = (unsigned short)(0x3F00u + (h & 0xFF) + ((h >> 8) & 1) * 0x8000u);    // random bf16 in +-[0.5, 1)
  }
  __syncthreads();
  f32x16 acc[TW][TW];
  for (int i = 0; i < TW; ++i) for (int j = 0; j < TW; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int fo = (lane & 31) * 16 + (lane >> 5) * 8;           // element offset of this lane's fragment inside a 32-row group
  for (int it = 0; it < iters; ++it) {
    const int buf = it & 1;
    bf16x8 a[3][TW], b[3][TW];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
#pragma unroll
      for (int i = 0; i < TW; ++i) a[s][i] = *reinterpret_cast<const bf16x8*>(&lds[buf][s][(wm * TW + i) * 32 * 16 + fo]);
#pragma unroll
      for (int j = 0; j < TW; ++j) b[s][j] = *reinterpret_cast<const bf16x8*>(&lds[buf][s][ROWS * 16 + (wn * TW + j) * 32 * 16 + fo]);
    }
#pragma unroll
    for (int i = 0; i < TW; ++i)
#pragma unroll
      for (int j = 0; j < TW; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0][j], acc[i][j], 0, 0, 0);   // hh
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1][j], acc[i][j], 0, 0, 0);   // hm
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0][j], acc[i][j], 0, 0, 0);   // mh
        if (TERMS == 6) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[2][j], acc[i][j], 0, 0, 0); // hl
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][i], b[0][j], acc[i][j], 0, 0, 0); // lh
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[1][j], acc[i][j], 0, 0, 0); // mm
        }
      }
  }
  float s = 0.f;
  for (int i = 0; i < TW; ++i) for (int j = 0; j < TW; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  if (s == 12345.678f) out[tid] = s;
}

template <typename K>
void run(const char* name, K kern, int tw, int g, float* out) {
  const int iters = 20000, blocks = 256 * g;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 100);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = 2.0 * (64.0 * tw) * (64.0 * tw) * 16.0 * iters * blocks;      // fp32-equivalent work of the block tiles
  printf("%-40s %d wg/CU  %8.3f ms  %7.1f fp32-equivalent TFLOP/s\n", name, g, ms, flop / (ms * 1e-3) / 1e12);
}

int main() {
  float* out; hipMalloc(&out, 1 << 20);
  for (int g : {1, 2, 4}) {
    run("wave tile 32x32, 6 terms (fp32-accurate)", probe<1, 6>, 1, g, out);
    run("wave tile 32x32, 3 terms (~1e-4)", probe<1, 3>, 1, g, out);
  }
  for (int g : {1, 2}) {
    run("wave tile 64x64, 6 terms (fp32-accurate)", probe<2, 6>, 2, g, out);
    run("wave tile 64x64, 3 terms (~1e-4)", probe<2, 3>, 2, g, out);
  }
  return 0;
}
