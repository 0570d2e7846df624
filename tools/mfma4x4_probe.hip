// Probe: operand / result lane mapping of v_mfma_f32_4x4x1_16b_f32 on gfx950, with and without A-broadcast (CBSZ / ABID).
// For one-hot A (lane la) and one-hot B (lane lb) prints every (lane, vgpr) of D that becomes non-zero.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CBSZ, int ABID>
__global__ void probe(int la, int lb, float* out) {
  const int l = threadIdx.x;
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(l == la ? 1.f : 0.f, l == lb ? 1.f : 0.f, c, CBSZ, ABID, 0);
  for (int v = 0; v < 4; ++v) out[l * 4 + v] = c[v];
}

template <int CBSZ, int ABID>
void run(int la, int lb) {
  float* d; hipMalloc(&d, 256 * 4);
  hipLaunchKernelGGL((probe<CBSZ, ABID>), dim3(1), dim3(64), 0, 0, la, lb, d);
  float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("cbsz %d abid %d  A lane %2d  B lane %2d ->", CBSZ, ABID, la, lb);
  for (int i = 0; i < 256; ++i) if (h[i] != 0.f) printf(" (lane %d, v %d)", i / 4, i % 4);
  printf("\n");
  hipFree(d);
}

int main() {
  const int las[] = {0, 1, 5, 21}, lbs[] = {0, 2, 7, 22, 63};
  for (int la : las) for (int lb : lbs) run<0, 0>(la, lb);
  for (int la : las) for (int lb : lbs) run<4, 0>(la, lb);
  for (int la : las) for (int lb : lbs) run<4, 1>(la, lb);
  for (int la : las) for (int lb : lbs) run<4, 5>(la, lb);
  return 0;
}
