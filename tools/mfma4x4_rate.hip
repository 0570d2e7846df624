// Issue rate of v_mfma_f32_4x4x1_16b_f32 (A-broadcast, NG independent accumulators) with one wave per SIMD, optionally fed by a
// ds_read_b128 per 4 k like lstm_persist.hip's inner loop.  Prints ns and (at the measured wall time) cycles @2.4 GHz per MFMA.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NG, bool LDS>
__global__ __launch_bounds__(256) void rate(float* out, int iters, float bval) {
  __shared__ float wl[16384];
  for (int i = threadIdx.x; i < 16384; i += 256) wl[i] = 1.0f / (1 + i);
  __syncthreads();
  f32x4 acc[NG];
  for (int g = 0; g < NG; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int lane = threadIdx.x & 63;
  float a = 1.f + lane;
  for (int it = 0; it < iters; ++it) {
    f32x4 a4 = {a, a, a, a};
    if (LDS) a4 = *reinterpret_cast<const f32x4*>(wl + (((it & 511) * 8 + (lane & 7)) * 4));
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#define G(X) if constexpr (NG > X) acc[X] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[kk], bval, acc[X], 4, X, 0);
      G(0) G(1) G(2) G(3) G(4) G(5) G(6) G(7)
#undef G
    }
  }
  float s = 0.f;
  for (int g = 0; g < NG; ++g) s += acc[g][0] + acc[g][1] + acc[g][2] + acc[g][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NG, bool LDS>
void run(const char* name) {
  float* d; hipMalloc(&d, 256 * 256 * 4);
  const int iters = 20000;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((rate<NG, LDS>), dim3(256), dim3(256), 0, 0, d, 100, 0.5f);
  hipEventRecord(a);
  hipLaunchKernelGGL((rate<NG, LDS>), dim3(256), dim3(256), 0, 0, d, iters, 0.5f);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double per = ms * 1e6 / ((double)iters * 4 * NG);
  printf("%-28s NG=%d  %.2f ns per MFMA per wave (%.1f cycles @2.4GHz)  => %.1f TFLOP/s on 1024 SIMDs\n", name, NG, per, per * 2.4,
         512.0 / per * 1024 / 1e3);
  hipFree(d);
}

int main() {
  run<1, false>("regs"); run<2, false>("regs"); run<5, false>("regs"); run<8, false>("regs");
  run<2, true>("ds_read_b128 per 4k"); run<5, true>("ds_read_b128 per 4k");
  return 0;
}
