// The filter transform of the 2-D F(2x2, 3x3) kernels, shared by wino2.hip (per-call pre-pass) and wino.hip (nnl_wino_filter_multi).
#pragma once
#include <hip/hip_runtime.h>

// one (k, c) filter: src[(r * 3 + s) * C] -> v[xi * 4 + nu] = (G g G^T)[xi][nu]; flip: read filt[.][2-r][2-s][.] (the dgrad filter)
__device__ __forceinline__ void wino2_filter_vals(const float* __restrict__ src, long C, int flip, float (&v)[16]) {
  float g[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int s = 0; s < 3; ++s) g[r][s] = src[((flip ? 2 - r : r) * 3 + (flip ? 2 - s : s)) * C];
  float t[4][3];                                      // G g
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    t[0][s] = g[0][s];
    t[1][s] = 0.5f * (g[0][s] + g[1][s] + g[2][s]);
    t[2][s] = 0.5f * (g[0][s] - g[1][s] + g[2][s]);
    t[3][s] = g[2][s];
  }
#pragma unroll
  for (int xi = 0; xi < 4; ++xi) {
    v[xi * 4 + 0] = t[xi][0];
    v[xi * 4 + 1] = 0.5f * (t[xi][0] + t[xi][1] + t[xi][2]);
    v[xi * 4 + 2] = 0.5f * (t[xi][0] - t[xi][1] + t[xi][2]);
    v[xi * 4 + 3] = t[xi][2];
  }
}

// filt [Nc][3][3][C] -> U [Nc][16][C]
__device__ __forceinline__ void wino2_filter_item(const float* __restrict__ src, float* __restrict__ dst, long C, int flip) {
  float v[16];
  wino2_filter_vals(src, C, flip, v);
#pragma unroll
  for (int i = 0; i < 16; ++i) dst[i * C] = v[i];
}
