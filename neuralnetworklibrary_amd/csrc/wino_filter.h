// The filter transform of the 2-D F(2x2, 3x3) kernels, shared by wino2.hip (per-call pre-pass), wino2s.hip and wino.hip (nnl_wino_filter_multi).
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

// filt [Nc][3][3][C] -> U tiled [ceil(Nc/64)][4 xi][C/8][4 nu][128 slots][4] (csrc/wino2s.hip); rows >= Nc are zero.  One block of 256
// threads per (64-row tile, 16 channels): thread -> row kl = t & 63, channel quad t >> 6, sixteen 16-B stores each — consecutive lanes
// write consecutive slots (2 KB runs per position and channel block; the first version, one (k, c) item per thread, wrote 4-byte pieces
// 8 KB apart: 86 us per ResNet-34 pass against 34 for the [rows][16][ch] layout).  blk = tile * ceil(C / 16) + channel group.
__device__ __forceinline__ void wino2s_filter_block(const float* __restrict__ src, float* __restrict__ dst, long blk, int t, int Nc, int C, int flip) {
  const int cg = (C + 15) / 16;
  const long tn = blk / cg;
  const int c4 = (int)(blk - tn * cg) * 4 + (t >> 6), kl = t & 63;
  if (c4 * 4 >= C) return;
  const long k = tn * 64 + kl;
  float v[4][16];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (k < Nc) wino2_filter_vals(src + k * 9 * C + c4 * 4 + e, C, flip, v[e]);
    else {
#pragma unroll
      for (int i = 0; i < 16; ++i) v[e][i] = 0.f;
    }
  }
  const long b = c4 >> 1, NB = C >> 3;
  const int slot = 2 * kl + ((c4 & 1) ^ ((kl >> 3) & 1));
#pragma unroll
  for (int xi = 0; xi < 4; ++xi)
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
      float4 o = {v[0][xi * 4 + nu], v[1][xi * 4 + nu], v[2][xi * 4 + nu], v[3][xi * 4 + nu]};
      *reinterpret_cast<float4*>(dst + (((((tn * 4 + xi) * NB + b) * 4 + nu) * 128 + slot) << 2)) = o;
    }
}
