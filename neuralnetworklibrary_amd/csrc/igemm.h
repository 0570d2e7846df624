// Implicit-GEMM building blocks on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) for gfx950.
//
// Two kernels cover every dense contraction on the Learner.fit() hot path:
//
//  igemm_rowk  : C[M][Nc] = A[M][Kg] * B[Nc][Kg]^T, both operands "k-contiguous" in memory.
//                A rows are GATHERED: row m = output pixel (n,p,q), k = (r,s,c) -> NHWC input pixel
//                (MODE_FWD), or row m = input pixel (n,h,w), k = (r,s,co) -> dy pixel with the stride
//                divisibility predicate (MODE_DGRAD).  R=S=1,H=W=1 degenerates to a plain NT GEMM (Linear).
//  igemm_kmajor: C[Mc][Nc] = sum_k A[k][Mc] * B[k][Nc], both operands "k-major" (rows indexed by the
//                reduction index = pixel / sample, channels contiguous): conv wgrad and Linear dW, with
//                split-K over the (huge) pixel dimension and a deterministic partial-slab reduction.
//
// MFMA operand maps (cdna_hip_programming.md §3): for 32x32x2 lane l holds A[i=l&31][k=l>>5] and
// B[k=l>>5][j=l&31]; C/D: col j = l&31, row i = (reg&3) + 8*(reg>>2) + 4*(l>>5).  The k index inside one
// MFMA is only a summation label, so in igemm_rowk a lane reads FOUR consecutive k with one ds_read_b128
// (lanes 0-31: k..k+3, lanes 32-63: k+4..k+7) and feeds MFMA t with element t: the two lane halves then sum
// k+t and k+4+t, and four MFMAs cover the 8-wide k group — A and B use the same assignment, so it is exact.
#pragma once
#include "nnl_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct IgemmRowkParams {
  const float* a;      // gathered operand: NHWC tensor [N][H][W][C]
  const float* b;      // [Nc][Kg] row-major (k contiguous)
  float* y;            // [M][Nc] row-major
  const float* bias;   // [Nc] or null
  const float* add;    // [M][Nc] addend (same layout as y) or null: y = acc + bias + add
  int N, H, W, C;      // dims of the A-source tensor
  int P, Q;            // spatial dims enumerated by GEMM rows: M = N*P*Q
  int R, S, stride, pad;
  int M, Nc, Kg;       // Kg = R*S*C
  int relu;
  int grid_m, grid_n;
};

struct IgemmKmajorParams {
  const float* a;      // [Kp][Mc] : dy, rows = output pixels (n,p,q), Mc = Cout
  const float* b;      // NHWC x [N][H][W][C]; B[k=(n,p,q)][j=(r,s,c)] gathered
  float* y;            // [Mc][Nc] (Nc = R*S*C) when splits==1, else partial slabs [splits][Mc][Nc]
  int N, H, W, C;
  int P, Q;
  int R, S, stride, pad;
  int Mc, Nc;          // Mc = Cout, Nc = R*S*C
  long Kp;             // N*P*Q
  int splits, k_per_split;   // k_per_split multiple of BK
  int grid_m, grid_n;
};

// Plain NT GEMM on the igemm_rowk kernel (defined in conv2d.hip): y[M][N] = a[M][K] * b[N][K]^T (+bias[N]) (+add[M][N])
// (ReLU).  K % 4 == 0.  Used by the per-timestep recurrent GEMMs of the LSTM (lstm.hip).
int nnl_internal_gemm_nt(const float* a, const float* b, float* y, const float* bias, const float* add, int M, int N,
                         int K, int relu, hipStream_t s);
int nnl_internal_gemm_nt_splitk(const float* a, const float* b, float* y_slabs, int M, int N, int K, int splits, hipStream_t s);
// y[Mc][Nc] = sum_k a[k][Mc] * b[k][Nc]  (both k-major; Mc % 4 == 0, Nc % 4 == 0); split-K workspace as for wgrad.
size_t nnl_internal_gemm_tn_workspace_bytes(int Mc, int Nc, long Kp);
int nnl_internal_gemm_tn(const float* a, const float* b, float* y, int Mc, int Nc, long Kp, void* ws, size_t ws_bytes,
                         hipStream_t s);

// bijective XCD-aware remap: blocks that share an XCD (equal bid % 8) get a contiguous range of logical ids
__device__ __forceinline__ int nnl_xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}
