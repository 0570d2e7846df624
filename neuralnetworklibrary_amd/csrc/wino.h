// 3x3 / stride 1 / pad 1 convolutions as a 1-D Winograd F(2,3) along the width, fused into the implicit-GEMM kernel (wino.hip).
#pragma once
#include "nnl_common.h"

// One problem: out[n][h][w][Nc] = conv3x3_pad1(in[n][h][w][Cin], filter) (+ bias[Nc]) (+ add[n][h][w][Nc]) (ReLU when relu == 1).
//   flip == 0: filter = w [Nc][3][3][Cin]                       (convolution forward)
//   flip == 1: filter = wt [Nc][3][3][Cin] read as wt[.][2-r][2-s][.]   (stride-1 dgrad: in = dy, Nc = C, Cin = K, wt = W^T [C][R][S][K])
struct WinoProblem {
  const float* in; const float* filt; float* out; const float* bias; const float* add;
  int N, H, W, Cin, Nc, relu, flip;
  float* bn_part; const float* bn_pivot;       // optional BatchNorm statistics of the output (see nnl_conv2d_fwd): one partial per 64-pair tile row
  const float* u_pre;                          // optional: the transformed filter U [Nc][4][3][Cin] prepared by nnl_wino_filter_multi (skips the per-call transform)
};

bool nnl_wino_ok(int N, int H, int W, int Cin, int Nc, int R, int S, int stride, int pad);
double nnl_wino_plan_time_us(int N, int H, int W, int Cin, int Nc);    // predicted launch time, us (the planner's cost model)
size_t nnl_wino_workspace_bytes(int N, int H, int W, int Cin, int Nc);
// tile rows (= BatchNorm partials) the launch writes when bn_part is given: ceil(N*H*ceil(W/2) / 64)
int nnl_wino_bn_rows(int N, int H, int W);
int nnl_wino_launch(const WinoProblem& q, void* ws, size_t ws_bytes, int* tile_counters, long n_counters, hipStream_t s);

// the 2-D F(2x2, 3x3) variant (wino2.hip): rows are 2x2 output quads, U is [Nc][16][Cin]; same problem struct
bool nnl_wino2_ok(int N, int H, int W, int Cin, int Nc, int R, int S, int stride, int pad);
double nnl_wino2_plan_time_us(int N, int H, int W, int Cin, int Nc);
bool nnl_wino2_plan_is_pos(int N, int H, int W, int Cin, int Nc);      // the plan is the position-split (small-grid) instantiation
size_t nnl_wino2_workspace_bytes(int N, int H, int W, int Cin, int Nc);
int nnl_wino2_bn_rows(int N, int H, int W);
int nnl_wino2_launch(const WinoProblem& q, void* ws, size_t ws_bytes, int* tile_counters, long n_counters, hipStream_t s);
