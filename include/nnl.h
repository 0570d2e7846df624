/*
 * nnl.h — C ABI of libnnl_hip.so: the MI355X (gfx950) kernels behind the Learner.fit() hot path.
 *
 * The reference (NickTravers/NeuralNetworkLibrary) has NO FFI / plugin registry: every FLOP is an
 * eager torch op called from its nn.Modules.  The drop-in boundary is therefore
 * the set of torch call sites on the hot path; each entry point below names the reference call site
 * (file:line, relative to the reference root) whose arithmetic it replaces.  The Python host side
 * (neuralnetworklibrary_amd/ops.py) binds these with ctypes and calls them from
 * torch.autograd.Function.forward/backward — see INTEGRATION.md for the reference-side stub.
 *
 * Conventions
 *  - plain pointers + sizes, no torch types.  All pointers are DEVICE pointers unless stated.
 *  - every call is asynchronous on `stream` (a hipStream_t passed as void*); no implicit device sync.
 *  - the library never allocates/frees device memory and keeps no pointer past return; workspaces are
 *    caller-owned, sized by the matching *_workspace_bytes().
 *  - return 0 on success, negative nnl_status_t on failure; nnl_last_error() gives the text
 *    (thread-local).  No C++ exception crosses the ABI.
 *  - fp32 everywhere (the reference is fp32, README.md:19-24); indices are int64 (torch LongTensor).
 *  - activations are NHWC ("channels_last" physical layout of a logical NCHW torch tensor);
 *    conv filters are KRSC (= channels_last physical layout of a logical [K,C,R,S] parameter).
 */
#ifndef NNL_H_
#define NNL_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  NNL_OK = 0,
  NNL_ERR_INVALID_ARG = -1,
  NNL_ERR_HIP = -2,
  NNL_ERR_UNSUPPORTED = -3,
  NNL_ERR_WORKSPACE = -4
} nnl_status_t;

int nnl_version(void);
const char* nnl_last_error(void);
/* The NNL_* tuning / A-B environment switches are read once per call site; this makes every site read its variable again
 * (tests and the A/B tools change a switch inside one process). */
int nnl_reload_env(void);

/* ---- profiling hooks (bench.py roofline leg): HIP events recorded on the launch stream --------- */
enum { NNL_PROF_CONV_FWD = 0, NNL_PROF_CONV_DGRAD = 1, NNL_PROF_CONV_WGRAD = 2, NNL_PROF_EMBDOT = 3,
       NNL_PROF_TABULAR = 4, NNL_PROF_RETINA_LOSS = 5, NNL_PROF_LSTM = 6, NNL_PROF_SOFTMAX_CE = 7,
       NNL_PROF_ELEMENTWISE = 8, NNL_PROF_GEMM = 9, NNL_PROF_OPTIM = 10, NNL_PROF_KINDS = 11 };
/* enable!=0 starts recording (bounded event pool); 0 stops. */
int nnl_prof_enable(int enable);
/* Synchronises the recorded events and returns, per kind: launches, total ms, total algorithmic work
 * (flops or bytes as documented per op).  Arrays have NNL_PROF_KINDS entries. Resets the pool. */
int nnl_prof_collect(int64_t* launches, double* total_ms, double* total_work);
/* The same plus, per kind, the EXECUTED work: algorithmic work x (multiplies issued / algorithmic multiplies) of each launch — 1 for the direct
 * kernels, 1 / 1.5 for the 1-D Winograd kernels (forward, dgrad, Winograd-domain wgrad), 1 / 2.25 for the 2-D ones.  total_exec may be NULL. */
int nnl_prof_collect2(int64_t* launches, double* total_ms, double* total_work, double* total_exec);
/* Data-parallel overlap under hipGraph replay (host side: neuralnetworklibrary_amd/dist.py, GradSync.reduce_overlapped; replaces nothing in the
 * single-GPU reference — SURVEY.md 8e).  A captured training step contains nnl_dp_bump(step) at its start and nnl_dp_signal(flag + k, step)
 * right after the last gradient of all-reduce bucket k; after each replay the host enqueues, on a side stream, nnl_dp_wait(flag + k, n, ...)
 * (n = number of replays so far) followed by bucket k's RCCL all-reduce, which therefore starts while the rest of the replayed backward
 * still runs.  The wait kernel is one lane polling with s_sleep, bounded in WALL TIME by timeout_us (the device's constant-rate counter; then
 * *err = 1 and it returns: the stream always drains; the host reads *err with the step's loss and raises). */
int nnl_dp_bump(int32_t* step, void* stream);
int nnl_dp_signal(int32_t* flag, const int32_t* step, void* stream);
int nnl_dp_wait(const int32_t* flag, int32_t value, int64_t timeout_us, int32_t* err, void* stream);
/* First 16 hex digits of the sha256 over the library's sources (csrc/ *.hip, *.h, Makefile, include/nnl.h; csrc/Makefile) it was built from. */
const char* nnl_source_stamp(void);

/* ---- K4: EmbeddingDotBias (CollabFilterNet.forward, Applications/CollabFiltering.py:196-204) ----
 * y_b = lo + (hi-lo)*sigmoid( sum_d U[x[b,0],d]*M[x[b,1],d] + bu[x[b,0]] + bi[x[b,1]] )   (has_range!=0)
 * y_b =                         sum_d ...                + bu + bi                       (has_range==0)
 * x: int64 [n,2]; U [n_user,D]; M [n_item,D]; bu [n_user]; bi [n_item]; y [n]; z [n] (pre-activation,
 * saved for backward; may be NULL).  Out-of-range indices are reported through *err_flag (device int32,
 * may be NULL; set to 1) and the sample is skipped — torch raises IndexError at the same place. */
int nnl_embdotbias_fwd(const int64_t* x, const float* U, const float* M, const float* bu, const float* bi,
                       float* y, float* z, int64_t n, int64_t n_user, int64_t n_item, int64_t D,
                       int has_range, float lo, float hi, int32_t* err_flag, void* stream);
/* Backward of the above = the four dense embedding_backward scatter-adds + sigmoid/mul backward
 * (autograd of CollabFiltering.py:198-203).  dU/dM/dbu/dbi are DENSE tables (nn.Embedding sparse=False,
 * General/Layers.py:59) which this call first zero-fills, then scatter-adds into. */
/* Deterministic by default: with a workspace of nnl_*_bwd_workspace_bytes the samples that hit one table row are added in
 * SAMPLE ORDER (rank sort + segment sum, csrc/scatter_det.h) — torch's CPU embedding_dense_backward order, bitwise reproducible.
 * workspace == NULL (or NNL_SCATTER_ATOMIC=1, or more than 32768 samples): fp32 atomicAdd in arrival order. */
size_t nnl_embdotbias_bwd_workspace_bytes(int64_t n);
int nnl_embdotbias_bwd(const int64_t* x, const float* U, const float* M, const float* z, const float* dy,
                       float* dU, float* dM, float* dbu, float* dbi, int64_t n, int64_t n_user,
                       int64_t n_item, int64_t D, int has_range, float lo, float hi, void* workspace, size_t workspace_bytes, void* stream);

/* ---- K1: conv2d as implicit GEMM on the exact-fp32 MFMA --------------------------------------------
 * Replaces the cuDNN convolutions called by nn.Conv2d inside BasicBlock.forward / Bottleneck.forward
 * (Applications/VisionModels/retinanet.py:43-59, 77-97), the stem (:304), the 1x1 downsample (:344-348),
 * PyramidFeatures (:126-148) and the RetinaNet heads (:187-217, :260-295), and their autograd backward.
 * x [N,H,W,C] NHWC, w [K,R,S,C] KRSC, y [N,P,Q,K] NHWC; P = (H+2*pad-R)/stride+1 (validated). C%4==0
 * (the host pads the 3-channel stem input to 4), K%4==0 for dgrad/wgrad.  fp32 in, fp32 accumulate:
 * results equal a k-ordered fmaf chain (MI355X_MICROARCH.md, Matrix cores). */
typedef struct {
  int32_t N, H, W, C;   /* input  NHWC */
  int32_t K, R, S;      /* filter KRSC */
  int32_t stride, pad;  /* same in both spatial dims (all reference convs are symmetric) */
  int32_t P, Q;         /* output spatial size */
} nnl_conv_geom_t;

/* y = act(conv(x, w) (+ bias[K])); `relu` selects the output activation: 0 none, 1 ReLU, 2 sigmoid (ClassificationModel's
 * `output_act`, reference retinanet.py:286; needs C % 16 == 0).  bias may be NULL.
 * workspace (optional, nnl_conv2d_fwd_workspace_bytes(g); NULL = none): lets the launch use the balanced schedule —
 * when the tile grid is not a multiple of the 256 CUs, the last tiles (or all of them) are cut into k slices whose
 * partial slabs are summed in a fixed order (bitwise reproducible run to run). */
size_t nnl_conv2d_fwd_workspace_bytes(const nnl_conv_geom_t* g);
/* tile_counters (optional): a PERSISTENT device array of nnl_conv2d_tile_counters() int32, zero-initialised once by the
 * caller and never touched otherwise (one per device and stream).  With it the last k slice of a split tile to finish sums
 * the slabs in slice order and writes the output inside the conv kernel (no separate reduce launch); each counter returns
 * to zero before the kernel ends.  NULL: the slabs are reduced by a second launch. */
int64_t nnl_conv2d_tile_counters(void);
/* BatchNorm statistics in the epilogue (optional; all three of bn_partials / bn_pivot / bn_rows or none): when the launch uses
 * the 64x64 tile, the workgroup that produces the final values of a tile also writes, per tile row t = m/64 and channel c,
 * bn_partials[(t*K + c)*2 + {0,1}] = sum over the tile's rows of (y - bn_pivot[c]) and of its square (bn_pivot: any [K] device
 * array close to the mean, e.g. running_mean).  *bn_rows (HOST int, set before return) = number of tile rows written, or 0
 * when this launch could not produce them (other tile shape, first-generation kernel): pass both to nnl_bn_fwd, which then
 * skips its own statistics pass over y.  bn_partials needs ceil(N*P*Q/64) * K * 2 floats. */
int nnl_conv2d_fwd(const float* x, const float* w, const float* bias, float* y, const nnl_conv_geom_t* g,
                   int relu, void* workspace, size_t workspace_bytes, int32_t* tile_counters, float* bn_partials,
                   const float* bn_pivot, int32_t* bn_rows, void* stream);
/* y = conv(x, w) + bias + nearest-x2-upsample(small), small = [N, P/2, Q/2, K]: the FPN top-down merge `P5_upsampled + P4_1(C4)`,
 * `P3_1(C3) + P4_upsampled` (reference retinanet.py:126-148: nn.Upsample(scale_factor=2, mode='nearest') + add) inside the
 * lateral convolution's epilogue.  P, Q even; C % 16 == 0. */
int nnl_conv2d_fwd_add_up2(const float* x, const float* w, const float* bias, const float* small, float* y,
                           const nnl_conv_geom_t* g, void* stream);
/* its gradient with respect to `small`: dsmall[n,h,w,c] = sum of the 2x2 block of dy[N,2h,2w,C] (upsample_nearest2d backward). */
int nnl_upsample2_bwd(const float* dy, float* dsmall, int64_t N, int64_t h, int64_t w, int64_t C, void* stream);
/* wt[C,R,S,K] = transpose of w[K,R,S,C] over (K,C): the B operand of dgrad. */
int nnl_conv2d_weight_transpose(const float* w, float* wt, int K, int R, int S, int C, void* stream);
/* The same transpose for MANY filters in one launch (all convolutions of a model, once per backward pass instead of one small
 * launch per layer).  desc: DEVICE array, one entry per filter; tile_tensor: DEVICE int32 [n_tiles], the entry each 32x32 tile
 * belongs to (entry i owns tiles [first_tile, first_tile + R*S*ceil(K/32)*ceil(C/32))). */
typedef struct {
  const float* w;   /* [K,RS,C] */
  float* wt;        /* [C,RS,K] */
  int32_t K, RS, C, first_tile;
} nnl_wt_desc_t;
int nnl_conv2d_weight_transpose_multi(const nnl_wt_desc_t* desc, const int32_t* tile_tensor, int64_t n_tiles,
                                      double total_elems, void* stream);
/* dx[N,H,W,C] = sum_{r,s,k} dy[n,(h+pad-r)/stride,(w+pad-s)/stride,k] * wt[c,r,s,k] (integral taps only). */
size_t nnl_conv2d_dgrad_workspace_bytes(const nnl_conv_geom_t* g);   /* optional workspace, as for the forward */
/* addend (optional, [N,H,W,C]; K % 16 == 0 and stride 1, or a 3x3 / pad 1 filter at stride 2): dx = dgrad + addend — the
 * gradient that reaches the block input through the shortcut of BasicBlock / Bottleneck (identity, or the input gradient
 * of the downsample convolution; retinanet.py:43-59,344-348) is added in the epilogue instead of by a separate autograd
 * accumulation kernel.  A stride-2 dgrad with even H and W runs its four output-parity classes in one launch. */
int nnl_conv2d_dgrad(const float* dy, const float* wt, float* dx, const nnl_conv_geom_t* g, const float* addend,
                     void* workspace, size_t workspace_bytes, int32_t* tile_counters, void* stream);
/* ---- 3x3 / stride 1 / pad 1 on the fused Winograd F(2,3) kernel (csrc/wino.hip) -------------------------------------------
 * nnl_conv2d_fwd and the stride-1 nnl_conv2d_dgrad use it on their own whenever the planner predicts it to be faster than the
 * direct implicit-GEMM kernel (NNL_CONV_WINO: 0 never, 1 by predicted time = default, 2 wherever it applies); results agree
 * with the direct kernel to fp32 rounding (measured 6e-6 .. 1e-5 absolute on O(1) outputs, tools/bench_wino.py).  The three
 * entry points below expose the kernel and the planners directly for the A/B tools and tests:
 *   out[N,H,W,K] = conv3x3_pad1(x[N,H,W,C], filt) (+ bias[K]) (+ add[N,H,W,K]) (ReLU when relu == 1); flip == 0: filt = w[K,3,3,C];
 *   flip == 1: filt = wt[K,3,3,C] read as wt[.,2-r,2-s,.] (the dgrad filter: x = dy, K = C_in of the layer, wt = W^T[C,R,S,K]).
 *   ws: nnl_debug_conv_wino_workspace_bytes bytes; counters: n_counters zeroed int32 (zero again on return) or NULL (plain grid);
 *   bn_part / bn_pivot (both or neither): BatchNorm partial statistics as for nnl_conv2d_fwd, one per 64 output pairs.
 * nnl_debug_conv_plan_times: out[0..3] (FOUR doubles) = predicted launch time (us) of the direct / the 1-D Winograd / the 2-D kernel for that
 *   problem (out[3] = -1: the spatially staged 2-D kernel of round 4 was removed in round 5);
 *   returns 1 when the dispatcher would pick the Winograd kernel, else 0. */
/* Prepared filters.  The transformed filter of a layer depends on its weights only, so a caller that knows all its layers can
 * transform them in ONE launch per step instead of one per convolution call:
 *   nnl_conv2d_wino_preferred(g, dgrad): which kernel nnl_conv2d_fwd (dgrad = 0) / nnl_conv2d_dgrad (dgrad = 1) takes for geometry g (given
 *     the workspace of the size query): 0 direct, 1 the 1-D F(2,3) Winograd kernel, 2 the 2-D F(2x2,3x3) one;
 *   nnl_wino_filter_multi: descriptor d transforms src [rows,3,3,ch] into dst [rows,4,3,ch]; flip = 0 with src = w[K,3,3,C]
 *     (rows = K, ch = C) gives the FORWARD filter, flip = 1 with src = W^T[C,3,3,K] (rows = C, ch = K: nnl_conv2d_weight_transpose)
 *     the DGRAD filter; block_desc[b] = descriptor served by block b, first_block = its first block, ceil(rows*3*ch / 256) blocks each;
 *   nnl_conv2d_fwd_pre / nnl_conv2d_dgrad_pre: nnl_conv2d_fwd / nnl_conv2d_dgrad with `u` = that prepared filter (NULL: exactly the
 *     plain entry points); u is used only when a Winograd kernel is taken and must then hold the layout of THAT kernel (nnl_conv2d_wino_preferred:
 *     1 -> rows * 12 * ch floats, 2 -> rows * 16 * ch floats, two_d = 1). */
typedef struct {
  const float* src;
  float* dst;
  int32_t rows, ch, flip, first_block;
  int32_t two_d, reserved;                   /* two_d = 1: dst [rows,16,ch] for the 2-D F(2x2,3x3) kernel, ceil(rows*ch / 256) blocks */
} nnl_wino_desc_t;
int nnl_conv2d_wino_preferred(const nnl_conv_geom_t* g, int dgrad);
int nnl_wino_filter_multi(const nnl_wino_desc_t* desc, const int32_t* block_desc, int64_t n_blocks, void* stream);
int nnl_conv2d_fwd_pre(const float* x, const float* w, const float* bias, float* y, const nnl_conv_geom_t* g, int relu,
                       void* workspace, size_t workspace_bytes, int32_t* tile_counters, float* bn_partials, const float* bn_pivot,
                       int32_t* bn_rows, const float* u, void* stream);
int nnl_conv2d_dgrad_pre(const float* dy, const float* wt, float* dx, const nnl_conv_geom_t* g, const float* addend,
                         void* workspace, size_t workspace_bytes, int32_t* tile_counters, const float* u, void* stream);
size_t nnl_debug_conv_wino_workspace_bytes(int N, int H, int W, int C, int K);
int nnl_debug_conv_wino_fwd(const float* x, const float* filt, const float* bias, const float* add, float* y, void* ws,
                            size_t ws_bytes, int32_t* counters, long n_counters, float* bn_part, const float* bn_pivot, int N,
                            int H, int W, int C, int K, int relu, int flip, void* stream);
/* the same through the 2-D F(2x2, 3x3) kernel (csrc/wino2.hip) */
size_t nnl_debug_conv_wino2_workspace_bytes(int N, int H, int W, int C, int K);
int nnl_debug_conv_wino2_fwd(const float* x, const float* filt, const float* bias, const float* add, float* y, void* ws,
                            size_t ws_bytes, int32_t* counters, long n_counters, float* bn_part, const float* bn_pivot, int N,
                            int H, int W, int C, int K, int relu, int flip, void* stream);
int nnl_debug_conv_plan_times(int N, int H, int W, int C, int K, double* out);
/* dw[K,R,S,C] = sum_{n,p,q} dy[n,p,q,k] * x[n,p*stride-pad+r,q*stride-pad+s,c]; split-K partial slabs are
 * reduced in a fixed order (bitwise reproducible).  workspace: nnl_conv2d_wgrad_workspace_bytes(g). */
size_t nnl_conv2d_wgrad_workspace_bytes(const nnl_conv_geom_t* g);
int nnl_conv2d_wgrad(const float* x, const float* dy, float* dw, const nnl_conv_geom_t* g, void* workspace,
                     size_t workspace_bytes, void* stream);
/* out[c] = sum_r a[r][c]  (bias gradients of conv / linear layers), two fixed-order stages (reproducible). */
size_t nnl_colsum_workspace_bytes(int64_t rows, int64_t cols);
/* out = src[0] + ... + src[n-1], added in that order; n in 1..8 dense fp32 tensors of numel elements, 16-byte aligned; src is a HOST
 * array of device pointers.  Replaces autograd's chain of accumulation kernels for a parameter that several forward calls share:
 * RetinaNet's head convolutions run on the five pyramid levels (RegressionModel / ClassificationModel.forward, retinanet.py:187-217,
 * 260-295), so each of their 20 weight / bias gradients was four ATen add launches per step (91 per step in all). */
int nnl_sum_tensors(const float* const* src, int n, float* out, int64_t numel, void* stream);
int nnl_colsum(const float* a, float* out, int64_t rows, int64_t cols, void* workspace, size_t workspace_bytes,
               void* stream);
/* ReLU backward gate of a conv + bias + ReLU layer (the RetinaNet head convs, retinanet.py:192-199,262-272) fused with its bias
 * gradient: g[rows,cols] = dy * [y > 0] and, if colsum != NULL, colsum[c] = sum_rows g[:,c] (fixed-order, reproducible) in one
 * pass over dy and y.  Workspace (only with colsum): nnl_colsum_workspace_bytes(rows, cols). */
int nnl_relu_gate_colsum(const float* dy, const float* y, float* g, float* colsum, int64_t rows, int64_t cols, void* workspace,
                         size_t workspace_bytes, void* stream);
/* The same pass for either fused output activation of nnl_conv2d_fwd: act 1 = ReLU gate (as above), act 2 = sigmoid gate
 * g = dy * y * (1 - y) with y the sigmoid's output (torch's sigmoid_backward; retinanet.py:286 under autograd). */
int nnl_act_gate_colsum(const float* dy, const float* y, float* g, float* colsum, int64_t rows, int64_t cols, int act,
                        void* workspace, size_t workspace_bytes, void* stream);

/* ---- K2: BatchNorm fused with the residual add and ReLU that follow it ---------------------------------
 * Replaces nn.BatchNorm2d + `out += residual` + ReLU of BasicBlock/Bottleneck.forward (retinanet.py:47-48,53-57,
 * 81-95), the stem bn1+relu (:305-306,372-373), and nn.BatchNorm1d of Linear / StructuredDataNet
 * (General/Layers.py:35,40; StructuredData.py:1046,1078).  x,y,residual: [rows, C] row-major (NHWC: rows=N*H*W).
 * training!=0: batch statistics (biased variance for the normalisation; running_var gets the unbiased one) and
 *   running = (1-momentum)*running + momentum*batch   (running_* may be NULL: track_running_stats=False);
 * training==0: normalise with running_mean / running_var.
 * y = (x-mean)*invstd*gamma + beta [+ residual] [ReLU].  save_mean/save_invstd [C] are kept for backward.
 * num_batches_tracked (device int64, may be NULL): nn.BatchNorm's step counter, incremented by the training call.
 * ext_partials / ext_rows / ext_pivot (optional, training only): the (sum, sum of squares) partials written by the
 * convolution that produced x (nnl_conv2d_fwd, bn_partials) with the pivot it used — the statistics pass over x is skipped.
 * pivot_out (optional, training; may alias ext_pivot): receives this step's batch mean — the natural pivot of the next step.
 * relu_mask (optional, with relu != 0): ceil(rows*C/32) + 2 words; bit e of the flat [rows, C] element index is set when
 * y > 0.  Pass it to the backward instead of y: the ReLU gate then costs 1 bit instead of 32 per element of HBM traffic. */
size_t nnl_bn_workspace_bytes(int64_t rows, int64_t C);
int nnl_bn_fwd(const float* x, const float* gamma, const float* beta, const float* residual, float* y,
               float* save_mean, float* save_invstd, float* running_mean, float* running_var, int64_t rows,
               int64_t C, float eps, float momentum, int training, int relu, int64_t* num_batches_tracked,
               uint32_t* relu_mask, const float* ext_partials, int64_t ext_rows, const float* ext_pivot, float* pivot_out,
               void* workspace, size_t workspace_bytes, void* stream);
/* g = dy * [y > 0] (if relu; the gate comes from relu_mask when given, else from y);  dbeta = sum g;  dgamma = sum g*xhat;
 * dres = g (if dres != NULL);
 * dx = gamma*invstd*(g - dbeta/n - xhat*dgamma/n) (training) or gamma*invstd*g (eval). dgamma/dbeta may be NULL. */
int nnl_bn_bwd(const float* dy, const float* y, const uint32_t* relu_mask, const float* x, const float* gamma,
               const float* mean, const float* invstd, float* dx, float* dres, float* dgamma, float* dbeta, int64_t rows,
               int64_t C, int training, int relu, void* workspace, size_t workspace_bytes, void* stream);

/* BatchNorm -> ReLU -> MaxPool2d(ks, stride, pad) in one pass: the ResNet stem (reference retinanet.py:372-374, torchvision's
 * bn1 / relu / maxpool).  x [N,H,W,C] is the convolution output; y [N,P,Q,C] the pooled activation, idx [N,P,Q,C] the window
 * position (kh*ks + kw) of each maximum (torch's tie rule); the normalised activation itself is never written.  save_scale /
 * save_shift [C] receive the per-channel affine (the backward recomputes its ReLU gate from them); statistics, running-stat
 * update and num_batches_tracked as nnl_bn_fwd.  C % 4 == 0 and C/4 dividing 256 (nnl_bn_relu_maxpool_supported).
 * Workspace: nnl_bn_workspace_bytes(N*H*W, C). */
int nnl_bn_relu_maxpool_supported(int64_t C);
int nnl_bn_relu_maxpool_fwd(const float* x, const float* gamma, const float* beta, float* y, uint8_t* idx, float* save_mean,
                            float* save_invstd, float* save_scale, float* save_shift, float* running_mean, float* running_var,
                            int64_t N, int64_t H, int64_t W, int64_t C, int64_t P, int64_t Q, int ks, int stride, int pad, float eps,
                            float momentum, int training, int64_t* num_batches_tracked, void* workspace, size_t workspace_bytes,
                            void* stream);
/* dx [N,H,W,C], dgamma, dbeta from the pooled gradient dpool [N,P,Q,C]: every input pixel gathers dpool over the windows whose
 * arg-max it is (no atomics: reproducible), gated by the recomputed ReLU, then the two BatchNorm backward passes.  y (optional: the
 * forward's pooled output) lets the reduction pass run over the pooled outputs alone (xhat at an arg-max = (y - beta) / gamma):
 * it then reads neither x nor the windows. */
int nnl_bn_relu_maxpool_bwd(const float* dpool, const float* y, const uint8_t* idx, const float* x, const float* gamma,
                            const float* beta, const float* mean, const float* invstd, const float* scale, const float* shift,
                            float* dx, float* dgamma, float* dbeta,
                            int64_t N, int64_t H, int64_t W, int64_t C, int64_t P, int64_t Q, int ks, int stride, int pad,
                            int training, void* workspace, size_t workspace_bytes, void* stream);

/* Cross-replica (synchronised) training-mode BatchNorm for data parallelism (global-batch statistics equal
 * to the single-GPU reference's).  Split-phase, the host runs the collective in between (neuralnetworklibrary_amd/ops.py):
 *   fwd:  nnl_bn_sync_stats -> all_gather of `stats` (2C+2 floats per rank) -> nnl_bn_sync_fwd
 *   bwd:  nnl_bn_sync_bwd_reduce -> all_reduce(sum) of `sums` (2C floats) -> nnl_bn_sync_bwd
 * stats = {mean_r[C], M2_r[C] = sum (x-mean_r)^2, rows>>16, rows&0xFFFF}; all_stats = [world][2C+2] in rank order, merged
 * with the pairwise (Chan) update in that order, so all ranks get bit-identical mean / invstd / running statistics.
 * dgamma / dbeta are this rank's LOCAL sums (the gradient all-reduce averages them like every other parameter);
 * dx uses the global sums and the global row count. */
int nnl_bn_sync_stats(const float* x, float* stats, int64_t rows, int64_t C, void* workspace, size_t workspace_bytes,
                      void* stream);
int nnl_bn_sync_fwd(const float* x, const float* all_stats, int world, const float* gamma, const float* beta,
                    const float* residual, float* y, float* save_mean, float* save_invstd, float* running_mean,
                    float* running_var, int64_t rows, int64_t C, float eps, float momentum, int relu,
                    int64_t* num_batches_tracked, uint32_t* relu_mask, void* workspace, size_t workspace_bytes, void* stream);
int nnl_bn_sync_bwd_reduce(const float* dy, const float* y, const uint32_t* relu_mask, const float* x, const float* mean,
                           const float* invstd, float* sums, int64_t rows, int64_t C, int relu, void* workspace,
                           size_t workspace_bytes, void* stream);
int nnl_bn_sync_bwd(const float* dy, const float* y, const uint32_t* relu_mask, const float* x, const float* gamma,
                    const float* mean, const float* invstd, const float* local_sums, const float* global_sums,
                    const float* all_stats,
                    int world, float* dx, float* dres, float* dgamma, float* dbeta, int64_t rows, int64_t C, int relu,
                    void* workspace, size_t workspace_bytes, void* stream);

/* ---- pooling of the vision path (NHWC, C % 4 == 0 for maxpool) ---------------------------------------------------------
 * nn.MaxPool2d(ksize, stride, pad) of the ResNet stem (Applications/VisionModels/retinanet.py:307,374; torchvision
 * resnet.maxpool) with torch's tie rule — `if ((val > max) || isnan(val))` in (kh, kw) scan order.  idx [N,P,Q,C] uint8 =
 * kh*ksize + kw of the winning tap, kept for the backward, which GATHERS (each input pixel sums dy over the windows whose
 * arg-max it is): no atomics, bitwise reproducible. */
int nnl_maxpool2d_fwd(const float* x, float* y, uint8_t* idx, int64_t N, int64_t H, int64_t W, int64_t C, int64_t P,
                      int64_t Q, int ksize, int stride, int pad, void* stream);
int nnl_maxpool2d_bwd(const float* dy, const uint8_t* idx, float* dx, int64_t N, int64_t H, int64_t W, int64_t C,
                      int64_t P, int64_t Q, int ksize, int stride, int pad, void* stream);
/* AdaptiveConcatPool2d (General/Layers.py:78-87): out[n, 0:C] = max over the HW pixels, out[n, C:2C] = their mean;
 * argmax [N,C] int32 = pixel index torch's adaptive_max_pool2d would report (first maximum / last NaN); the backward sends
 * dout[:, :C] to that pixel and dout[:, C:]/HW to every pixel. */
int nnl_concat_pool_fwd(const float* x, float* out, int32_t* argmax, int64_t N, int64_t HW, int64_t C, void* stream);
int nnl_concat_pool_bwd(const float* dout, const int32_t* argmax, float* dx, int64_t N, int64_t HW, int64_t C, void* stream);

/* ---- detection inference: the device half of BBoxPredictor.__call__ / nms (Applications/VisionModels/retinanet.py:523-812)
 * nnl_bbox_decode: per (image, anchor) best class (first maximum) and its score; keep score > thresh; decode the box
 *   (cx + w*(reg0*std0+mean0), ..., w*exp(reg2*std2+mean2), ...), clip to [0,width]x[0,height], drop empty boxes
 *   (retinanet.py:762-800).  anchors [A,4], reg [bs,A,4], clas [bs,A,K]; mean4 / std4 are HOST arrays.  Candidates are
 *   appended per image in arbitrary order: cand_* have capacity A per image, cand_order = anchor index, cand_count [bs].
 * nnl_nms: top_k candidates by (score descending, order ascending), then greedy non-maximum suppression inside a class with
 *   IoU > max_overlap (retinanet.py:581-604; IoU in the reference's fp32 operation order).  cand_order may be NULL (= position).
 *   kept_* [bs][top_k] in descending score order, kept_count [bs].  The remaining list filters of nms() (relative
 *   thresholds, inclusions, cross-class duplicates, max_boxes) act on the few survivors and stay on the host. */
int nnl_bbox_decode(const float* anchors, const float* reg, const float* clas, int64_t bs, int64_t A, int64_t K,
                    const float* mean4, const float* std4, float thresh, float width, float height, float* cand_boxes,
                    int32_t* cand_classes, float* cand_scores, int32_t* cand_order, int32_t* cand_count, void* stream);
size_t nnl_nms_workspace_bytes(int64_t bs, int64_t top_k);
int nnl_nms(const float* cand_boxes, const int32_t* cand_classes, const float* cand_scores, const int32_t* cand_order,
            const int32_t* cand_count, int64_t bs, int64_t cap, int64_t top_k, float max_overlap, float* kept_boxes,
            int32_t* kept_classes, float* kept_scores, int32_t* kept_count, void* workspace, size_t workspace_bytes,
            void* stream);

/* ---- K3: categorical-embedding front end of StructuredDataNet --------------------------------------------
 * Replaces, per categorical column j, EmbeddingDrop.forward (General/Layers.py:74-76: nn.Embedding(max_norm=1.5)
 * in-place renorm + gather + per-sample dropout mask) and the two torch.cat calls of StructuredDataNet.forward
 * (Applications/StructuredData.py:1075-1082).  Descriptor arrays are DEVICE arrays with one entry per column:
 * tables[j] (pointer to W_j [card[j], dim[j]]), card, dim, col_off (first output column of column j);
 * col_table[c] = column owning output column c (c < cat_width = sum dim); row_off[j] = first global row id of
 * table j, row_table[r] = table owning global row r (r < total_rows = sum card); flags: int32[total_rows], zero
 * on entry and on exit. */
/* In-place max_norm renormalisation of every row looked up by xcat [bs, ncat] (each distinct row exactly once):
 * rows with L2 norm > max_norm are scaled by max_norm/(norm+1e-7)  (torch embedding_renorm_). */
int nnl_tab_renorm(const int64_t* xcat, float* const* tables, const int32_t* card, const int32_t* dim,
                   const int32_t* row_off, const int32_t* row_table, int32_t* flags, int64_t bs, int32_t ncat,
                   int32_t total_rows, float max_norm, int32_t* err_flag, void* stream);
/* out [bs, ld_out]: columns [col_off[j], +dim[j]) = tables[j][xcat[b,j]] * row_mask[j,b]  (row_mask [ncat,bs] or
 * NULL = ones); columns [cat_width, +n_cont) = cont[b,:] * cont_mask[b,:] (cont_mask NULL = ones); rest = 0. */
int nnl_tab_gather_fwd(const int64_t* xcat, const float* const* tables, const int32_t* card, const int32_t* dim,
                       const int32_t* col_off, const int32_t* col_table, const float* row_mask, const float* cont,
                       const float* cont_mask, float* out, int64_t bs, int32_t ncat, int32_t cat_width,
                       int32_t n_cont, int32_t ld_out, void* stream);
/* Backward: dense table gradients (nn.Embedding sparse=False) scatter-added into ONE flat zero-filled buffer
 * dtab_flat (table j at element offset grad_off[j]), and dcont [bs, n_cont] = dout[:, cat_width:] * cont_mask. */
size_t nnl_tab_scatter_bwd_workspace_bytes(int64_t bs, int32_t ncat);      /* deterministic scatter: see nnl_embdotbias_bwd */
int nnl_tab_scatter_bwd(const int64_t* xcat, const int32_t* card, const int32_t* dim, const int32_t* col_off,
                        const int32_t* col_table, const int64_t* grad_off, const float* row_mask,
                        const float* cont_mask, const float* dout, float* dtab_flat, int64_t dtab_elems,
                        float* dcont, int64_t bs, int32_t ncat, int32_t cat_width, int32_t n_cont, int32_t ld_out,
                        void* workspace, size_t workspace_bytes, void* stream);
/* The same backward WITHOUT a sort, in one launch (round 4; embedding widths <= 32): block b < n_scan_blocks owns the flat elements
 * blk_first[b] .. +255 of column blk_col[b]'s [card][dim] gradient and scans that column's bs indices in sample order (the
 * summation order of nnl_tab_scatter_bwd's deterministic path); every element of dtab_flat is written (no zero fill needed), the
 * continuous columns' gradient rides along.  dout may have any row stride ld_out >= cat_width + n_cont. */
int nnl_tab_scan_bwd(const int64_t* xcat, const int32_t* card, const int32_t* dim, const int32_t* col_off,
                     const int64_t* grad_off, const float* row_mask, const float* cont_mask, const float* dout,
                     float* dtab_flat, float* dcont, const int32_t* blk_col, const int32_t* blk_first, int32_t n_scan_blocks,
                     int32_t max_dim, int64_t bs, int32_t ncat, int32_t cat_width, int32_t n_cont, int32_t ld_out, void* stream);

/* dst [rows, Cp] = src [rows, C] followed by zero columns (channel padding to the kernels' 4-float granularity, one launch). */
int nnl_pad_cols(const float* src, float* dst, int64_t rows, int64_t C, int64_t Cp, void* stream);
/* Two dropout keep masks from ONE uniform draw u [na + nb]: a[i] = (u[i] < keep_a) / keep_a, b[i] = (u[na + i] < keep_b) / keep_b —
 * the per-sample row masks of EmbeddingDrop and the continuous-input mask of StructuredDataNet (General/Layers.py:75-76,
 * StructuredData.py:1079: nn.Dropout applied to ones / to the inputs: Bernoulli(keep) / keep). */
int nnl_keep_masks(const float* u, float* a, int64_t na, float keep_a, float* b, int64_t nb, float keep_b, void* stream);

/* nn.Linear with 1 - 4 output features (the last layer of FullyConnectedNet, General/Layers.py:146 — the tabular regression head):
 * y [M,N] = x [M,K] (row stride ldx) w[N,K]^T + bias.  Backward: dx [M,K] dense, dw [N,K], db [N] (any may be NULL), fixed-order
 * sums (csrc/linear_small.hip). */
int nnl_linear_small_supported(int64_t N);
int nnl_linear_small_fwd(const float* x, const float* w, const float* bias, float* y, int64_t M, int64_t K, int64_t ldx, int64_t N,
                         void* stream);
size_t nnl_linear_small_bwd_workspace_bytes(int64_t M, int64_t K, int64_t N);
int nnl_linear_small_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, float* db, int64_t M, int64_t K,
                         int64_t ldx, int64_t N, void* workspace, size_t workspace_bytes, void* stream);

/* ---- K6: fused RetinaNet detection loss (anchor matching + focal + smooth-L1) -----------------------------
 * Replaces SSD_loss.__call__ and everything it calls per image (Applications/Vision.py:1620-1644 -> ssd1 :1568-1605,
 * match_anchors_objects :1474-1511, jaccard :234-256, focal_loss_retina :1513-1530, smoothL1_loss_retina :1532-1566).
 * anchors [A,4]; reg [bs,A,4]; clas [bs,A,K] (probabilities); boxes [bs,M,4], cats int64 [bs,M], both padded with -1
 * (Vision.py:798-809), M <= 128.  Outputs: out[3] = {(1-beta)*reg + beta*clas, reg_loss, clas_loss} (batch means),
 * state int32 [bs,A] (>=0 matched object, -1 negative, -2 ignored) and npos float [bs] are kept for backward. */
size_t nnl_retina_loss_workspace_bytes(int64_t bs, int64_t A);
int nnl_retina_loss_fwd(const float* anchors, const float* reg, const float* clas, const float* boxes,
                        const int64_t* cats, int32_t* state, float* npos, float* out, int64_t bs, int64_t A,
                        int64_t K, int64_t M, float beta, float alpha, float gamma, void* workspace,
                        size_t workspace_bytes, void* stream);
/* dreg [bs,A,4], dclas [bs,A,K] = d out[0] / d reg, d clas times *grad_out (device scalar). */
int nnl_retina_loss_bwd(const float* anchors, const float* reg, const float* clas, const float* boxes,
                        const int64_t* cats, const int32_t* state, const float* npos, const float* grad_out,
                        float* dreg, float* dclas, int64_t bs, int64_t A, int64_t K, int64_t M, float beta,
                        float alpha, float gamma, void* stream);

/* ---- K5: recurrence of one weight-dropped LSTM layer --------------------------------------------------------
 * Replaces nn.LSTM(num_layers=1) as called by WeightDropLSTM1.forward (Applications/Text.py:495-513) inside
 * LSTM_Encoder.forward (:543-548): gates_t = gx_t + h_{t-1} W_hh^T, c_t = s(f) c_{t-1} + s(i) tanh(g),
 * h_t = s(o) tanh(c_t), gate order i,f,g,o (torch).  gx [T,B,4H] = x_t W_ih^T + b_ih + b_hh for all t (one big
 * GEMM done by the caller with nnl_conv2d_fwd as a 1x1 conv).  Padded operands (the GEMM k dimension must be a
 * multiple of 32): Hp = nnl_lstm_padded_hidden(H) = ceil32(H), Gp = nnl_lstm_padded_gates(H) = ceil32(4H);
 * w_hh_pad [4H, Hp] = (dropped) W_hh with zero-padded rows; h0, c0 [B,H].  Outputs y [T,B,H] (h_t), cy [T,B,H] (c_t)
 * and gates [T,B,4H] (ACTIVATED i,f,g,o) — the last two are saved for backward.
 * One kernel launch per timestep: the recurrent GEMM runs in k slices over <= 256 workgroups and the workgroup that finishes
 * a tile last (atomic ticket, nobody waits) sums the slices in a fixed order and applies the cell (csrc/lstm.hip).
 * PERSISTENT path (csrc/lstm_persist.hip; B <= 64 and the per-workgroup W_hh slice fits the LDS — the AWD-LSTM shapes): ONE
 * cooperative launch per layer and direction; ceil(H/U) <= 256 workgroups keep their 4U rows of W_hh in LDS for all T steps,
 * exchange h_t through a k-major buffer with one slot per timestep and meet at a grid barrier per step.
 * err_flag: one int32 of device memory (may be NULL: per-timestep path only); a persistent launch whose grid barrier times out
 * sets it to 2 and drains. */
int64_t nnl_lstm_padded_hidden(int64_t H);
int64_t nnl_lstm_padded_gates(int64_t H);
size_t nnl_lstm_workspace_bytes(int64_t T, int64_t B, int64_t H);
int nnl_lstm_fwd(const float* gx, const float* w_hh_pad, const float* h0, const float* c0, float* y, float* cy,
                 float* gates, int64_t T, int64_t B, int64_t H, void* workspace, size_t workspace_bytes, int32_t* err_flag,
                 void* stream);
/* BPTT: dy [T,B,H] (may be NULL), dhT / dcT [B,H] (may be NULL = 0), w_hh_t_pad [H, Gp] = W_hh^T with zero-padded rows.
 * Outputs dgates_pad [T,B,Gp] (columns < 4H: gradient of the PRE-activation gates = d gx; columns >= 4H are not written
 * and must be zero on entry; the caller derives dW_ih, dW_hh, db, dx from it with the GEMM entry points), dh0, dc0 [B,H]. */
int nnl_lstm_bwd(const float* dy, const float* dhT, const float* dcT, const float* gates, const float* cy,
                 const float* c0, const float* w_hh_t_pad, float* dgates_pad, float* dh0, float* dc0, int64_t T, int64_t B,
                 int64_t H, void* workspace, size_t workspace_bytes, int32_t* err_flag, void* stream);
/* The partition (KG x NG workgroups, k-slice Ks, column slice Ns, NT column tiles -> out5) the 2-D persistent BPTT kernel
 * (csrc/lstm_bptt2.hip) would use for this shape; returns 0 when the shape does not fit it.  Host-only. */
int nnl_debug_lstm_bptt2_plan(int64_t B, int64_t H, int32_t* out5);

/* nn.MSELoss(reduction='mean') — `loss_func_dict['cont']` (General/Learner.py:20), the loss of the collaborative-filtering and
 * structured-data heads: *loss = mean((pred - target)^2) over n fp32 elements (one launch up to 65 536 samples, fixed-order sum);
 * backward: dpred = *grad_out (device scalar, NULL = 1) * 2 (pred - target) / n.  Workspace only above 65 536 samples. */
size_t nnl_mse_workspace_bytes(int64_t n);
int nnl_mse_fwd(const float* pred, const float* target, float* loss, int64_t n, void* workspace, size_t workspace_bytes, void* stream);
int nnl_mse_bwd(const float* pred, const float* target, const float* grad_out, float* dpred, int64_t n, void* stream);
/* FullyConnectedNet's 'sigmoidal' output activation (General/Layers.py:150-152): y = lo + (hi - lo) * sig, sig = sigmoid(x) (both
 * written); backward: dx = dy * (hi - lo) * sig * (1 - sig). */
int nnl_scaled_sigmoid_fwd(const float* x, float* y, float* sig, int64_t n, float lo, float hi, void* stream);
int nnl_scaled_sigmoid_bwd(const float* dy, const float* sig, float* dx, int64_t n, float lo, float hi, void* stream);

/* ---- K5b: embedding with per-vocabulary-row dropout mask, fused softmax + cross-entropy -----------------------
 * nnl_embedding_rowmask_*: EmbeddingDropout.forward, F.embedding(x, W * mask[V,1], pad) (Text.py:465-475):
 * out[i,:] = W[x[i],:] * rowmask[x[i]] (rowmask NULL = ones); backward zero-fills dW [V,D] and scatter-adds
 * dout[i,:]*rowmask[x[i]] except for x[i] == padding_idx (pass -1 for none). */
int nnl_embedding_rowmask_fwd(const int64_t* x, const float* W, const float* rowmask, float* out, int64_t n,
                              int64_t V, int64_t D, int32_t* err_flag, void* stream);
size_t nnl_embedding_rowmask_bwd_workspace_bytes(int64_t n);               /* deterministic scatter: see nnl_embdotbias_bwd */
int nnl_embedding_rowmask_bwd(const int64_t* x, const float* rowmask, const float* dout, float* dW, int64_t n,
                              int64_t V, int64_t D, int64_t padding_idx, void* workspace, size_t workspace_bytes, void* stream);
/* F.cross_entropy(logits [rows,V], target [rows], reduction='mean') (Text.py:773; also nn.CrossEntropyLoss of
 * General/Learner.py:20): lse[r] = logsumexp, loss_rows[r] = lse[r] - logits[r,target[r]], *loss_mean = mean.
 * Backward: dlogits = (softmax - onehot) * (*grad_out) / rows. */
int nnl_softmax_ce_fwd(const float* logits, const int64_t* target, float* lse, float* loss_rows, float* loss_mean,
                       int64_t rows, int64_t V, int32_t* err_flag, void* stream);
/* ld_dlogits >= V: row stride of dlogits; columns [V, ld_dlogits) are written as zeros (a gradient buffer whose rows are already
 * padded to the GEMM granularity saves the consumer an 848 MB pad copy at V = 47 343). */
int nnl_softmax_ce_bwd(const float* logits, const int64_t* target, const float* lse, const float* grad_out,
                       float* dlogits, int64_t rows, int64_t V, int64_t ld_dlogits, void* stream);
/* AR / TAR regularisers of RegSeqCrossEntropyLoss (Text.py:765-777) on h = enc_out [T, R] (R = bs * emb_dim, contiguous):
 * out3[0] = alpha * mean(h^2) + beta * mean((h[1:] - h[:-1])^2), out3[1] = mean(h^2), out3[2] = the second mean; fixed-order
 * two-stage sums (bitwise reproducible).  Backward: dh = *grad_out (device scalar, NULL = 1) * d out3[0] / dh. */
size_t nnl_seq_reg_workspace_bytes(int64_t T, int64_t R);
int nnl_seq_reg_fwd(const float* h, float* out3, int64_t T, int64_t R, float alpha, float beta, void* workspace,
                    size_t workspace_bytes, void* stream);
int nnl_seq_reg_bwd(const float* h, const float* grad_out, float* dh, int64_t T, int64_t R, float alpha, float beta, void* stream);
/* WeightDropLSTM1's weight drop (Text.py:495-513: W = Dropout_p(W_hh_raw), one mask per forward call) fused with the zero
 * padding / un-padding of the recurrent matrix: out[r, c] = src[r*ld_src + c] * m(r, c) for c < H and 0 for H <= c < ld_out.
 * m = mask[r*H + c] when mask != NULL (an explicit, already scaled mask: parity tests, keyed dropout); otherwise
 * m = [u(seed, r*H + c) >= p] / (1 - p) from a counter-based hash (p = 0: m = 1, a padded copy).  The same call with
 * (src = dW, ld_src = its row stride, ld_out = H) is the backward: dW_raw = dW * m — no mask tensor is stored. */
int nnl_weight_drop(const float* src, int64_t ld_src, const float* mask, float* out, int64_t ld_out, int64_t rows, int64_t H,
                    uint64_t seed, float p, void* stream);

/* ---- K8: fused multi-tensor Optimizer.step ------------------------------------------------------------------------
 * Replaces Optimizer.step (General/Optimizer.py:58-70): decoupled weight decay X *= 1 - wd_g*lr_g (:60-67), global-norm
 * clip (:54-56, torch.nn.utils.clip_grad_norm_) and the torch.optim SGD(momentum) / Adam update (General/Learner.py:17-19)
 * for all parameter tensors at once.  `tensors` is a DEVICE array of n_tensors descriptors; param / grad / state arrays of
 * one tensor share one dense layout and are indexed flat.  grad == NULL: decay only.  lr and decay (= 1 - wd*lr, or 1)
 * are per tensor (layer-group learning rates).  (chunk_tensor[c], chunk_off[c]) maps workgroup c to a piece of
 * nnl_optim_chunk_elems() elements.  kind 0 = SGD (state1 = momentum buffer, zero before the first step; momentum may be
 * 0), kind 1 = Adam (state1 = exp_avg, state2 = exp_avg_sq).  `hyper` is a DEVICE array of 8 floats {momentum (SGD) or
 * 1-beta1 (Adam), beta1, beta2, eps, bc1 = 1-beta1^t, sqrt(bc2) = sqrt(1-beta2^t), clip max_norm, 1-beta2} (1-beta rounded once from
 * double by the caller, as torch.optim.Adam's scalar arguments are): hyper-parameters are read from
 * memory so that a captured hipGraph of the whole step can be replayed with new values.  use_clip != 0: gradients are
 * scaled by min(1, max_norm/(||g||_2 + 1e-6)) (also written back to .grad, as clip_grad_norm_ does); clip_workspace:
 * n_chunks + 2 floats, [0] = coefficient, [1] = total norm on return. */
typedef struct {
  float* param;
  float* grad;
  float* state1;
  float* state2;
  int64_t numel;
  float lr;
  float decay;
} nnl_optim_tensor_t;
int64_t nnl_optim_chunk_elems(void);
/* tensors[i].lr / .decay = dyn[2i] / dyn[2i+1] (i < n), hyper[0..8) = dyn[2n ..): the per-step values of a REPLAYED step, patched into the
 * table after its captured upload (see csrc/optim.hip). */
int nnl_optim_patch(nnl_optim_tensor_t* tensors, float* hyper, const float* dyn, int64_t n, void* stream);
int nnl_optim_step(const nnl_optim_tensor_t* tensors, const int32_t* chunk_tensor, const int64_t* chunk_off,
                   int64_t n_chunks, int kind, const float* hyper, int use_clip, float* clip_workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NNL_H_ */
