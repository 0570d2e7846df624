// K3 — the categorical-embedding front end of StructuredDataNet (Applications/StructuredData.py:1072-1084):
// per column j: nn.Embedding(max_norm) in-place renorm of the looked-up rows, gather, per-SAMPLE dropout mask
// (EmbeddingDrop.forward, General/Layers.py:74-76), torch.cat of all columns, and the concat with the (already
// batch-normed, dropped-out) continuous block — one gather kernel instead of ~5 launches per column.
//
// HBM/gather-bound and tiny: per sample 8*ncat B of indices + 4*sum(d_j) B gathered + 4*D_out B written.
// Descriptors live in device memory: tables[j] (pointer), card[j], dim[j], col_off[j] (first output column).
#include "scatter_det.h"

typedef int i32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kBlock = 256;

// flags[row_off[j] + idx] = 1 for every looked-up row (idempotent scatter)
__global__ void tab_mark_kernel(const int64_t* __restrict__ xcat, const int32_t* __restrict__ card,
                                const int32_t* __restrict__ row_off, int32_t* __restrict__ flags, long bs, int ncat,
                                int32_t* __restrict__ err_flag) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= bs * ncat) return;
  const int j = (int)(i % ncat);
  const int64_t idx = xcat[i];
  if (idx < 0 || idx >= card[j]) { if (err_flag) *err_flag = 1; return; }
  flags[row_off[j] + idx] = 1;
}

// one 16-lane group per table row: if flagged, renormalise to L2 norm <= max_norm exactly like torch's
// embedding_renorm_ (scale = max_norm / (norm + 1e-7) when norm > max_norm), then clear the flag
__global__ void tab_renorm_kernel(float* const* __restrict__ tables, const int32_t* __restrict__ dim,
                                  const int32_t* __restrict__ row_off, const int32_t* __restrict__ row_table,
                                  int32_t* __restrict__ flags, int total_rows, float max_norm) {
  const int grp = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int sub = threadIdx.x & 15;
  if (grp >= total_rows) return;
  if (flags[grp] == 0) return;                       // uniform inside the 16-lane group
  const int j = row_table[grp];
  const int d = dim[j];
  float* row = tables[j] + (long)(grp - row_off[j]) * d;
  float ss = 0.f;
  for (int e = sub; e < d; e += 16) { const float v = row[e]; ss += v * v; }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
  const float norm = sqrtf(ss);
  if (norm > max_norm) {
    const float scale = max_norm / (norm + 1e-7f);
    for (int e = sub; e < d; e += 16) row[e] *= scale;
  }
  if (sub == 0) flags[grp] = 0;
}

// out[b, col_off[j] + e] = tables[j][xcat[b,j], e] * row_mask[j, b];  out[b, cat_width + e] = cont[b,e]*cont_mask[b,e];
// columns >= cat_width + n_cont (padding up to ld_out) are zero-filled.
__global__ void tab_gather_kernel(const int64_t* __restrict__ xcat, const float* const* __restrict__ tables,
                                  const int32_t* __restrict__ card, const int32_t* __restrict__ dim,
                                  const int32_t* __restrict__ col_off, const int32_t* __restrict__ col_table,
                                  const float* __restrict__ row_mask, const float* __restrict__ cont,
                                  const float* __restrict__ cont_mask, float* __restrict__ out, long bs, int ncat,
                                  int cat_width, int n_cont, int ld_out) {
  // thread <-> one output element; consecutive threads walk one sample's row (coalesced 4-B stores)
  const long total = bs * ld_out;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / ld_out;
    const int col = (int)(i - b * ld_out);
    float v = 0.f;
    if (col < cat_width) {
      const int j = col_table[col];
      const int64_t idx = xcat[b * ncat + j];
      if (idx >= 0 && idx < card[j]) {
        v = tables[j][idx * dim[j] + (col - col_off[j])];
        if (row_mask) v *= row_mask[(long)j * bs + b];
      }
    } else if (col < cat_width + n_cont) {
      const int e = col - cat_width;
      v = cont[b * n_cont + e];
      if (cont_mask) v *= cont_mask[b * n_cont + e];
    }
    out[i] = v;
  }
}

// dtab_flat[grad_off[j] + idx*d + e] += dout[b, col_off[j]+e] * row_mask[j,b];  dcont[b,e] = dout[b,cat_width+e]*cont_mask
__global__ void tab_scatter_kernel(const int64_t* __restrict__ xcat, const int32_t* __restrict__ card,
                                   const int32_t* __restrict__ dim, const int32_t* __restrict__ col_off,
                                   const int32_t* __restrict__ col_table, const int64_t* __restrict__ grad_off,
                                   const float* __restrict__ row_mask, const float* __restrict__ cont_mask,
                                   const float* __restrict__ dout, float* __restrict__ dtab_flat, float* __restrict__ dcont,
                                   long bs, int ncat, int cat_width, int n_cont, int ld_out) {
  const int width = cat_width + n_cont;
  const long total = bs * width;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / width;
    const int col = (int)(i - b * width);
    const float g = dout[b * ld_out + col];
    if (col < cat_width) {
      const int j = col_table[col];
      const int64_t idx = xcat[b * ncat + j];
      if (idx >= 0 && idx < card[j]) {
        const float m = row_mask ? row_mask[(long)j * bs + b] : 1.f;
        if (m != 0.f) atomicAdd(dtab_flat + grad_off[j] + idx * dim[j] + (col - col_off[j]), g * m);
      }
    } else if (dcont) {
      const int e = col - cat_width;
      dcont[b * n_cont + e] = cont_mask ? g * cont_mask[b * n_cont + e] : g;
    }
  }
}

__global__ void tab_dcont_kernel(const float* __restrict__ dout, const float* __restrict__ cont_mask, float* __restrict__ dcont,
                                 long bs, int cat_width, int n_cont, int ld_out) {
  const long total = bs * n_cont;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / n_cont;
    const int e = (int)(i - b * n_cont);
    const float g = dout[b * ld_out + cat_width + e];
    dcont[i] = cont_mask ? g * cont_mask[i] : g;
  }
}


// ---- backward without a sort (round 4): one thread per (table row r, component d) of one column scans the column's indices in
// SAMPLE ORDER and adds the matching rows of dout (times the sample's row mask) — the same summation order as the rank-sort +
// segment-sum pair (bitwise reproducible, no atomics), but ONE launch instead of memset + sort + segment sum + dcont (85 -> ~8 us
// at the Rossmann shape: 1024 samples x 32 columns, 1.5 k table rows).  The indices and the masked gradient slice of 256 samples
// at a time are staged in LDS, so the scan is LDS-only: (bs / 256) x 256 x (broadcast read, read, compare, select-add).
// Blocks [0, n_scan): (column blk_col[b], flat elements blk_first[b] .. +255 of that column's [card][dim] gradient) — every
// element is written, zeros included; blocks >= n_scan: the continuous columns' gradient (dout * cont_mask).
constexpr int kScanMaxDim = 32;

__global__ __launch_bounds__(256) void tab_scan_bwd_kernel(const int64_t* __restrict__ xcat, const int32_t* __restrict__ card,
                                                           const int32_t* __restrict__ dim, const int32_t* __restrict__ col_off,
                                                           const int64_t* __restrict__ grad_off, const float* __restrict__ row_mask,
                                                           const float* __restrict__ cont_mask, const float* __restrict__ dout,
                                                           float* __restrict__ dtab, float* __restrict__ dcont,
                                                           const int32_t* __restrict__ blk_col, const int32_t* __restrict__ blk_first,
                                                           int n_scan, long bs, int ncat, int cat_width, int n_cont, int ld_out) {
  __shared__ __attribute__((aligned(16))) int s_idx[256];
  __shared__ float s_val[256 * kScanMaxDim];
  __shared__ float s_mask[256];
  const int t = threadIdx.x;
  if ((int)blockIdx.x >= n_scan) {
    const long i = ((long)blockIdx.x - n_scan) * 256 + t;
    if (dcont && i < bs * n_cont) {
      const long b = i / n_cont;
      const int e = (int)(i - b * n_cont);
      const float g = dout[b * ld_out + cat_width + e];
      dcont[i] = cont_mask ? g * cont_mask[i] : g;
    }
    return;
  }
  const int j = blk_col[blockIdx.x];
  const int dj = dim[j], coff = col_off[j];
  const int f = blk_first[blockIdx.x] + t;
  const bool mine = f < card[j] * dj;
  const int r = mine ? f / dj : -2, d = mine ? f - r * dj : 0;
  float acc = 0.f;
  const unsigned magic = dj > 1 ? (unsigned)((0x100000000ull + dj - 1) / dj) : 0u;       // e / dj for e < 2^13 (exact: e * dj < 2^32 / dj ... checked by the test shapes)
  for (long s0 = 0; s0 < bs; s0 += 256) {
    const long sb = s0 + t;
    s_idx[t] = sb < bs ? (int)xcat[sb * ncat + j] : -1;
    const float mk = (sb < bs && row_mask) ? row_mask[(long)j * bs + sb] : 1.f;     // this thread's SAMPLE (sl = t): its mask, then its
    // share of the 256 x dj gradient slice — all dj (<= 32) loads of a thread in flight at once (batches of 8 cost a ~1.5 us round
    // trip each: 24 of the first version's 38 us); element e = t + 256 u lives at sample e / dj (multiply-high by ceil(2^32 / dj))
    s_mask[t] = mk;
    float v[kScanMaxDim];
#pragma unroll
    for (int u = 0; u < kScanMaxDim; ++u) {
      const int e = t + u * 256;
      const int sl = dj == 1 ? e : (int)__umulhi((unsigned)e, magic);
      const int dd = e - sl * dj;
      const long ss = s0 + sl;
      v[u] = (u < dj && ss < bs) ? dout[ss * ld_out + coff + dd] : 0.f;
    }
    __syncthreads();                                        // (s_mask complete)
#pragma unroll
    for (int u = 0; u < kScanMaxDim; ++u) {
      if (u < dj) {
        const int e = t + u * 256;
        const int sl = dj == 1 ? e : (int)__umulhi((unsigned)e, magic);
        s_val[e] = row_mask ? v[u] * s_mask[sl] : v[u];
      }
    }
    __syncthreads();
    // eight samples per batch: all LDS reads first, then compare / select / add IN SAMPLE ORDER (written as a conditional add
    // the compiler emitted read -> wait -> branch -> read -> wait per sample: two serialized LDS round trips, 80 us per launch)
    const float* vp = s_val + d;
    for (int sl0 = 0; sl0 < 256; sl0 += 8) {
      const i32x4 ia = *reinterpret_cast<const i32x4*>(s_idx + sl0), ib = *reinterpret_cast<const i32x4*>(s_idx + sl0 + 4);
      float vv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) vv[u] = vp[(sl0 + u) * dj];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int iu = u < 4 ? ia[u & 3] : ib[u & 3];
        acc += iu == r ? vv[u] : 0.f;
      }
    }
    __syncthreads();
  }
  if (mine) dtab[grad_off[j] + f] = acc;
}

// dst [rows][Cp] = src [rows][C] with zero pad columns (ops._pad_c4: channel counts that are no multiple of 4 — the tabular input
// width 203, 3-channel images): one launch instead of a fill + a strided copy
__global__ void pad_cols_kernel(const float* __restrict__ src, float* __restrict__ dst, long rows, int C, int Cp) {
  const long total = rows * Cp;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / Cp;
    const int c = (int)(i - r * Cp);
    dst[i] = c < C ? src[r * C + c] : 0.f;
  }
}

// two dropout keep masks from one uniform draw: out[i] = (u[i] < keep) / keep  (the reference draws them with
// nn.Dropout(ones) / bernoulli_: the same distribution; Layers.py:75-76, StructuredData.py:1079)
__global__ void keep_masks_kernel(const float* __restrict__ u, float* __restrict__ a, long na, float keep_a, float* __restrict__ b,
                                  long nb, float keep_b) {
  const long total = na + nb;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    if (i < na) a[i] = u[i] < keep_a ? 1.f / keep_a : 0.f;
    else b[i - na] = u[i] < keep_b ? 1.f / keep_b : 0.f;
  }
}

int grid_for(long n) {
  long b = nnl_cdiv(n, kBlock);
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" int nnl_tab_renorm(const int64_t* xcat, float* const* tables, const int32_t* card, const int32_t* dim,
                              const int32_t* row_off, const int32_t* row_table, int32_t* flags, int64_t bs, int32_t ncat,
                              int32_t total_rows, float max_norm, int32_t* err_flag, void* stream) {
  NNL_CHECK_ARG(bs >= 0 && ncat > 0 && total_rows > 0, "tab_renorm: bad sizes");
  if (bs == 0) return NNL_OK;
  NNL_CHECK_ARG(xcat && tables && card && dim && row_off && row_table && flags, "tab_renorm: null pointer");
  hipStream_t s = (hipStream_t)stream;
  NnlProfScope prof(NNL_PROF_TABULAR, s, 8.0 * bs * ncat);
  hipLaunchKernelGGL(tab_mark_kernel, dim3(grid_for(bs * ncat)), dim3(kBlock), 0, s, xcat, card, row_off, flags, (long)bs, ncat,
                     err_flag);
  NNL_CHECK_LAUNCH();
  hipLaunchKernelGGL(tab_renorm_kernel, dim3((unsigned)nnl_cdiv((long)total_rows * 16, kBlock)), dim3(kBlock), 0, s, tables, dim,
                     row_off, row_table, flags, total_rows, max_norm);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

extern "C" int nnl_tab_gather_fwd(const int64_t* xcat, const float* const* tables, const int32_t* card, const int32_t* dim,
                                  const int32_t* col_off, const int32_t* col_table, const float* row_mask, const float* cont,
                                  const float* cont_mask, float* out, int64_t bs, int32_t ncat, int32_t cat_width,
                                  int32_t n_cont, int32_t ld_out, void* stream) {
  NNL_CHECK_ARG(bs >= 0 && ncat >= 0 && cat_width >= 0 && n_cont >= 0 && ld_out >= cat_width + n_cont && ld_out > 0,
                "tab_gather_fwd: bad sizes");
  if (bs == 0) return NNL_OK;
  NNL_CHECK_ARG(out && (ncat == 0 || (xcat && tables && card && dim && col_off && col_table)) && (n_cont == 0 || cont),
                "tab_gather_fwd: null pointer");
  hipStream_t s = (hipStream_t)stream;
  NnlProfScope prof(NNL_PROF_TABULAR, s, (double)bs * (8.0 * ncat + 4.0 * (cat_width + n_cont) + 4.0 * ld_out));
  hipLaunchKernelGGL(tab_gather_kernel, dim3(grid_for(bs * ld_out)), dim3(kBlock), 0, s, xcat, tables, card, dim, col_off,
                     col_table, row_mask, cont, cont_mask, out, (long)bs, ncat, cat_width, n_cont, ld_out);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

extern "C" size_t nnl_tab_scatter_bwd_workspace_bytes(int64_t bs, int32_t ncat) {
  return (bs > 0 && ncat > 0) ? nnl_det::order_bytes(bs, ncat) : 0;
}

extern "C" int nnl_tab_scatter_bwd(const int64_t* xcat, const int32_t* card, const int32_t* dim, const int32_t* col_off,
                                   const int32_t* col_table, const int64_t* grad_off, const float* row_mask,
                                   const float* cont_mask, const float* dout, float* dtab_flat, int64_t dtab_elems,
                                   float* dcont, int64_t bs, int32_t ncat, int32_t cat_width, int32_t n_cont, int32_t ld_out,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  NNL_CHECK_ARG(bs >= 0 && ncat >= 0 && cat_width >= 0 && n_cont >= 0 && ld_out >= cat_width + n_cont && dtab_elems >= 0,
                "tab_scatter_bwd: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  if (dtab_flat && dtab_elems > 0) NNL_CHECK_HIP(hipMemsetAsync(dtab_flat, 0, sizeof(float) * dtab_elems, s));
  if (bs == 0) return NNL_OK;
  NNL_CHECK_ARG(dout && (ncat == 0 || (xcat && card && dim && col_off && col_table && grad_off && dtab_flat)),
                "tab_scatter_bwd: null pointer");
  NnlProfScope prof(NNL_PROF_TABULAR, s, (double)bs * (8.0 * ncat + 8.0 * (cat_width + n_cont)));
  if (ncat > 0 && nnl_det::use_det(bs, workspace) && workspace_bytes >= nnl_tab_scatter_bwd_workspace_bytes(bs, ncat)) {
    // deterministic: per column, the samples of a table row are added in sample order (scatter_det.h); one sort launch and one
    // segment-sum launch cover all ncat columns
    int* order = (int*)workspace;
    int st = nnl_det::sort_rows(xcat, ncat, bs, ncat, order, s);
    if (st) return st;
    nnl_det::SegSumParams q{};
    q.idx = xcat; q.idx_stride = ncat; q.order = order; q.n = (int)bs;
    q.card_arr = card; q.dim_arr = dim; q.coff_arr = col_off; q.dst_off_arr = grad_off; q.dst = dtab_flat;
    q.src = dout; q.ld = ld_out; q.scale_i = row_mask; q.scale_i_stride = bs; q.skip_row = -1;
    if ((st = nnl_det::segsum(q, ncat, s))) return st;
    if (dcont && n_cont > 0) {
      hipLaunchKernelGGL(tab_dcont_kernel, dim3(grid_for(bs * n_cont)), dim3(kBlock), 0, s, dout, cont_mask, dcont, (long)bs, cat_width,
                         n_cont, ld_out);
      NNL_CHECK_LAUNCH();
    }
    return NNL_OK;
  }
  hipLaunchKernelGGL(tab_scatter_kernel, dim3(grid_for(bs * (cat_width + n_cont))), dim3(kBlock), 0, s, xcat, card, dim, col_off,
                     col_table, grad_off, row_mask, cont_mask, dout, dtab_flat, dcont, (long)bs, ncat, cat_width, n_cont, ld_out);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

extern "C" int nnl_tab_scan_bwd(const int64_t* xcat, const int32_t* card, const int32_t* dim, const int32_t* col_off,
                                const int64_t* grad_off, const float* row_mask, const float* cont_mask, const float* dout,
                                float* dtab_flat, float* dcont, const int32_t* blk_col, const int32_t* blk_first, int32_t n_scan_blocks,
                                int32_t max_dim, int64_t bs, int32_t ncat, int32_t cat_width, int32_t n_cont, int32_t ld_out,
                                void* stream) {
  NNL_CHECK_ARG(bs > 0 && ncat > 0 && cat_width >= 0 && n_cont >= 0 && ld_out >= cat_width + n_cont && n_scan_blocks > 0,
                "tab_scan_bwd: bad sizes");
  NNL_CHECK_ARG(max_dim >= 1 && max_dim <= kScanMaxDim, "tab_scan_bwd: embedding width %d > %d (use nnl_tab_scatter_bwd)", max_dim, kScanMaxDim);
  NNL_CHECK_ARG(dout && xcat && card && dim && col_off && grad_off && dtab_flat && blk_col && blk_first, "tab_scan_bwd: null pointer");
  hipStream_t s = (hipStream_t)stream;
  NnlProfScope prof(NNL_PROF_TABULAR, s, (double)bs * (8.0 * ncat + 8.0 * (cat_width + n_cont)));
  const long cont_blocks = (dcont && n_cont > 0) ? nnl_cdiv(bs * n_cont, 256) : 0;
  hipLaunchKernelGGL(tab_scan_bwd_kernel, dim3((unsigned)(n_scan_blocks + cont_blocks)), dim3(256), 0, s, xcat, card, dim, col_off, grad_off,
                     row_mask, cont_mask, dout, dtab_flat, dcont, blk_col, blk_first, n_scan_blocks, (long)bs, ncat, cat_width, n_cont, ld_out);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

extern "C" int nnl_pad_cols(const float* src, float* dst, int64_t rows, int64_t C, int64_t Cp, void* stream) {
  NNL_CHECK_ARG(rows >= 0 && C > 0 && Cp >= C && Cp < (1 << 24), "pad_cols: bad sizes");
  if (rows == 0) return NNL_OK;
  NNL_CHECK_ARG(src && dst, "pad_cols: null pointer");
  hipLaunchKernelGGL(pad_cols_kernel, dim3(grid_for(rows * Cp)), dim3(kBlock), 0, (hipStream_t)stream, src, dst, (long)rows, (int)C, (int)Cp);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}

extern "C" int nnl_keep_masks(const float* u, float* a, int64_t na, float keep_a, float* b, int64_t nb, float keep_b, void* stream) {
  NNL_CHECK_ARG(na >= 0 && nb >= 0 && (na == 0 || (keep_a > 0.f && keep_a <= 1.f)) && (nb == 0 || (keep_b > 0.f && keep_b <= 1.f)),
                "keep_masks: bad arguments");
  if (na + nb == 0) return NNL_OK;
  NNL_CHECK_ARG(u && (na == 0 || a) && (nb == 0 || b), "keep_masks: null pointer");
  hipLaunchKernelGGL(keep_masks_kernel, dim3(grid_for(na + nb)), dim3(kBlock), 0, (hipStream_t)stream, u, a, (long)na, keep_a, b, (long)nb,
                     keep_b);
  NNL_CHECK_LAUNCH();
  return NNL_OK;
}
