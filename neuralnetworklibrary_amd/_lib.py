"""ctypes binding of libnnl_hip.so (the C ABI declared in include/nnl.h).

The product path has NO CPU fallback: if the shared library is missing or a symbol cannot be bound this
module raises at import time, and every op in ops.py refuses non-CUDA tensors.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must be imported first: libnnl_hip.so binds to the libamdhip64.so.7 torch loaded)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('NNL_LIB_PATH') or os.path.join(_HERE, 'libnnl_hip.so')   # NNL_LIB_PATH: A/B another BUILD of the same ABI

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(or `make -C neuralnetworklibrary_amd/csrc`). There is no CPU fallback for the HIP hot path.")

lib = C.CDLL(LIB_PATH)

c_p = C.c_void_p
i32, i64, f32, f64 = C.c_int32, C.c_int64, C.c_float, C.c_double
sz = C.c_size_t


class ConvGeom(C.Structure):
    """nnl_conv_geom_t (include/nnl.h)."""
    _fields_ = [('N', i32), ('H', i32), ('W', i32), ('C', i32), ('K', i32), ('R', i32), ('S', i32),
                ('stride', i32), ('pad', i32), ('P', i32), ('Q', i32)]


# name -> (restype, argtypes).  Every symbol declared in include/nnl.h must be listed here
# (tests/test_abi.py cross-checks the header against this table and against the .so exports).
SIGNATURES = {
    'nnl_version': (C.c_int, []),
    'nnl_last_error': (C.c_char_p, []),
    'nnl_reload_env': (C.c_int, []),
    'nnl_prof_enable': (C.c_int, [C.c_int]),
    'nnl_prof_collect': (C.c_int, [C.POINTER(i64), C.POINTER(f64), C.POINTER(f64)]),
    'nnl_prof_collect2': (C.c_int, [C.POINTER(i64), C.POINTER(f64), C.POINTER(f64), C.POINTER(f64)]),
    'nnl_source_stamp': (C.c_char_p, []),
    'nnl_dp_bump': (C.c_int, [c_p, c_p]),
    'nnl_dp_signal': (C.c_int, [c_p, c_p, c_p]),
    'nnl_dp_wait': (C.c_int, [c_p, i32, i64, c_p, c_p]),
    'nnl_embdotbias_fwd': (C.c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, i64, i64, i64, i64, C.c_int, f32, f32,
                                     c_p, c_p]),
    'nnl_embdotbias_bwd_workspace_bytes': (sz, [i64]),
    'nnl_embdotbias_bwd': (C.c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, i64, i64, i64, i64, C.c_int,
                                     f32, f32, c_p, sz, c_p]),
    'nnl_conv2d_fwd_workspace_bytes': (sz, [C.POINTER(ConvGeom)]),
    'nnl_conv2d_dgrad_workspace_bytes': (sz, [C.POINTER(ConvGeom)]),
    'nnl_conv2d_tile_counters': (i64, []),
    'nnl_conv2d_wino_preferred': (C.c_int, [C.POINTER(ConvGeom), C.c_int]),
    'nnl_wino_filter_multi': (C.c_int, [c_p, c_p, i64, c_p]),
    'nnl_conv2d_fwd_pre': (C.c_int, [c_p, c_p, c_p, c_p, C.POINTER(ConvGeom), C.c_int, c_p, sz, c_p, c_p, c_p, C.POINTER(i32), c_p, c_p]),
    'nnl_conv2d_dgrad_pre': (C.c_int, [c_p, c_p, c_p, C.POINTER(ConvGeom), c_p, c_p, sz, c_p, c_p, c_p]),
    'nnl_debug_conv_wino_workspace_bytes': (sz, [C.c_int] * 5),
    'nnl_debug_conv_wino_fwd': (C.c_int, [c_p] * 6 + [sz, c_p, C.c_long, c_p, c_p] + [C.c_int] * 7 + [c_p]),
    'nnl_debug_conv_wino2_workspace_bytes': (sz, [C.c_int] * 5),
    'nnl_debug_conv_wino2_fwd': (C.c_int, [c_p] * 6 + [sz, c_p, C.c_long, c_p, c_p] + [C.c_int] * 7 + [c_p]),
    'nnl_debug_conv_plan_times': (C.c_int, [C.c_int] * 5 + [C.POINTER(C.c_double)]),
    'nnl_conv2d_fwd': (C.c_int, [c_p, c_p, c_p, c_p, C.POINTER(ConvGeom), C.c_int, c_p, sz, c_p, c_p, c_p, C.POINTER(i32), c_p]),
    'nnl_conv2d_weight_transpose': (C.c_int, [c_p, c_p, C.c_int, C.c_int, C.c_int, C.c_int, c_p]),
    'nnl_conv2d_weight_transpose_multi': (C.c_int, [c_p, c_p, i64, f64, c_p]),
    'nnl_conv2d_dgrad': (C.c_int, [c_p, c_p, c_p, C.POINTER(ConvGeom), c_p, c_p, sz, c_p, c_p]),
    'nnl_conv2d_wgrad_workspace_bytes': (sz, [C.POINTER(ConvGeom)]),
    'nnl_conv2d_wgrad': (C.c_int, [c_p, c_p, c_p, C.POINTER(ConvGeom), c_p, sz, c_p]),
    'nnl_colsum_workspace_bytes': (sz, [i64, i64]),
    'nnl_colsum': (C.c_int, [c_p, c_p, i64, i64, c_p, sz, c_p]),
    'nnl_relu_gate_colsum': (C.c_int, [c_p, c_p, c_p, c_p, i64, i64, c_p, sz, c_p]),
    'nnl_conv2d_fwd_add_up2': (C.c_int, [c_p, c_p, c_p, c_p, c_p, C.POINTER(ConvGeom), c_p]),
    'nnl_upsample2_bwd': (C.c_int, [c_p, c_p, i64, i64, i64, i64, c_p]),
    'nnl_act_gate_colsum': (C.c_int, [c_p, c_p, c_p, c_p, i64, i64, C.c_int, c_p, sz, c_p]),
    'nnl_maxpool2d_fwd': (C.c_int, [c_p, c_p, c_p, i64, i64, i64, i64, i64, i64, C.c_int, C.c_int, C.c_int, c_p]),
    'nnl_maxpool2d_bwd': (C.c_int, [c_p, c_p, c_p, i64, i64, i64, i64, i64, i64, C.c_int, C.c_int, C.c_int, c_p]),
    'nnl_sum_tensors': (C.c_int, [c_p, C.c_int, c_p, i64, c_p]),
    'nnl_bn_relu_maxpool_supported': (C.c_int, [i64]),
    'nnl_bn_relu_maxpool_fwd': (C.c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, i64, i64, i64, i64, i64, i64,
                                          C.c_int, C.c_int, C.c_int, f32, f32, C.c_int, c_p, c_p, sz, c_p]),
    'nnl_bn_relu_maxpool_bwd': (C.c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, i64, i64, i64, i64, i64, i64,
                                          C.c_int, C.c_int, C.c_int, C.c_int, c_p, sz, c_p]),
    'nnl_concat_pool_fwd': (C.c_int, [c_p, c_p, c_p, i64, i64, i64, c_p]),
    'nnl_concat_pool_bwd': (C.c_int, [c_p, c_p, c_p, i64, i64, i64, c_p]),
    'nnl_bbox_decode': (C.c_int, [c_p, c_p, c_p, i64, i64, i64, c_p, c_p, f32, f32, f32, c_p, c_p, c_p, c_p, c_p, c_p]),
    'nnl_nms_workspace_bytes': (sz, [i64, i64]),
    'nnl_nms': (C.c_int, [c_p, c_p, c_p, c_p, c_p, i64, i64, i64, f32, c_p, c_p, c_p, c_p, c_p, sz, c_p]),
    'nnl_tab_renorm': (C.c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, i64, i32, i32, f32, c_p, c_p]),
    'nnl_tab_gather_fwd': (C.c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, i64, i32, i32, i32, i32, c_p]),
    'nnl_tab_scatter_bwd_workspace_bytes': (sz, [i64, i32]),
    'nnl_tab_scatter_bwd': (C.c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, i64, c_p, i64, i32, i32, i32, i32, c_p, sz, c_p]),
    'nnl_tab_scan_bwd': (C.c_int, [c_p] * 12 + [i32, i32, i64, i32, i32, i32, i32, c_p]),
    'nnl_pad_cols': (C.c_int, [c_p, c_p, i64, i64, i64, c_p]),
    'nnl_keep_masks': (C.c_int, [c_p, c_p, i64, f32, c_p, i64, f32, c_p]),
    'nnl_linear_small_supported': (C.c_int, [i64]),
    'nnl_linear_small_fwd': (C.c_int, [c_p, c_p, c_p, c_p, i64, i64, i64, i64, c_p]),
    'nnl_linear_small_bwd_workspace_bytes': (sz, [i64, i64, i64]),
    'nnl_linear_small_bwd': (C.c_int, [c_p, c_p, c_p, c_p, c_p, c_p, i64, i64, i64, i64, c_p, sz, c_p]),
    'nnl_retina_loss_workspace_bytes': (sz, [i64, i64]),
    'nnl_retina_loss_fwd': (C.c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, i64, i64, i64, i64, f32, f32, f32, c_p, sz, c_p]),
    'nnl_retina_loss_bwd': (C.c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, i64, i64, i64, i64, f32, f32, f32, c_p]),
    'nnl_lstm_padded_hidden': (i64, [i64]),
    'nnl_lstm_padded_gates': (i64, [i64]),
    'nnl_lstm_workspace_bytes': (sz, [i64, i64, i64]),
    'nnl_lstm_fwd': (C.c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, i64, i64, i64, c_p, sz, c_p, c_p]),
    'nnl_lstm_bwd': (C.c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, i64, i64, i64, c_p, sz, c_p, c_p]),
    'nnl_debug_lstm_bptt2_plan': (C.c_int, [i64, i64, c_p]),
    'nnl_embedding_rowmask_fwd': (C.c_int, [c_p, c_p, c_p, c_p, i64, i64, i64, c_p, c_p]),
    'nnl_embedding_rowmask_bwd_workspace_bytes': (sz, [i64]),
    'nnl_embedding_rowmask_bwd': (C.c_int, [c_p, c_p, c_p, c_p, i64, i64, i64, i64, c_p, sz, c_p]),
    'nnl_softmax_ce_fwd': (C.c_int, [c_p, c_p, c_p, c_p, c_p, i64, i64, c_p, c_p]),
    'nnl_softmax_ce_bwd': (C.c_int, [c_p, c_p, c_p, c_p, c_p, i64, i64, i64, c_p]),
    'nnl_mse_workspace_bytes': (sz, [i64]),
    'nnl_mse_fwd': (C.c_int, [c_p, c_p, c_p, i64, c_p, sz, c_p]),
    'nnl_mse_bwd': (C.c_int, [c_p, c_p, c_p, c_p, i64, c_p]),
    'nnl_scaled_sigmoid_fwd': (C.c_int, [c_p, c_p, c_p, i64, C.c_float, C.c_float, c_p]),
    'nnl_scaled_sigmoid_bwd': (C.c_int, [c_p, c_p, c_p, i64, C.c_float, C.c_float, c_p]),
    'nnl_seq_reg_workspace_bytes': (sz, [i64, i64]),
    'nnl_seq_reg_fwd': (C.c_int, [c_p, c_p, i64, i64, C.c_float, C.c_float, c_p, sz, c_p]),
    'nnl_seq_reg_bwd': (C.c_int, [c_p, c_p, c_p, i64, i64, C.c_float, C.c_float, c_p]),
    'nnl_weight_drop': (C.c_int, [c_p, i64, c_p, c_p, i64, i64, i64, C.c_uint64, C.c_float, c_p]),
    'nnl_optim_chunk_elems': (i64, []),
    'nnl_optim_patch': (C.c_int, [c_p, c_p, c_p, i64, c_p]),
    'nnl_optim_step': (C.c_int, [c_p, c_p, c_p, i64, C.c_int, c_p, C.c_int, c_p, c_p]),
    'nnl_bn_workspace_bytes': (sz, [i64, i64]),
    'nnl_bn_fwd': (C.c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, i64, i64, f32, f32, C.c_int, C.c_int, c_p, c_p, c_p, i64,
                             c_p, c_p, c_p, sz, c_p]),
    'nnl_bn_sync_stats': (C.c_int, [c_p, c_p, i64, i64, c_p, sz, c_p]),
    'nnl_bn_sync_fwd': (C.c_int, [c_p, c_p, C.c_int, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, i64, i64, f32, f32, C.c_int, c_p,
                                  c_p, c_p, sz, c_p]),
    'nnl_bn_sync_bwd_reduce': (C.c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, i64, i64, C.c_int, c_p, sz, c_p]),
    'nnl_bn_sync_bwd': (C.c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, C.c_int, c_p, c_p, c_p, c_p, i64, i64,
                                  C.c_int, c_p, sz, c_p]),
    'nnl_bn_bwd': (C.c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, i64, i64, C.c_int, C.c_int, c_p, sz, c_p]),
}

for _name, (_res, _args) in SIGNATURES.items():
    try:
        _fn = getattr(lib, _name)
    except AttributeError as e:  # fail loudly: a stale .so must not silently lose an op
        raise ImportError(f"libnnl_hip.so does not export {_name}; rebuild it") from e
    _fn.restype = _res
    _fn.argtypes = _args

PROF_KINDS = ['conv_fwd', 'conv_dgrad', 'conv_wgrad', 'embdot', 'tabular', 'retina_loss', 'lstm',
              'softmax_ce', 'elementwise', 'gemm', 'optim']


class NnlError(RuntimeError):
    pass


def check(status):
    if status != 0:
        raise NnlError(f"nnl error {status}: {lib.nnl_last_error().decode()}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise NnlError("nnl HIP op called with a non-CUDA tensor: the MI355X path has no CPU fallback")


def prof_enable(flag=True):
    check(lib.nnl_prof_enable(1 if flag else 0))


def prof_collect():
    n = len(PROF_KINDS)
    a, b, c, d = (i64 * n)(), (f64 * n)(), (f64 * n)(), (f64 * n)()
    check(lib.nnl_prof_collect2(a, b, c, d))
    return {k: {'launches': int(a[i]), 'ms': float(b[i]), 'work': float(c[i]), 'exec': float(d[i])} for i, k in enumerate(PROF_KINDS)}


def source_stamp_of_tree():
    """the stamp csrc/Makefile compiles into the library, recomputed from the sources next to this file: sha256 over the sorted csrc/*.hip and
    csrc/*.h, the Makefile and include/nnl.h, first 16 hex digits"""
    import glob
    import hashlib
    csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
    names = sorted(os.path.basename(p) for p in glob.glob(os.path.join(csrc, '*.hip')) + glob.glob(os.path.join(csrc, '*.h')))
    hsh = hashlib.sha256()
    for p in [os.path.join(csrc, n) for n in names] + [os.path.join(csrc, 'Makefile'), os.path.join(csrc, '..', '..', 'include', 'nnl.h')]:
        with open(p, 'rb') as f:
            hsh.update(f.read())
    return hsh.hexdigest()[:16]


def source_stamp():
    return lib.nnl_source_stamp().decode()
