"""ctypes binding of ``librwk.so`` (the C ABI declared in ``include/rwk.h``).

There is no fallback: if the shared library is missing or a call fails, an exception is
raised.  Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C gnn_amd/csrc``.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librwk.so")

_P = c_void_p          # every device pointer travels as void*
_I64 = c_int64

#: name -> (restype, argtypes); kept in one table so the tests can check it against rwk.h
SIGNATURES = {
    "rwk_version": (c_int, []),
    "rwk_error_string": (ctypes.c_char_p, [c_int]),
    "rwk_coo_inspect": (c_int, [_P, _P, _I64, _I64, _P, _P]),
    "rwk_csr_from_sorted_coo": (c_int, [_P, _P, _P, _I64, _I64, _P, _P, _P, _P]),
    "rwk_csr_count_asymmetric": (c_int, [_P, _P, _P, _I64, _P, _P]),
    "rwk_indicator_inspect": (c_int, [_P, _I64, _P, _P]),
    "rwk_segments_build": (c_int, [_P, _I64, _I64, _P, _P, _P, _P, _P, _P]),
    "rwk_batch_gather_coo": (c_int, [_P, _P, _P, _P, _P, _P, _I64, _P, _P, _P, _P]),
    "rwk_batch_build": (c_int, [_P, _P, _P, _P, _I64, _P, _P, _I64, _I64, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                _P, _P, _P, _P, _P]),
    "rwk_spmm_csr": (c_int, [_P, _P, _P, _I64, _I64, _P, _P, c_int, c_int32, _P, _I64, c_int32, _P]),
    "rwk_csr_compact": (c_int, [_P, _P, _P, _I64, _P, _P, _P]),
    "rwk_spmm_csr_compact": (c_int, [_P, _P, _P, _I64, _I64, _P, _P, c_int, _P]),
    "rwk_spmm_tile_reduce_rows": (c_int, [_P, _P, _P, _P, _I64, _I64, _P, _P, c_int, _P, _I64, _I64, _P, _P]),
    "rwk_hidden_forward": (c_int, [_P, _P, _I64, _I64, _I64, _I64, _P, _P, _P, _P, _P]),
    "rwk_hidden_backward": (c_int, [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _P, _P, _P]),
    "rwk_fused_n_streams": (_I64, [_I64, _I64]),
    "rwk_fused_k_partials_size": (_I64, [_I64, _I64, _I64, _I64, _I64]),
    "rwk_fused_dz_partials_count": (_I64, [_I64, _I64, _I64, _I64, _I64, _I64]),
    "rwk_fused_forward": (c_int, [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, c_float, _P, _P, _P]),
    "rwk_fused_reduce": (c_int, [_P, _P, _P, _I64, _I64, _I64, _I64, c_float, c_int, _P, _P]),
    "rwk_fused_backward": (c_int, [_P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, c_float, c_int,
                                   _P, _P, _P, _P, _P]),
    "rwk_fused_backward_workspace": (_I64, [_I64, _I64, _I64, _I64, _I64]),
    "rwk_fused_keep_words": (_I64, [_I64, _I64, _I64, _I64, _I64]),
    "rwk_fused_forward_bits": (c_int, [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, c_float, _P, _P, _P, _P]),
    "rwk_fused_backward_bits": (c_int, [_P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, c_float, c_int,
                                        _P, _P, _P, _P, _P, _P]),
    "rwk_fused_path": (c_int, [_I64, _I64, _I64, _I64]),
    "rwk_fused_stream_len": (_I64, [_I64, _I64, _I64, _I64, _I64, c_int]),
    "rwk_fused_keep_s_supported": (c_int, [_I64, _I64, _I64, _I64]),
    "rwk_fused_forward_keep_s": (c_int, [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, c_float, _P, _P, _P]),
    "rwk_fused_backward_keep_s": (c_int, [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, c_float, _P, _P, _P,
                                          _P]),
    "rwk_head_supported": (c_int, [_I64, _I64, _I64, _I64]),
    "rwk_head_n_blocks": (_I64, [_I64]),
    "rwk_head_stats_size": (_I64, [_I64, _I64]),
    "rwk_head_grad_size": (_I64, [_I64, _I64, _I64]),
    "rwk_head_forward": (c_int, [_P, _I64, _I64, _I64, _I64, _P, _P, _P, _P, _P, c_float, c_float, c_int, _P,
                                 _P, _P, _P, _P, c_float, _P, _P, _P, _P, _P, _P, _P]),
    "rwk_head_backward": (c_int, [_P, _I64, _I64, _I64, _I64, _P, _P, _P, _P, c_int, _P, _P, _P, _P, _P, c_float,
                                  _P, _P, _P, _P, _P]),
    "rwk_head_forward_loss": (c_int, [_P, _I64, _I64, _I64, _I64, _P, _P, _P, _P, _P, c_float, c_float, c_int, _P,
                                      _P, _P, _P, _P, c_float, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "rwk_head_backward_loss": (c_int, [_P, _I64, _I64, _I64, _I64, _P, _P, _P, _P, c_int, _P, _P, _P, _P, _P, c_float,
                                       _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "rwk_khop_count": (c_int, [_P, _P, _P, _P, _I64, _I64, _I64, _P, _P, _P]),
    "rwk_khop_fill": (c_int, [_P, _P, _P, _P, _I64, _I64, _I64, _P, _P, _P, _P]),
    "rwk_segment_sum": (c_int, [_P, _P, _I64, _I64, _P, _P]),
    "rwk_distance_bias_combine": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_float, _I64, _I64, _P, _P]),
    "rwk_row_scale": (c_int, [_P, _P, _I64, _I64, _P, _P]),
    "rwk_cross_entropy_forward": (c_int, [_P, _P, _I64, _I64, _P, _P]),
    "rwk_cross_entropy_backward": (c_int, [_P, _P, _I64, _I64, _P, _P, _P]),
    "rwk_project_forward": (c_int, [_P, _P, _P, _I64, _I64, _I64, _P, _P]),
    "rwk_project_backward_blocks": (_I64, [_I64, _I64]),
    "rwk_project_backward": (c_int, [_P, _P, _P, _P, _I64, _I64, _I64, _P, _P, _P]),
    "rwk_linear_forward": (c_int, [_P, _P, _P, _I64, _I64, _I64, _P, _P]),
    "rwk_linear_backward": (c_int, [_P, _P, _P, _I64, _I64, _I64, _P, _P, _P]),
    "rwk_project_hidden_forward": (c_int, [_P, _P, _P, _I64, _I64, _I64, _P, _P, _P, _I64, _I64, _I64, _P, _P, _P, _P,
                                           _P]),
    "rwk_project_hidden_backward": (c_int, [_P, _P, _P, _P, _I64, _I64, _I64, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64,
                                            _I64, _P, _P, _P]),
    "rwk_spmm_project_backward_blocks": (_I64, [_I64]),
    "rwk_spmm_project_backward": (c_int, [_P, _P, _P, _P, _P, _I64, _I64, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _P,
                                          _P]),
    "rwk_hidden_backward_reduce": (c_int, [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _P, _P, _P, _I64, _I64, _P,
                                           _P]),
    "rwk_step_tail_n_counters": (_I64, [_I64, _I64, _I64]),
    "rwk_hidden_backward_reduce_adam": (c_int, [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _P, _P, _P, _I64, _I64,
                                                _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _P, _P, _I64, _I64, _P,
                                                c_float, c_float, c_float, c_float, _P, _P, _P]),
    "rwk_step_tail_n_counters_head": (_I64, [_I64, _I64, _I64]),
    "rwk_hidden_backward_reduce_adam_head": (c_int, [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _P, _P, _P, _I64,
                                                     _I64, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _P, _P, _I64,
                                                     _I64, _P, c_float, c_float, c_float, c_float, _P, _P, _P, _I64, _I64,
                                                     _P, _P, _P, _P, _I64, _P]),
    "rwk_fused_backward_bn_supported": (c_int, [_I64, _I64, _I64, _I64]),
    "rwk_fused_backward_bn": (c_int, [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, c_float, c_int, _P, _P,
                                      _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _P, _I64, _P]),
    "rwk_fused_reduce_stats_rows": (_I64, [_I64, _I64]),
    "rwk_fused_reduce_stats_blocks": (_I64, [_I64, _I64]),
    "rwk_fused_reduce_stats": (c_int, [_P, _P, _P, _I64, _I64, _I64, _I64, c_float, c_int, _P, _P, _P, _P, _P]),
    "rwk_head_apply_loss": (c_int, [_P, _I64, _I64, _I64, _I64, _P, _P, _P, _P, _P, c_float, c_float, _P, _I64, _P,
                                    _P, _P, _P, c_float, _P, _P, _P, _P, _P, _P, _P, _P]),
    "rwk_head_train_supported": (c_int, [_I64, _I64, _I64, _I64]),
    "rwk_head_train_forward": (c_int, [_P, _I64, _I64, _I64, _I64, _P, _P, _P, _P, _P, c_float, c_float, _P, _I64, _P,
                                       _P, _P, _P, c_float, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "rwk_head_train_backward": (c_int, [_P, _I64, _I64, _I64, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "rwk_reduce_rows_workspace": (_I64, [_I64, _I64]),
    "rwk_reduce_rows": (c_int, [_P, _I64, _I64, _P, _P, _P]),
    "rwk_reduce_rows_level1": (_I64, [_I64]),
    "rwk_reduce_rows_to32": (c_int, [_P, _I64, _I64, _P, _P]),
    "rwk_reduce_rows_static": (c_int, [_P, _I64, _I64, _P, _P]),
    "rwk_gram_supported": (c_int, [_I64, _I64, _I64, _I64]),
    "rwk_gram_forward": (c_int, [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, c_int, _P, _P, _P]),
    "rwk_gram_backward": (c_int, [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, c_int, _P, _P, _P]),
    "rwk_eval_forward_lds": (_I64, [_I64] * 8),
    "rwk_eval_forward": (c_int, [_P, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _I64,
                                 c_int, _P, _P, _P, _P, c_float, _P, _P, _P, _P, _I64, _I64, _P, _P, _P]),
    "rwk_eval_forward_gather": (c_int, [_P, _P, _P, _P, _I64, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, c_int,
                                        _P, _P, _P, _P, c_float, _P, _P, _P, _P, _I64, _I64, _P, _P, _P]),
    "rwk_x_chain_lds": (_I64, [_I64, _I64, _I64]),
    "rwk_x_chain_hidden_forward": (c_int, [_P, _P, _P, _I64, _I64, _P, _P, _P, _P, _I64, _I64, _I64, _P, _P, _P, _I64,
                                           _I64, _P, _P, _P, _P, _P]),
    "rwk_x_chain_hidden_forward_gather": (c_int, [_P, _P, _P, _I64, _I64, _I64, _I64, _I64, _P, _P, _P, _I64, _I64,
                                                  _P, _P, _P, _P, _P]),
    "rwk_graph_chain_forward_lds": (_I64, [_I64, _I64, _I64]),
    "rwk_graph_chain_forward": (c_int, [_P, _P, _P, _I64, _I64, _P, _P, _P, _P, _I64, _I64, _I64, _P, _P, _P, _I64,
                                        _I64, _P, _P, _P, _P, _P]),
    "rwk_graph_chain_backward_lds": (_I64, [_I64, _I64]),
    "rwk_graph_chain_backward": (c_int, [_P, _P, _P, _P, _I64, _I64, _P, _P, _P, _P, _I64, _I64, _I64, _P, _P, _P,
                                         _I64, _I64, _P, _P]),
    "rwk_gather_chunks": (c_int, [_P, _P, _I64, _P, _P]),
    "rwk_adam_n_counters": (_I64, [_I64]),
    "rwk_adam_step": (c_int, [_P, _P, _P, _P, _I64, _P, c_float, c_float, c_float, c_float, _P, _P]),
    "rwk_adam_step_chunks": (c_int, [_P, _P, _P, _I64, _P, _P, _P, _I64, _P, c_float, c_float, c_float, c_float, _P,
                                     _P]),
    "rwk_dropout_mask": (c_int, [_P, _I64, _I64, _I64, _I64, c_float, _P, _P]),
    "rwk_model_plan": (c_int, [_P, _P]),
    "rwk_model_forward": (c_int, [_P, _P, _P, _P, _P]),
    "rwk_model_backward": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P]),
    "rwk_model_train_step": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "rwk_model_train_pass_plan": (c_int, [_P, _P, _I64, _P, _I64, _P, _P]),
    "rwk_model_train_pass": (c_int, [_P, _P, _P, _I64, _P, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int]),
}

_lib = None


class RwkError(RuntimeError):
    pass


def load():
    """Load ``librwk.so`` once and attach the prototypes.  Raises ``ImportError`` when absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built "
            "(run `make -C gnn_amd/csrc` or `__graft_entry__.build()`); there is no CPU fallback")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)     # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(code: int, what: str):
    if code != 0:
        msg = load().rwk_error_string(code).decode()
        raise RwkError(f"{what} failed: {msg} (code {code})")


def ptr(t):
    """Device pointer of a tensor (or None)."""
    return None if t is None else t.data_ptr()


def current_stream(device):
    """Raw ``hipStream_t`` of torch's current stream on ``device`` (no Stream object is built: this is called
    several times per training step)."""
    import torch
    idx = device.index if device.index is not None else torch.cuda.current_device()
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if raw is not None:
        return raw(idx)
    return torch.cuda.current_stream(device).cuda_stream


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def device_guard(device):
    """``torch.cuda.device(device)``, or a no-op context when ``device`` is already the current one (the usual
    case: one process per GPU; autograd's worker thread runs a backward on the device of its inputs)."""
    import torch
    if device.index is None or torch.cuda.current_device() == device.index:
        return _NO_GUARD
    return torch.cuda.device(device)
