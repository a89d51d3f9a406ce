"""The training-mode forward and backward of ``RW_NN`` as ONE autograd node with one library call per direction
(``rwk_model_forward`` / ``rwk_model_backward``): what ``model(adj, features, graph_indicator)`` followed by
``loss.backward()`` costs in the reference's eager loop (``rwgnn2/main.py:82-88``).

The launches are those of :mod:`gnn_amd.function` and :mod:`gnn_amd.head`, in their order, so every result has their
bits; what goes is host work: two autograd nodes, a dozen library calls and some thirty ``torch.empty`` per step become
one node, two calls and four allocations (the two workspaces, the output, the gradient vector).
"""
from __future__ import annotations

import ctypes
from ctypes import c_float, c_int32, c_int64, c_void_p

import torch

from . import _lib

__all__ = ["COMPOSITE_STEP", "model_step_supported", "model_step"]

#: False keeps the separate autograd nodes of the layer and the head (tests compare the two)
COMPOSITE_STEP = True


class ModelStep(ctypes.Structure):
    """``rwk_model_step_t`` of include/rwk.h."""
    _fields_ = [
        ("n", c_int64), ("f", c_int64), ("nnz", c_int64), ("n_graphs", c_int64), ("max_graph_nodes", c_int64),
        ("max_degree", c_int32), ("max_degree_t", c_int32),
        ("features", c_void_p), ("crow", c_void_p), ("col", c_void_p), ("val", c_void_p),
        ("crow_t", c_void_p), ("col_t", c_void_p), ("col16_t", c_void_p), ("val_t", c_void_p),
        ("unit_values_t", c_int32), ("long_threshold_t", c_int32),
        ("seg_ptr", c_void_p), ("gid", c_void_p), ("counts_f32", c_void_p),
        ("h", c_int64), ("s", c_int64), ("d", c_int64), ("max_step", c_int64), ("p", c_int64), ("nc", c_int64),
        ("normalize", c_int32), ("reserved", c_int32),
        ("drop_p", c_float), ("momentum", c_float), ("eps", c_float), ("reserved_f", c_float),
        ("fc_w", c_void_p), ("fc_b", c_void_p), ("adj_hidden", c_void_p), ("features_hidden", c_void_p),
        ("bn_w", c_void_p), ("bn_b", c_void_p), ("w1", c_void_p), ("b1", c_void_p), ("w2", c_void_p), ("b2", c_void_p),
        ("running_mean", c_void_p), ("running_var", c_void_p), ("num_batches_tracked", c_void_p),
        ("rng_state", c_void_p),
    ]


class ModelPlan(ctypes.Structure):
    """``rwk_model_plan_t`` of include/rwk.h."""
    _fields_ = [
        ("forward_bytes", c_int64), ("backward_bytes", c_int64), ("grad_floats", c_int64),
        ("stream_len_f", c_int64), ("stream_len_b", c_int64), ("n_partials", c_int64), ("dz_rows", c_int64),
        ("stats_blocks", c_int64), ("spb_blocks", c_int64), ("head_blocks", c_int64), ("head_grad_floats", c_int64),
        ("off_f", c_int64 * 12), ("off_b", c_int64 * 8),
    ]


_PARAM_FIELDS = ("fc_w", "fc_b", "adj_hidden", "features_hidden", "bn_w", "bn_b", "w1", "b1", "w2", "b2")


def _entry(model, st, f_in: int, drop: float):
    """(step struct, plan, split sizes) of one (batch structure, hyper-parameters) pair, or None when the composite
    calls do not serve it; computed once and kept with the structure (batches are reused every epoch)."""
    mods = model._modules                      # plain dict reads: nn.Module.__getattr__ costs ~0.3 us each
    h, s, d = model._parameters["features_hidden"].shape
    p, nc = mods["fc1"].out_features, mods["fc2"].out_features
    bn = mods["bn"]
    key = ("model_step", f_in, h, s, d, int(model.max_step), bool(model.normalize), p, nc, float(drop),
           float(bn.momentum), float(bn.eps))
    if key in st.plans:
        return st.plans[key]
    entry = None
    from .eval import _facts
    if st.perm is None and st.nnz > 0 and d == 4 and f_in <= 15:
        max_nodes, block_diagonal = _facts(st)
        if block_diagonal:
            lib = _lib.load()
            csr, csr_t = st.csr, st.csr_t
            m = ModelStep()
            m.n, m.f, m.nnz, m.n_graphs, m.max_graph_nodes = st.n, f_in, st.nnz, st.n_graphs, max_nodes
            m.max_degree, m.max_degree_t = int(csr.max_degree), int(csr_t.max_degree)
            m.crow, m.col = csr.crow.data_ptr(), csr.col.data_ptr()
            m.val = None if csr.unit_values else csr.val.data_ptr()
            m.crow_t, m.col_t = csr_t.crow.data_ptr(), csr_t.col.data_ptr()
            m.col16_t = _lib.ptr(csr_t.col16)
            m.val_t = _lib.ptr(csr_t.val)
            m.unit_values_t, m.long_threshold_t = int(bool(csr_t.unit_values)), int(csr_t.long_threshold or 0)
            m.seg_ptr, m.gid, m.counts_f32 = st.seg_ptr.data_ptr(), st.gid.data_ptr(), st.counts_f32.data_ptr()
            m.h, m.s, m.d, m.max_step, m.p, m.nc = h, s, d, int(model.max_step), p, nc
            m.normalize = int(bool(model.normalize))
            m.drop_p, m.momentum, m.eps = float(drop), float(bn.momentum), float(bn.eps)
            plan = ModelPlan()
            if lib.rwk_model_plan(ctypes.byref(m), ctypes.byref(plan)) == 0:
                n_pairs = s * (s - 1) // 2
                sizes = [h * n_pairs, h * s * d, d * f_in, d, nc * p, nc, p * h * int(model.max_step), p,
                         h * int(model.max_step), h * int(model.max_step)]
                entry = (m, plan, sizes, (st.csr, st.csr_t))       # the struct holds raw pointers: keep their owners
    st.plans[key] = entry
    return entry


def model_step_supported(model, st, features, drop: float):
    """The cached entry when ``model(adj, features, graph_indicator)`` in training mode can run as one node."""
    if not (COMPOSITE_STEP and features.is_cuda and features.dtype is torch.float32 and features.dim() == 2
            and not features.requires_grad):
        return None
    from . import function as fn
    if not (fn.MERGED_LAUNCHES and fn.FUSED_STEP_LAUNCHES and fn.GRAPH_X_CHAIN):
        return None
    if drop == 0.0 and fn.GRAM_FAST_PATH:
        return None                                    # without a mask the per-graph Gram form is cheaper
    return _entry(model, st, int(features.shape[1]), drop)


class _RWModelStep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, adj_hidden, features_hidden, fc_w, fc_b, bn_w, bn_b, w1, b1, w2, b2, model, st, entry):
        lib = _lib.load()
        m, plan, sizes, _ = entry
        dev = features.device
        bn = model._modules["bn"]
        tensors = (fc_w, fc_b, adj_hidden, features_hidden, bn_w, bn_b, w1, b1, w2, b2)
        if not (features.is_contiguous() and all(t.is_contiguous() and t.dtype is torch.float32 for t in tensors)):
            raise RuntimeError("the composite step takes contiguous float32 parameters and features")
        if st.n_graphs < 2:
            raise ValueError("Expected more than 1 value per channel when training, got input size %s"
                             % ((st.n_graphs, bn.num_features),))
        with _lib.device_guard(dev):
            ws = torch.empty(plan.forward_bytes, dtype=torch.uint8, device=dev)
            out = torch.empty((st.n_graphs, m.nc), dtype=torch.float32, device=dev)
            m.features = features.data_ptr()
            for name, t in zip(_PARAM_FIELDS, tensors):
                setattr(m, name, t.data_ptr())
            buf = bn._buffers
            m.running_mean, m.running_var = buf["running_mean"].data_ptr(), buf["running_var"].data_ptr()
            m.num_batches_tracked = _lib.ptr(buf["num_batches_tracked"])
            m.rng_state = model._buffers["_rng_state"].data_ptr() if m.drop_p > 0.0 else None
            _lib.check(lib.rwk_model_forward(ctypes.byref(m), ctypes.byref(plan), ws.data_ptr(), out.data_ptr(),
                                             _lib.current_stream(dev)), "rwk_model_forward")
        ctx.entry = entry
        ctx.save_for_backward(ws, out, features, *tensors)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out):
        lib = _lib.load()
        m, plan, sizes, _ = ctx.entry
        saved = ctx.saved_tensors
        ws, out, features = saved[:3]
        dev = out.device
        with _lib.device_guard(dev):
            d_out = d_out.contiguous().float()
            bws = torch.empty(plan.backward_bytes, dtype=torch.uint8, device=dev)
            grads = torch.empty(plan.grad_floats, dtype=torch.float32, device=dev)
            # the struct is shared by every call on this (structure, hyper-parameters) pair: point it at this
            # node's own tensors again (another forward may have run in between)
            m.features = features.data_ptr()
            for name, t in zip(_PARAM_FIELDS, saved[3:]):
                setattr(m, name, t.data_ptr())
            _lib.check(lib.rwk_model_backward(ctypes.byref(m), ctypes.byref(plan), ws.data_ptr(), bws.data_ptr(),
                                              out.data_ptr(), d_out.data_ptr(), grads.data_ptr(),
                                              _lib.current_stream(dev)), "rwk_model_backward")
        h, s, d = m.h, m.s, m.d
        d_adj, d_feat, d_w, d_b, d_w2, d_b2, d_w1, d_b1, d_bn_w, d_bn_b = torch.split(grads, sizes)
        return (None, d_adj.view(h, -1), d_feat.view(h, s, d), d_w.view(d, m.f), d_b,
                d_bn_w, d_bn_b, d_w1.view(m.p, -1), d_b1, d_w2.view(m.nc, m.p), d_b2, None, None, None)


def model_step(model, st, features, entry):
    """``log_softmax`` output (G, n_classes) of the training-mode forward; gradients reach the ten parameters."""
    mods, own = model._modules, model._parameters
    fc, bn, fc1, fc2 = (mods[k]._parameters for k in ("fc", "bn", "fc1", "fc2"))
    return _RWModelStep.apply(features, own["adj_hidden"], own["features_hidden"], fc["weight"], fc["bias"],
                              bn["weight"], bn["bias"], fc1["weight"], fc1["bias"], fc2["weight"], fc2["bias"],
                              model, st, entry)
