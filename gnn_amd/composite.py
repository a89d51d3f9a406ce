"""The training-mode forward and backward of ``RW_NN`` as ONE autograd node with one library call per direction
(``rwk_model_forward`` / ``rwk_model_backward``): what ``model(adj, features, graph_indicator)`` followed by
``loss.backward()`` costs in the reference's eager loop (``rwgnn2/main.py:82-88``).

The launches are those of :mod:`gnn_amd.function` and :mod:`gnn_amd.head`, in their order, so every result has their
bits; what goes is host work: two autograd nodes, a dozen library calls and some thirty ``torch.empty`` per step become
one node, two calls and four allocations (the two workspaces, the output, the gradient vector).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int32, c_int64, c_void_p

import torch

from . import _lib

__all__ = ["COMPOSITE_STEP", "ONE_CALL_TRAIN_STEP", "model_step_supported", "model_step", "TrainStep"]

#: False keeps the separate autograd nodes of the layer and the head (tests compare the two)
COMPOSITE_STEP = True
#: ``Trainer``'s eager step as one library call without autograd (``rwk_model_train_step``); False keeps the autograd
#: path (tests compare the two)
ONE_CALL_TRAIN_STEP = True


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
        ("off_f", c_int64 * 12), ("off_b", c_int64 * 8), ("train_step", c_int64),
    ]


class TrainTail(ctypes.Structure):
    """``rwk_train_tail_t`` of include/rwk.h."""
    _fields_ = [
        ("param", c_void_p), ("exp_avg", c_void_p), ("exp_avg_sq", c_void_p), ("grad_out", c_void_p),
        ("n", c_int64), ("off_adj", c_int64), ("off_feat", c_int64), ("off_red", c_int64), ("rest_off", c_int64),
        ("n_rest", c_int64), ("rest_columns", c_int64 * 8), ("rest_sizes", c_int64 * 8),
        ("lr", c_void_p), ("beta1", c_float), ("beta2", c_float), ("eps", c_float), ("weight_decay", c_float),
        ("step_counters", c_void_p), ("corr_slot", c_void_p),
    ]


class PassDataset(ctypes.Structure):
    """``rwk_pass_dataset_t`` of include/rwk.h."""
    _fields_ = [("row_ptr", c_void_p), ("col", c_void_p), ("val", c_void_p), ("features", c_void_p),
                ("n_feat", c_int64), ("y", c_void_p), ("unit_values", c_int32), ("reserved", c_int32)]


class PassSlot(ctypes.Structure):
    """``rwk_pass_slot_t`` of include/rwk.h."""
    _fields_ = [("feats", c_void_p), ("y", c_void_p), ("crow", c_void_p), ("col", c_void_p), ("val", c_void_p),
                ("col16", c_void_p), ("gid", c_void_p), ("seg_ptr", c_void_p), ("counts_f32", c_void_p)]


class PassBatch(ctypes.Structure):
    """``rwk_pass_batch_t`` of include/rwk.h."""
    _fields_ = [("b", c_int64), ("n", c_int64), ("nnz", c_int64), ("max_graph_nodes", c_int64),
                ("table_off", c_int64), ("max_degree", c_int32), ("fits16", c_int32), ("long_threshold", c_int32),
                ("reserved", c_int32)]


class PassNeed(ctypes.Structure):
    """``rwk_pass_need_t`` of include/rwk.h."""
    _fields_ = [("served", c_int64), ("train_step", c_int64), ("forward_bytes", c_int64),
                ("backward_bytes", c_int64), ("grad_floats", c_int64), ("out_floats", c_int64)]


#: where ``rwk_model_train_pass`` runs the assembly (include/rwk.h)
PASS_SAME_STREAM, PASS_SIDE_STREAM, PASS_IN_FORWARD = 0, 1, 2

#: the pass call of ``StreamTrainer(side_stream=None)`` lets the forward's first launch assemble the batch
#: (``rwk_x_chain_hidden_forward_gather``); False keeps the assembly launch in front of each step (tests and
#: measurements compare the two; RWK_ASSEMBLE_IN_FORWARD=0 in the environment does the same)
ASSEMBLE_IN_FORWARD = os.environ.get("RWK_ASSEMBLE_IN_FORWARD", "1") != "0"

#: ``StreamTrainer`` issues the steps of a pass through ``rwk_model_train_pass`` (one library call for all of them);
#: False keeps one ``rwk_model_train_step`` call per batch (tests compare the two)
ONE_CALL_PASS = True

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


class TrainStep:
    """``Trainer``'s single-process step (zero_grad, forward, cross-entropy, backward, Adam; ``rwgnn2/main.py:82-88``)
    as ONE library call, ``rwk_model_train_step``: no autograd graph, no allocation per step beyond the loss scalar.
    The launches are those of the autograd path with the optimiser's update in the last one (``optim.StepTail``), so
    parameters, moments, running statistics and generator state come out the same bits.  What an eager loop gains is
    host time: batches that change every step (``StreamTrainer``) cannot replay a captured graph, and launching the
    step through autograd costs the host four times what the device needs to run it."""

    def __init__(self, trainer):
        self.trainer = trainer
        self.tail_struct = None
        self.ws = [None, None, None, None]     # forward workspace, backward workspace, gradient vector, output
        self.flat_views = None

    def _room(self, k, numel, dtype, dev):
        t = self.ws[k]
        if t is None or t.numel() < numel:
            t = torch.empty(max(numel, 1), dtype=dtype, device=dev)
            self.ws[k] = t
        return t

    def supported(self, model, st, feats, y, drop):
        tr = self.trainer
        tail = tr.tail
        from . import function as fn
        if not (ONE_CALL_TRAIN_STEP and tail is not None and tr.loss_in_head and model.training
                and fn.FOLD_HEAD_FINISH and fn.MERGED_LAUNCHES and fn.FUSED_STEP_LAUNCHES and fn.GRAPH_X_CHAIN
                and not (drop == 0.0 and fn.GRAM_FAST_PATH) and drop < 1.0
                and feats.is_cuda and feats.dtype is torch.float32 and feats.dim() == 2 and feats.is_contiguous()
                and not feats.requires_grad and y.dtype is torch.int64 and y.dim() == 1 and y.is_contiguous()
                and y.numel() == st.n_graphs and torch.is_grad_enabled()):
            return None
        if not model._head_fusable(st.n_graphs, model.hidden_graphs * model.max_step, drop):
            return None
        for p in tr.bucket.params:             # nothing may expect autograd to deliver these gradients
            if getattr(p, "_backward_hooks", None) or getattr(p, "_post_accumulate_grad_hooks", None) \
                    or not p.is_contiguous():
                return None
        entry = _entry(model, st, int(feats.shape[1]), drop)
        if entry is None or not entry[1].train_step:
            return None
        return entry

    # ---- batches that live in the preallocated slots of a DeviceDataset (StreamTrainer) ------------------------
    def slots_supported(self, model, dataset) -> bool:
        """Once per pass: whether :meth:`run_slot` may serve the batches of ``dataset`` (each batch is still checked
        by ``rwk_model_plan``: a graph too large for the LDS, say, sends that one batch through the tensors)."""
        tr = self.trainer
        from . import function as fn
        drop = float(model._modules["dropout"].p) if model.training else 0.0
        if not (ONE_CALL_TRAIN_STEP and tr.tail is not None and tr.loss_in_head and not tr.forward_fn_given
                and hasattr(model, "_head_fusable") and model.training
                and fn.FOLD_HEAD_FINISH and fn.MERGED_LAUNCHES and fn.FUSED_STEP_LAUNCHES and fn.GRAPH_X_CHAIN
                and not (drop == 0.0 and fn.GRAM_FAST_PATH) and drop < 1.0 and torch.is_grad_enabled()
                and dataset.symmetric and 1 <= dataset.n_feat <= 15
                and model._parameters["features_hidden"].shape[2] == 4
                and model._modules["fc"].in_features == dataset.n_feat):
            return False
        for p in tr.bucket.params:
            if getattr(p, "_backward_hooks", None) or getattr(p, "_post_accumulate_grad_hooks", None) \
                    or not p.is_contiguous():
                return False
        return True

    def run_slot(self, model, dataset, slot, meta) -> bool:
        """The step on the batch that ``DeviceDataset.launch_planned`` left in ``slot`` (``meta``: its entry of the
        plan), straight from the slot's buffers: no tensor, no structure object, one ``rwk_model_plan`` and one
        ``rwk_model_train_step`` call.  False (nothing launched): the calls do not serve this batch."""
        from .structure import spmm_schedule
        lib = _lib.load()
        tr = self.trainer
        b, n, nnz, max_degree, fits16, _, y_lo, y_hi, max_graph = meta
        if nnz <= 0 or max_degree > 64:
            return False
        drop = float(model._modules["dropout"].p)
        m = getattr(slot, "step", None)
        if m is None:
            m = ModelStep()                    # the slot's buffers never move: their addresses are set once
            m.features = slot.feats.data_ptr()
            m.crow = m.crow_t = slot.crow.data_ptr()
            m.col = m.col_t = slot.col.data_ptr()
            m.val = None if dataset.unit_values else slot.val.data_ptr()
            m.unit_values_t = int(bool(dataset.unit_values))
            m.seg_ptr, m.gid, m.counts_f32 = slot.seg_ptr.data_ptr(), slot.gid.data_ptr(), slot.counts_f32.data_ptr()
            m.f, m.d = dataset.n_feat, 4
            slot.step = m
            slot.plan = ModelPlan()
        mods = model._modules
        bn = mods["bn"]
        h, s, _ = model._parameters["features_hidden"].shape
        m.n, m.nnz, m.n_graphs, m.max_graph_nodes = n, nnz, b, max_graph
        m.max_degree = m.max_degree_t = max_degree
        m.col16_t = slot.col16.data_ptr() if (fits16 and slot.col16 is not None) else None
        # (a lean assembly leaves the value array of an unweighted batch with compact columns unwritten)
        m.val_t = None if (m.col16_t and dataset.unit_values) else slot.val.data_ptr()
        m.long_threshold_t = int(spmm_schedule(n, nnz, max_degree)[1] or 0)
        m.h, m.s, m.max_step = h, s, int(model.max_step)
        m.p, m.nc = mods["fc1"].out_features, mods["fc2"].out_features
        m.normalize = int(bool(model.normalize))
        m.drop_p, m.momentum, m.eps = drop, float(bn.momentum), float(bn.eps)
        plan = slot.plan
        if bn.num_features != h * m.max_step or not bn.affine or not bn.track_running_stats or bn.momentum is None \
                or not model.fused_head or b < 2:
            return False
        if lib.rwk_model_plan(ctypes.byref(m), ctypes.byref(plan)) != 0 or not plan.train_step:
            return False
        if y_lo < 0 or y_hi >= m.nc:
            raise IndexError(f"Target {y_lo if y_lo < 0 else y_hi} is out of bounds for {m.nc} classes")
        tr.loss = self._launch(lib, model, m, plan, slot.y.data_ptr(), b, slot.feats.device)
        return True

    @staticmethod
    def _point_at_model(m, model):
        """The parameter, running-statistics and generator-state addresses of ``model`` into the step struct."""
        mods, own = model._modules, model._parameters
        fc, bn, fc1, fc2 = (mods[k]._parameters for k in ("fc", "bn", "fc1", "fc2"))
        m.adj_hidden, m.features_hidden = own["adj_hidden"].data_ptr(), own["features_hidden"].data_ptr()
        m.fc_w, m.fc_b = fc["weight"].data_ptr(), fc["bias"].data_ptr()
        m.bn_w, m.bn_b = bn["weight"].data_ptr(), bn["bias"].data_ptr()
        m.w1, m.b1 = fc1["weight"].data_ptr(), fc1["bias"].data_ptr()
        m.w2, m.b2 = fc2["weight"].data_ptr(), fc2["bias"].data_ptr()
        buf = mods["bn"]._buffers
        m.running_mean, m.running_var = buf["running_mean"].data_ptr(), buf["running_var"].data_ptr()
        m.num_batches_tracked = _lib.ptr(buf["num_batches_tracked"])
        m.rng_state = model._buffers["_rng_state"].data_ptr() if m.drop_p > 0.0 else None

    def _grads_to_params(self):
        """The parameters' gradients: the flat bucket the last launch filled, as zero_grad / backward would leave
        them."""
        tr = self.trainer
        views = self.flat_views
        if views is None or views[0] is not tr.bucket.flat:
            flat, off, vs = tr.bucket.flat, 0, []
            for p in tr.bucket.params:
                vs.append(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
            views = self.flat_views = (flat, vs)
        for p, v in zip(tr.bucket.params, views[1]):
            p.grad = v

    def run_pass_slots(self, model, dataset, slots, plan, t0: int, main_stream: int, side_stream,
                       in_forward: bool = False):
        """The steps of batches ``t0, t0 + 1, ...`` of ``plan``, each assembled into slot ``t % len(slots)`` right
        before it (``side_stream``: a raw handle, or None for the assembly on the launch stream - or, with
        ``in_forward``, inside the step's first launch: ``RWK_PASS_IN_FORWARD`` of include/rwk.h), as ONE library call
        (``rwk_model_train_pass``).  Returns (k, losses): the number of batches stepped - the call stops in front of a
        batch it does not serve, 0 when that is batch ``t0`` - and their losses as one tensor of k elements."""
        lib = _lib.load()
        tr = self.trainer
        mods = model._modules
        bn = mods["bn"]
        h, s, _ = model._parameters["features_hidden"].shape
        n_b = len(plan.metas) - t0
        depth = len(slots)
        if n_b <= 0 or depth > 8 or bn.num_features != h * int(model.max_step) or not bn.affine \
                or not bn.track_running_stats or bn.momentum is None or not model.fused_head:
            return 0, None
        dev = dataset.device
        drop = float(mods["dropout"].p)
        m = ModelStep()
        m.f, m.d, m.h, m.s, m.max_step = dataset.n_feat, 4, h, s, int(model.max_step)
        m.p, m.nc = mods["fc1"].out_features, mods["fc2"].out_features
        m.normalize = int(bool(model.normalize))
        m.drop_p, m.momentum, m.eps = drop, float(bn.momentum), float(bn.eps)
        if drop > 0.0 and not model._rng_seeded:
            model._seed_from_torch()
        self._point_at_model(m, model)
        ds = getattr(dataset, "_pass_struct", None)
        if ds is None:
            ds = PassDataset()
            ds.row_ptr, ds.col, ds.val = dataset.row_ptr.data_ptr(), _lib.ptr(dataset.col), _lib.ptr(dataset.val)
            ds.features, ds.n_feat, ds.y = dataset.features.data_ptr(), dataset.n_feat, dataset.y.data_ptr()
            ds.unit_values = int(bool(dataset.unit_values))
            dataset._pass_struct = ds
        cs = (PassSlot * depth)()
        for j in range(depth):                       # batch t0 + j goes to slot (t0 + j) % depth
            sl, c = slots[(t0 + j) % depth], cs[j]
            c.feats, c.y, c.crow, c.col, c.val = (sl.feats.data_ptr(), sl.y.data_ptr(), sl.crow.data_ptr(),
                                                  sl.col.data_ptr(), sl.val.data_ptr())
            c.col16 = _lib.ptr(sl.col16)
            c.gid, c.seg_ptr, c.counts_f32 = sl.gid.data_ptr(), sl.seg_ptr.data_ptr(), sl.counts_f32.data_ptr()
        batches = plan.pass_batches(PassBatch)
        first = ctypes.byref(batches, t0 * ctypes.sizeof(PassBatch))
        need = PassNeed()
        _lib.check(lib.rwk_model_train_pass_plan(ctypes.byref(ds), first, n_b, cs, depth, ctypes.byref(m),
                                                 ctypes.byref(need)), "rwk_model_train_pass_plan")
        k = int(need.served)
        if k == 0:
            return 0, None
        for meta in plan.metas[t0:t0 + k]:
            if meta[6] < 0 or meta[7] >= m.nc:
                raise IndexError(f"Target {meta[6] if meta[6] < 0 else meta[7]} is out of bounds for {m.nc} classes")
        with _lib.device_guard(dev):
            fws = self._room(0, need.forward_bytes, torch.uint8, dev)
            bws = self._room(1, need.backward_bytes, torch.uint8, dev)
            grads = self._room(2, need.grad_floats, torch.float32, dev)
            out = self._room(3, need.out_floats, torch.float32, dev)
            losses = torch.empty(k, dtype=torch.float32, device=dev)
            fold = need.train_step == 1
            t = self._tail(lib, "head" if fold else "plain")
            _lib.check(lib.rwk_model_train_pass(
                ctypes.byref(ds), plan.tables.data_ptr(), first, k, cs, depth, ctypes.byref(m), ctypes.byref(t),
                ctypes.byref(need), fws.data_ptr(), bws.data_ptr(), grads.data_ptr(), out.data_ptr(),
                tr._one.data_ptr(), losses.data_ptr(), main_stream, side_stream,
                PASS_SIDE_STREAM if side_stream is not None else (PASS_IN_FORWARD if in_forward else PASS_SAME_STREAM)),
                "rwk_model_train_pass")
        tr.tail.head_folds += k if fold else 0
        tr.tail.done = False                          # every step of the pass carried its own update
        self._grads_to_params()
        tr.loss = losses[k - 1]
        return k, losses

    def _tail(self, lib, shape):
        tail, o = self.trainer.tail, self.trainer.opt
        t = self.tail_struct
        if t is None:
            t = TrainTail()
            c = self.trainer.model.bn.num_features
            p, nc = self.trainer.model.fc1.out_features, self.trainer.model.fc2.out_features
            cols = tail.head_columns(c, p, nc)
            t.n, t.off_adj, t.off_feat, t.off_red, t.rest_off = tail.n, tail.off_adj, tail.off_feat, tail.off_red, \
                tail.rest_off
            t.n_rest = len(cols)
            for i, (col, size) in enumerate(zip(cols, tail.rest_sizes)):
                t.rest_columns[i], t.rest_sizes[i] = col, size
            self.tail_struct = t
        upd = tail.update
        t.param = o.flat_param.data_ptr() if upd else None
        t.exp_avg = o.exp_avg.data_ptr() if upd else None
        t.exp_avg_sq = o.exp_avg_sq.data_ptr() if upd else None
        t.grad_out = o.bucket.flat.data_ptr()
        t.lr = o.lr.data_ptr() if upd else None
        t.beta1, t.beta2, t.eps, t.weight_decay = o.betas[0], o.betas[1], o.eps, o.weight_decay
        t.step_counters = tail.counters.data_ptr() if upd else None
        t.corr_slot = _lib.ptr(tail.corr_slot)
        tail.before_launch(shape)
        return t

    def run(self, entry, model, st, feats, y):
        """Queue the step; returns the loss tensor (complete once the step has run)."""
        from .head import check_labels
        m, plan, _, _ = entry
        check_labels(y, m.nc)
        m.features = feats.data_ptr()
        return self._launch(_lib.load(), model, m, plan, y.data_ptr(), st.n_graphs, feats.device)

    def _launch(self, lib, model, m, plan, labels_ptr, n_graphs, dev):
        tr = self.trainer
        if m.drop_p > 0.0 and not model._rng_seeded:
            model._seed_from_torch()
        with _lib.device_guard(dev):
            fws = self._room(0, plan.forward_bytes, torch.uint8, dev)
            bws = self._room(1, plan.backward_bytes, torch.uint8, dev)
            grads = self._room(2, plan.grad_floats, torch.float32, dev)
            out = self._room(3, n_graphs * m.nc, torch.float32, dev)
            loss = torch.empty((), dtype=torch.float32, device=dev)
            self._point_at_model(m, model)
            fold = plan.train_step == 1
            t = self._tail(lib, "head" if fold else "plain")
            _lib.check(lib.rwk_model_train_step(ctypes.byref(m), ctypes.byref(plan), ctypes.byref(t), fws.data_ptr(),
                                                bws.data_ptr(), labels_ptr, tr._one.data_ptr(), out.data_ptr(),
                                                loss.data_ptr(), grads.data_ptr(), _lib.current_stream(dev)),
                       "rwk_model_train_step")
        tr.tail.after_launch(fold)
        self._grads_to_params()
        return loss
