"""``RW_NN.forward`` in evaluation mode as one launch (``csrc/rwk_eval.hip``).

The reference's validation and test loops (``rwgnn2/main.py:110-117,148-151``) call ``model(adj, features,
graph_indicator)`` under ``model.eval()`` every epoch - without ``torch.no_grad()``, so an autograd graph is recorded
that nobody walks.  Here that call is one kernel launch (one workgroup per graph: projection, propagation, hidden
graphs, Gram matrices, K row and head in LDS).  The result still carries a ``grad_fn``: should anyone call
``backward`` on it, the gradients are computed by running the ordinary launches again with the parameters as they are
then (the usual recomputation caveat: parameters changed in between are seen changed).
"""
from __future__ import annotations

import torch

from . import _lib

__all__ = ["eval_forward_supported", "eval_forward", "eval_forward_planned"]

#: False keeps the separate launches in evaluation mode (tests compare the two)
FUSED_EVAL_FORWARD = True
#: largest graph the one-workgroup-per-graph forward takes
MAX_GRAPH_NODES = 2048


def _facts(st):
    """(largest graph, block-diagonal?) of a structure, read from the device once and kept on it."""
    facts = getattr(st, "eval_facts", None)
    if facts is None:
        crow, col, gid = st.csr.crow, st.csr.col, st.gid
        if st.nnz:
            rows = torch.repeat_interleave(torch.arange(st.n, device=crow.device), (crow[1:] - crow[:-1]).long(),
                                           output_size=st.nnz)
            cross = (gid[rows] != gid[col.long()]).sum()
        else:
            cross = torch.zeros((), dtype=torch.int64, device=crow.device)
        both = torch.stack([st.counts.max(), cross.to(torch.int64)]).cpu().tolist()
        facts = (int(both[0]), both[1] == 0)
        st.eval_facts = facts
    return facts


def eval_forward_supported(model, st, features) -> bool:
    """Shapes the one-launch evaluation forward serves: narrow input or any input through ``fc``, hidden_dim <= 4,
    sorted graph ids, a block-diagonal matrix, graphs that fit the LDS with the hidden graphs' rows."""
    if not FUSED_EVAL_FORWARD or st.perm is not None or not features.is_cuda:
        return False
    d = model.features_hidden.shape[2]
    if d > 4 or model.max_step > 4 or features.dtype != torch.float32:
        return False
    bn = model.bn
    if not (model.fused_head and bn.affine and bn.track_running_stats and model.fc1.bias is not None
            and model.fc2.bias is not None and bn.num_features == model.hidden_graphs * model.max_step):
        return False
    max_nodes, block_diagonal = _facts(st)
    # one workgroup walks its graph alone: graphs of thousands of nodes, or rows of hundreds of entries (one thread
    # per row), are better served by the launches that spread a graph over the chip
    if not block_diagonal or max_nodes > MAX_GRAPH_NODES or st.csr.max_degree > 64:
        return False
    f = features.shape[1] if features.shape[1] <= 16 else 1
    return bool(_lib.load().rwk_eval_forward_lds(d, model.hidden_graphs, model.size_hidden_graphs, model.max_step, f,
                                                 max_nodes, model.fc1.out_features, model.fc2.out_features))


class _EvalForward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, st, features, *params):
        lib = _lib.load()
        dev = features.device
        h, s, d = model.features_hidden.shape
        g = st.n_graphs
        narrow = features.shape[1] <= 16
        with _lib.device_guard(dev), torch.no_grad():
            inp = (features if narrow else model.fc(features)).detach().contiguous()
            out = torch.empty((g, model.fc2.out_features), dtype=torch.float32, device=dev)
            t = [x.detach().contiguous() for x in (model.fc.weight, model.fc.bias, model.adj_hidden,
                                                   model.features_hidden, model.bn.weight, model.bn.bias,
                                                   model.fc1.weight, model.fc1.bias, model.fc2.weight, model.fc2.bias)]
            fc_w, fc_b, adj_h, feat_h, bn_w, bn_b, w1, b1, w2, b2 = t
            csr = st.csr
            _lib.check(lib.rwk_eval_forward(
                inp.data_ptr(), inp.shape[1], fc_w.data_ptr() if narrow else None, fc_b.data_ptr() if narrow else None,
                _lib.ptr(adj_h) if adj_h.numel() else None, feat_h.data_ptr(), csr.crow.data_ptr(), _lib.ptr(csr.col),
                None if csr.unit_values else _lib.ptr(csr.val), st.seg_ptr.data_ptr(), st.counts_f32.data_ptr(), st.n,
                g, _facts(st)[0], d, h, s, model.max_step, int(bool(model.normalize)), bn_w.data_ptr(), bn_b.data_ptr(),
                model.bn.running_mean.data_ptr(), model.bn.running_var.data_ptr(), float(model.bn.eps), w1.data_ptr(),
                b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), w1.shape[0], w2.shape[0], None, out.data_ptr(),
                _lib.current_stream(dev)), "rwk_eval_forward")
        ctx.model, ctx.st = model, st
        ctx.save_for_backward(features)
        return out

    @staticmethod
    def backward(ctx, d_out):
        # Nobody in the reference walks this graph (main.py:110-117 never calls backward on a validation output); if
        # somebody does, the ordinary launches run the forward again with autograd on and hand back their gradients.
        (features,) = ctx.saved_tensors
        model = ctx.model
        params = list(model.parameters())
        with torch.enable_grad():
            feats = features.detach().requires_grad_(ctx.needs_input_grad[2])
            out = model._forward_launches(ctx.st, feats)
            wanted = ([feats] if feats.requires_grad else []) + [p for p in params if p.requires_grad]
            grads = list(torch.autograd.grad(out, wanted, d_out, allow_unused=True))
        d_feats = grads.pop(0) if feats.requires_grad else None
        by_param = iter(grads)
        return (None, None, d_feats) + tuple(next(by_param) if p.requires_grad else None for p in params)


def eval_forward(model, st, features):
    return _EvalForward.apply(model, st, features, *model.parameters())


def eval_forward_planned(model, dataset, plan, i: int, out: torch.Tensor = None):
    """``model.eval()``'s output for batch ``i`` of ``plan`` (``DeviceDataset.plan``: lists of graph ids of a
    device-resident data set) WITHOUT assembling the batch: one launch whose workgroups read their graphs straight
    from the data set (``rwk_eval_forward_gather``) - the validation / test loop (``rwgnn2/main.py:110-117,148-151``)
    for data sets that do not stay resident as batches.  Returns the (graphs, n_classes) log-probabilities (into
    ``out`` when given), or None when the launch does not serve the model or the batch (a weighted data set, wide
    features, hidden_dim > 4, a graph too large for the LDS, rows of more than 64 entries): assemble the batch
    (``dataset.batch_planned``) and call the model then.  No autograd graph is recorded."""
    if not FUSED_EVAL_FORWARD or model.training or not dataset.unit_values or not 1 <= dataset.n_feat <= 16:
        return None
    h, s, d = model.features_hidden.shape
    bn = model.bn
    if d > 4 or model.max_step > 4 or model.fc.in_features != dataset.n_feat or not (
            model.fused_head and bn.affine and bn.track_running_stats and model.fc1.bias is not None
            and model.fc2.bias is not None and bn.num_features == model.hidden_graphs * model.max_step):
        return None
    b, n, nnz, max_degree, _, off, _, _, max_graph = plan.metas[i]
    if max_graph > MAX_GRAPH_NODES or max_degree > 64:
        return None
    lib = _lib.load()
    p, nc = model.fc1.out_features, model.fc2.out_features
    if not lib.rwk_eval_forward_lds(d, h, s, model.max_step, dataset.n_feat, max_graph, p, nc):
        return None
    dev = dataset.device
    with _lib.device_guard(dev), torch.no_grad():
        if out is None:
            out = torch.empty((b, nc), dtype=torch.float32, device=dev)
        elif out.shape != (b, nc) or out.dtype is not torch.float32 or not out.is_contiguous():
            raise ValueError("out must be a contiguous float32 tensor of shape (graphs, n_classes)")
        t = [x.detach().contiguous() for x in (model.fc.weight, model.fc.bias, model.adj_hidden, model.features_hidden,
                                               bn.weight, bn.bias, model.fc1.weight, model.fc1.bias, model.fc2.weight,
                                               model.fc2.bias)]
        fc_w, fc_b, adj_h, feat_h, bn_w, bn_b, w1, b1, w2, b2 = t
        _lib.check(lib.rwk_eval_forward_gather(
            plan.tables.data_ptr() + 8 * off, dataset.row_ptr.data_ptr(), _lib.ptr(dataset.col),
            dataset.features.data_ptr(), dataset.n_feat, fc_w.data_ptr(), fc_b.data_ptr(),
            _lib.ptr(adj_h) if adj_h.numel() else None, feat_h.data_ptr(), b, max_graph, d, h, s, model.max_step,
            int(bool(model.normalize)), bn_w.data_ptr(), bn_b.data_ptr(), bn.running_mean.data_ptr(),
            bn.running_var.data_ptr(), float(bn.eps), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), p, nc,
            None, out.data_ptr(), _lib.current_stream(dev)), "rwk_eval_forward_gather")
    out._rwk_plan_tables = plan.tables          # the launch reads them asynchronously
    return out
