"""Functional PyTorch-CPU restatement of the whole RW_NN forward pass.

TEST INFRASTRUCTURE ONLY (see ``oracle/rwnn_oracle.py``): used by the tests to obtain
autograd gradients of the full model in float64/float32, and by ``bench.py`` as the
``cpu_baseline`` ("port": the same operator sequence the reference issues, timed on the
host cores).  Never imported by ``gnn_amd``.

It follows ``/root/reference/rwgnn2/model.py:31-72`` operator for operator -- sparse-COO
``torch.spmm``, ``einsum``, ``dropout``, ``index_add_``, batch-norm, two linears,
``log_softmax`` -- but as a pure function of a parameter dict, so the same code runs in
any dtype and with any weights.  Parity status: PINNED by ``tests/golden`` (vectors made
by running the reference itself, ``oracle/make_golden.py``).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

PARAM_KEYS = (
    "adj_hidden", "features_hidden", "fc.weight", "fc.bias", "bn.weight", "bn.bias",
    "fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias",
)


def hidden_adjacency(adj_hidden: torch.Tensor, size: int) -> torch.Tensor:
    """model.py:39-42."""
    n_hidden = adj_hidden.shape[0]
    dense = adj_hidden.new_zeros(n_hidden, size, size)
    iu = torch.triu_indices(size, size, 1)
    dense[:, iu[0], iu[1]] = torch.relu(adj_hidden)
    return dense + dense.transpose(1, 2)


def rw_kernel_features(params, adj, features, graph_indicator, max_step, normalize,
                       drop_p=0.0, training=False):
    """model.py:32-67: the (G, max_step*H) random-walk-kernel features that feed batch-norm."""
    z = params["features_hidden"]
    n_hidden = z.shape[0]
    ids, counts = torch.unique(graph_indicator, return_counts=True)
    n_graphs = ids.numel()
    a_hidden = hidden_adjacency(params["adj_hidden"], z.shape[1])
    x = torch.sigmoid(F.linear(features, params["fc.weight"], params["fc.bias"]))
    zx = torch.einsum("hsd,nd->hsn", z, x)
    cols = []
    for step in range(max_step):
        if step > 0:
            x = torch.spmm(adj, x)
            z = torch.einsum("hst,htd->hsd", a_hidden, z)
        t = torch.einsum("hsd,nd->hsn", z, x)
        t = F.dropout(t, drop_p, training)
        t = zx * t
        seg = t.new_zeros(n_hidden, t.shape[1], n_graphs).index_add_(2, graph_indicator, t)
        k = seg.sum(dim=1).t()
        if normalize:
            k = k / counts.unsqueeze(1).to(k.dtype)
        cols.append(k)
    return torch.cat(cols, dim=1)


def head(params, k, drop_p=0.0, training=False, bn_stats=None, eps=1e-5):
    """model.py:68-72.  ``bn_stats`` = (running_mean, running_var) selects eval-mode batch-norm."""
    if training or bn_stats is None:
        out = F.batch_norm(k, None, None, params["bn.weight"], params["bn.bias"], True, 0.0, eps)
    else:
        out = F.batch_norm(k, bn_stats[0], bn_stats[1], params["bn.weight"], params["bn.bias"],
                           False, 0.0, eps)
    out = torch.relu(F.linear(out, params["fc1.weight"], params["fc1.bias"]))
    out = F.dropout(out, drop_p, training)
    out = F.linear(out, params["fc2.weight"], params["fc2.bias"])
    return F.log_softmax(out, dim=1)


def forward(params, adj, features, graph_indicator, max_step, normalize, drop_p=0.0,
            training=False, bn_stats=None):
    k = rw_kernel_features(params, adj, features, graph_indicator, max_step, normalize,
                           drop_p, training)
    return head(params, k, drop_p, training, bn_stats), k


def cast_params(state_dict, dtype, requires_grad=False):
    out = {}
    for key in PARAM_KEYS:
        out[key] = state_dict[key].detach().to(dtype).clone().requires_grad_(requires_grad)
    return out
