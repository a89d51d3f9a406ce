"""TEST INFRASTRUCTURE ONLY: float64 restatement of ``RWConv`` (``/root/reference/rwgnn/rwgnn_pyg.py:19-46``).

PARITY UNPINNED: the reference runs on torch_geometric (``MessagePassing``, ``add_self_loops``, ``degree``), which is
not installed here and is not vendored in the reference, and the reference holds no fixtures for this layer.  This
file restates the published semantics of those three pieces (torch_geometric 2.x): ``add_self_loops`` appends
(i, i) for every node and keeps the edges it was given, duplicates included; ``degree(index, N)`` counts
occurrences of each node in ``index``; ``propagate`` with ``aggr='add'`` and the default flow
``source_to_target`` gathers ``x_j = x[edge_index[0]]`` and scatter-adds the messages at ``edge_index[1]``.
"""
import numpy as np


def rwconv_forward(x, edge_index, weight, bias):
    """x (N, C) -> lin(sum over edges j->i (self-loops appended) of x[j] / deg[j]), all in float64."""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[0]
    loops = np.arange(n)
    src = np.concatenate([np.asarray(edge_index[0]), loops])         # rwgnn_pyg.py:27
    dst = np.concatenate([np.asarray(edge_index[1]), loops])
    deg = np.bincount(src, minlength=n).astype(np.float64)           # :31, degree(row)
    with np.errstate(divide="ignore"):
        deg_inv = np.where(deg > 0, 1.0 / deg, 0.0)                  # :32-33
    msg = deg_inv[src][:, None] * x[src]                             # :34, :42  norm * x_j
    agg = np.zeros_like(x)
    np.add.at(agg, dst, msg)                                         # aggr='add' at the target
    return agg @ np.asarray(weight, np.float64).T + np.asarray(bias, np.float64), agg     # :46


def rwconv_backward(x, edge_index, weight, d_out):
    """Gradients of sum(out * d_out): (d_x, d_weight, d_bias)."""
    x = np.asarray(x, dtype=np.float64)
    d_out = np.asarray(d_out, dtype=np.float64)
    w = np.asarray(weight, np.float64)
    n = x.shape[0]
    _, agg = rwconv_forward(x, edge_index, w, np.zeros(w.shape[0]))
    d_agg = d_out @ w
    loops = np.arange(n)
    src = np.concatenate([np.asarray(edge_index[0]), loops])
    dst = np.concatenate([np.asarray(edge_index[1]), loops])
    deg = np.bincount(src, minlength=n).astype(np.float64)
    deg_inv = np.where(deg > 0, 1.0 / np.maximum(deg, 1), 0.0)
    d_x = np.zeros_like(x)
    np.add.at(d_x, src, deg_inv[src][:, None] * d_agg[dst])
    return d_x, d_out.T @ agg, d_out.sum(0)
