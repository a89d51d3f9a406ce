"""Row-normalised random-walk propagation layer ``RWConv`` (``rwgnn/rwgnn_pyg.py:19-46``) on the SpMM engine.

The reference builds it on torch_geometric's ``MessagePassing`` (``aggr='add'``): self-loops are appended to
``edge_index`` for every node (``add_self_loops``, existing ones stay), ``deg`` counts how often a node is the
SOURCE of an edge (``degree(row)``, ``:30-31``), every edge ``j -> i`` carries ``x[j] / deg[j]`` (``:34,42``),
messages are summed at the target, and a ``Linear`` follows (``:46``).  In matrix form ``out = lin(M x)`` with
``M[i, j] = (number of edges j -> i, self-loop included) / deg[j]``.  Here ``M`` is built once per ``edge_index``
tensor (cached by identity) as a CSR with those values, and ``M x`` / ``M^T dy`` are launches of the same SpMM
kernels the random-walk-kernel layer uses.  Only the propagation operator of that model family is covered; its
per-hidden-node output layout (``:128-150``) is not.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib
from .function import _reduce_rows
from .structure import GraphStructure, IdentityCache

__all__ = ["RWConv", "propagation_structure", "propagate", "narrow_linear"]

_CACHE = IdentityCache(256)


def propagation_structure(edge_index: torch.Tensor, num_nodes: int, batch: torch.Tensor = None) -> GraphStructure:
    """CSR of ``M`` (and of ``M^T``) for ``edge_index`` (2, E) int64 on a HIP device; cached by tensor identity.
    ``batch`` (N,) int64: the graph of every node, for callers that also need the per-graph segments."""
    if edge_index.dim() != 2 or edge_index.shape[0] != 2 or edge_index.dtype != torch.int64:
        raise TypeError("edge_index must be an int64 tensor of shape (2, E)")
    if not edge_index.is_cuda:
        raise RuntimeError("gnn_amd runs on HIP devices only (no CPU fallback)")
    key = (edge_index.data_ptr(), edge_index.shape[1], edge_index._version, int(num_nodes), str(edge_index.device),
           None if batch is None else (batch.data_ptr(), batch._version))
    st = _CACHE.get(key)
    if st is not None:
        return st
    dev = edge_index.device
    loops = torch.arange(num_nodes, dtype=torch.int64, device=dev)
    src = torch.cat([edge_index[0], loops])                      # add_self_loops, rwgnn_pyg.py:27
    dst = torch.cat([edge_index[1], loops])
    deg = torch.bincount(src, minlength=num_nodes).to(torch.float32)        # degree(row), :31
    deg_inv = torch.where(deg > 0, 1.0 / deg, torch.zeros_like(deg))        # :32-33
    # row = target, column = source; duplicates of an edge add up, as the scatter-add of aggr='add' does
    m = torch.sparse_coo_tensor(torch.stack([dst, src]), deg_inv[src], (num_nodes, num_nodes)).coalesce()
    st = GraphStructure.build(m, batch if batch is not None else torch.zeros(num_nodes, dtype=torch.int64, device=dev))
    _CACHE.put(key, st, (edge_index, batch))
    return st


class _Propagate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, st: GraphStructure):
        if not x.is_cuda:
            raise RuntimeError("gnn_amd runs on HIP devices only (no CPU fallback)")
        if x.dim() != 2 or x.shape[0] != st.n:
            raise ValueError(f"x has shape {tuple(x.shape)} for {st.n} nodes")
        if x.shape[1] > 64:
            raise NotImplementedError("the SpMM kernels cover up to 64 channels")
        xc = x.detach().contiguous().float()
        y = torch.empty_like(xc)
        st.spmm(xc, y)
        ctx.st = st
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        dyc = dy.contiguous().float()
        dx = torch.empty_like(dyc)
        ctx.st.spmm(dyc, dx, transpose=True)
        return dx, None


def propagate(x: torch.Tensor, edge_index: torch.Tensor, structure: GraphStructure = None) -> torch.Tensor:
    """``M x``: every node receives ``x[j] / deg[j]`` from each of its in-edges ``j -> i`` and from itself.
    ``structure``: the :func:`propagation_structure` of ``edge_index`` when the caller has built it already."""
    st = structure if structure is not None else propagation_structure(edge_index, x.shape[0])
    return _Propagate.apply(x, st)


class _NarrowLinear(torch.autograd.Function):
    """``x W^T + b`` for at most 16 input and 64 output channels (``rwgnn_pyg.py:46`` at the reference's widths):
    the narrow-input kernels of the input projection without their sigmoid, instead of a GEMM library's
    N x 4 times 4 x 4 product and its N-long weight-gradient reduction."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        lib = _lib.load()
        dev = x.device
        n, f = x.shape
        d = weight.shape[0]
        xc, wc = x.detach().contiguous().float(), weight.detach().contiguous().float()
        y = torch.empty((n, d), dtype=torch.float32, device=dev)
        with _lib.device_guard(dev):
            _lib.check(lib.rwk_linear_forward(xc.data_ptr(), wc.data_ptr(), bias.detach().contiguous().float().data_ptr(),
                                              n, f, d, y.data_ptr(), _lib.current_stream(dev)), "rwk_linear_forward")
        ctx.save_for_backward(xc, wc)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_y):
        lib = _lib.load()
        xc, wc = ctx.saved_tensors
        dev = xc.device
        n, f = xc.shape
        d = wc.shape[0]
        d_yc = d_y.contiguous().float()
        blocks = lib.rwk_project_backward_blocks(n, d)
        partials = torch.empty((blocks, d * f + d), dtype=torch.float32, device=dev)
        d_x = torch.empty_like(xc) if ctx.needs_input_grad[0] else None
        with _lib.device_guard(dev):
            stream = _lib.current_stream(dev)
            _lib.check(lib.rwk_linear_backward(d_yc.data_ptr(), xc.data_ptr(), wc.data_ptr(), n, f, d,
                                               partials.data_ptr(), _lib.ptr(d_x), stream), "rwk_linear_backward")
            wb = _reduce_rows(lib, partials, stream)
        return d_x, wb[:d * f].view(d, f), wb[d * f:]


def narrow_linear(x, weight, bias):
    """``F.linear(x, weight, bias)`` on the narrow-input kernels when they cover the widths, else torch's."""
    if x.is_cuda and x.dim() == 2 and bias is not None and x.shape[1] <= 16 and weight.shape[0] <= 64 \
            and x.shape[0] > 0:
        return _NarrowLinear.apply(x, weight, bias)
    return torch.nn.functional.linear(x, weight, bias)


class RWConv(nn.Module):
    """Same constructor, parameters (``lin.weight``, ``lin.bias``) and ``forward(x, edge_index)`` as the
    reference's ``RWConv``."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.lin = nn.Linear(in_channels, out_channels)

    def forward(self, x, edge_index, structure: GraphStructure = None):
        return narrow_linear(propagate(x, edge_index, structure), self.lin.weight, self.lin.bias)
