"""``RW_GNN`` of ``rwgnn/rwgnn_pyg.py:48-170`` without torch_geometric: row-normalised propagation (``RWConv``) with
a ``Linear`` per step, and random-walk-kernel features kept per hidden NODE (G x H*S per step instead of G x H).

Kept per hidden node, the feature of (hidden graph h, hidden node s) is what the fused kernels compute for a
hidden graph of size one whose only row is ``Z_i[h, s]``.  Two ways to get them: keep-S instances of the fused kernels
(hidden_dim <= 4, 9..16 hidden graphs, L*S <= 20: one launch each way, the lane of a hidden graph keeps its S sums
apart), and, for every other shape, the H*S pairs handed to the ordinary kernels as one-node hidden graphs, up to
sixty-four at a launch.  The hidden-graph chain
``Z_i = Ah Z_{i-1}`` and its backward are the usual kernels on the real H x S graphs; the node features of the steps
(``x_i = lin_i(M x_{i-1})``) are built by autograd from :mod:`gnn_amd.rwconv` and enter the fused kernels as a
stack, whose gradient they return.

Everything the reference does after the per-step features (its transposes and ``reshape`` - which, applied to a
non-contiguous (S, G, H) tensor, interleaves graphs -, lazily created batch-norm and linear layers, ReLU, dropout,
log-softmax, ``:143-170``) is done with the same torch operators on the small (G, H*S*L) matrix.
Parity of this model is unpinned (torch_geometric is not available to run the reference).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .function import _reduce_rows, choose_stream_len
from .head import fused_head, head_supported
from .rwconv import RWConv, narrow_linear, propagation_structure
from .structure import GraphStructure

__all__ = ["RW_GNN", "hidden_chain", "pair_kernel_features"]


class _HiddenChain(torch.autograd.Function):
    """(adj_hidden, features_hidden) -> Z (L, H, S, D), Z_0 = features_hidden, Z_i = Ah Z_{i-1}
    (``rwgnn_pyg.py:105-110,131``)."""

    @staticmethod
    def forward(ctx, adj_hidden, features_hidden, steps: int):
        lib = _lib.load()
        if not features_hidden.is_cuda:
            raise RuntimeError("gnn_amd runs on HIP devices only (no CPU fallback)")
        h, s, d = features_hidden.shape
        dev = features_hidden.device
        adj_c = adj_hidden.detach().contiguous()
        feat_c = features_hidden.detach().contiguous()
        z = torch.empty((steps, h, s, d), dtype=torch.float32, device=dev)
        ah = torch.empty((h, s, s), dtype=torch.float32, device=dev)
        with _lib.device_guard(dev):
            _lib.check(lib.rwk_hidden_forward(_lib.ptr(adj_c), feat_c.data_ptr(), h, s, d, steps, ah.data_ptr(),
                                              z.data_ptr(), None, None, _lib.current_stream(dev)),
                       "rwk_hidden_forward")
        ctx.save_for_backward(adj_c, ah, z)
        ctx.steps = steps
        return z

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_z):
        lib = _lib.load()
        adj_c, ah, z = ctx.saved_tensors
        steps, h, s, d = z.shape
        dev = z.device
        d_zc = d_z.contiguous().float()                       # one "partial" row: the gradient itself
        d_adj = torch.empty_like(adj_c)
        d_feat = torch.empty((h, s, d), dtype=torch.float32, device=dev)
        with _lib.device_guard(dev):
            _lib.check(lib.rwk_hidden_backward(_lib.ptr(adj_c), ah.data_ptr(), z.data_ptr(), d_zc.data_ptr(), 1, h, s,
                                               d, steps, _lib.ptr(d_adj), d_feat.data_ptr(),
                                               _lib.current_stream(dev)), "rwk_hidden_backward")
        return (d_adj if steps > 1 else None), d_feat, None


def hidden_chain(adj_hidden, features_hidden, steps: int):
    return _HiddenChain.apply(adj_hidden, features_hidden, steps)


#: use the keep-S kernel instances where they are built (False: always the one-node-hidden-graph formulation; the
#: tests compare the two)
KEEP_S_KERNELS = True


def _chunk(d: int) -> int:
    """Pairs per launch: the kernels' limit on hidden graphs."""
    return 64


class _PairKernels(torch.autograd.Function):
    """xs (L, N, D), z (L, H, S, D) -> K (G, L, H*S):
    ``K[g, i, h*S + s] = sum_{n in g} keep * (Z_0[h,s] . xs[0][n]) * (Z_i[h,s] . xs[i][n]) / (1 - p)``
    (``rwgnn_pyg.py:115,125,132-141``)."""

    @staticmethod
    def forward(ctx, xs, z, st: GraphStructure, drop_p: float, rng_used):
        lib = _lib.load()
        if not xs.is_cuda:
            raise RuntimeError("gnn_amd runs on HIP devices only (no CPU fallback)")
        steps, n, d = xs.shape
        _, h, s, _ = z.shape
        if n != st.n or z.shape[0] != steps or z.shape[3] != d:
            raise ValueError("xs, z and the batch structure do not belong together")
        if d > 64 or steps > 4:
            raise NotImplementedError("the HIP kernels cover hidden_dim <= 64 and max_step <= 4")
        dev, g, pairs = xs.device, st.n_graphs, h * s
        drop = float(drop_p) if (drop_p and rng_used is not None) else 0.0
        xs_c = xs.detach().contiguous().float()
        z_pairs = z.detach().reshape(steps, pairs, 1, d)
        k = torch.empty((g, steps, pairs), dtype=torch.float32, device=dev)
        chunks = []
        keep_s = bool(lib.rwk_fused_keep_s_supported(d, h, s, steps)) and KEEP_S_KERNELS
        with _lib.device_guard(dev):
            stream = _lib.current_stream(dev)
            if keep_s:                                     # one launch on the real H x S hidden graphs
                zc = z.detach().contiguous().float()
                stream_len = choose_stream_len(n, d, h, s, steps)
                k_part = torch.empty(lib.rwk_fused_k_partials_size(n, stream_len, g, pairs, steps),
                                     dtype=torch.float32, device=dev)
                _lib.check(lib.rwk_fused_forward_keep_s(xs_c.data_ptr(), zc.data_ptr(), st.gid.data_ptr(),
                                                        _lib.ptr(st.perm), n, d, h, s, steps, stream_len, drop,
                                                        _lib.ptr(rng_used) if drop else None, k_part.data_ptr(),
                                                        stream), "rwk_fused_forward_keep_s")
                _lib.check(lib.rwk_fused_reduce(k_part.data_ptr(), st.seg_ptr.data_ptr(), st.counts_f32.data_ptr(), g,
                                                pairs, steps, stream_len, drop, 0, k.data_ptr(), stream),
                           "rwk_fused_reduce")
                chunks.append((0, pairs, zc, rng_used if drop else None))
            for c, a in enumerate(range(0, 0 if keep_s else pairs, _chunk(d))):
                b = min(a + _chunk(d), pairs)
                hq = b - a
                zc = z_pairs[:, a:b].contiguous()
                rng_c = None
                if drop:                                   # one generator stream per chunk of pairs
                    rng_c = rng_used.clone()
                    rng_c[1] += c << 40
                stream_len = choose_stream_len(n, d, hq, 1, steps)
                k_part = torch.empty(lib.rwk_fused_k_partials_size(n, stream_len, g, hq, steps), dtype=torch.float32,
                                     device=dev)
                _lib.check(lib.rwk_fused_forward(xs_c.data_ptr(), zc.data_ptr(), st.gid.data_ptr(), _lib.ptr(st.perm),
                                                 n, d, hq, 1, steps, stream_len, drop, _lib.ptr(rng_c),
                                                 k_part.data_ptr(), stream), "rwk_fused_forward")
                kc = torch.empty((g, steps * hq), dtype=torch.float32, device=dev)
                _lib.check(lib.rwk_fused_reduce(k_part.data_ptr(), st.seg_ptr.data_ptr(), st.counts_f32.data_ptr(), g,
                                                hq, steps, stream_len, drop, 0, kc.data_ptr(), stream),
                           "rwk_fused_reduce")
                k[:, :, a:b] = kc.view(g, steps, hq)
                chunks.append((a, b, zc, rng_c))
        ctx.st, ctx.drop, ctx.chunks, ctx.shape_z, ctx.keep_s = st, drop, chunks, tuple(z.shape), keep_s
        ctx.save_for_backward(xs_c)
        return k

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_k):
        lib = _lib.load()
        (xs_c,) = ctx.saved_tensors
        st, drop = ctx.st, ctx.drop
        steps, n, d = xs_c.shape
        _, h, s, _ = ctx.shape_z
        dev, g = xs_c.device, st.n_graphs
        d_k = d_k.contiguous().float()
        dx = torch.zeros_like(xs_c)
        dz = torch.empty((steps, h * s, 1, d), dtype=torch.float32, device=dev)
        with _lib.device_guard(dev):
            stream = _lib.current_stream(dev)
            if ctx.keep_s:
                _, _, zc, rng_c = ctx.chunks[0]
                stream_len = choose_stream_len(n, d, h, s, steps, backward=True)
                n_part = lib.rwk_fused_dz_partials_count(n, stream_len, d, h, s, steps)
                dz_part = torch.empty((n_part, steps * h * s * d), dtype=torch.float32, device=dev)
                dx = torch.empty_like(xs_c)
                _lib.check(lib.rwk_fused_backward_keep_s(xs_c.data_ptr(), zc.data_ptr(), st.gid.data_ptr(),
                                                         _lib.ptr(st.perm), d_k.data_ptr(), n, d, h, s, steps,
                                                         stream_len, drop, _lib.ptr(rng_c), dx.data_ptr(),
                                                         dz_part.data_ptr(), stream), "rwk_fused_backward_keep_s")
                return dx, _reduce_rows(lib, dz_part, stream).view(steps, h, s, d), None, None, None
            for a, b, zc, rng_c in ctx.chunks:
                hq = b - a
                stream_len = choose_stream_len(n, d, hq, 1, steps, backward=True)
                n_part = lib.rwk_fused_dz_partials_count(n, stream_len, d, hq, 1, steps)
                if n_part <= 0:
                    raise NotImplementedError("unsupported hyper-parameters for the fused backward kernel")
                dkc = d_k[:, :, a:b].reshape(g, steps * hq).contiguous()
                dxc = torch.empty_like(xs_c)
                dz_part = torch.empty((n_part, steps * hq * d), dtype=torch.float32, device=dev)
                ws_n = lib.rwk_fused_backward_workspace(n, d, hq, 1, steps)
                ws = torch.empty(ws_n, dtype=torch.float32, device=dev) if ws_n else None
                _lib.check(lib.rwk_fused_backward(xs_c.data_ptr(), zc.data_ptr(), st.gid.data_ptr(),
                                                  _lib.ptr(st.perm), dkc.data_ptr(), st.counts_f32.data_ptr(), n, d,
                                                  hq, 1, steps, stream_len, drop, 0, _lib.ptr(rng_c), dxc.data_ptr(),
                                                  dz_part.data_ptr(), _lib.ptr(ws), stream), "rwk_fused_backward")
                dx += dxc
                dz[:, a:b] = _reduce_rows(lib, dz_part, stream).view(steps, hq, 1, d)
        return dx, dz.view(steps, h, s, d), None, None, None


def pair_kernel_features(xs, z, st: GraphStructure, drop_p: float = 0.0, rng_used=None):
    return _PairKernels.apply(xs, z, st, drop_p, rng_used)


class _Data:
    """The three attributes of a torch_geometric ``Batch`` the model reads (``rwgnn_pyg.py:95``)."""

    def __init__(self, x, edge_index, batch):
        self.x, self.edge_index, self.batch = x, edge_index, batch


class RW_GNN(nn.Module):
    """Same constructor arguments, parameter names and ``forward(data)`` (``data.x``, ``data.edge_index``,
    ``data.batch``) as ``rwgnn_pyg.py:48-170``; ``bn``, ``fc1``, ``fc2`` are created on the first forward pass, as
    there.  ``RW_GNN.Data(x, edge_index, batch)`` stands in for a torch_geometric batch."""

    Data = _Data

    def __init__(self, input_dim, max_step, hidden_graphs, size_hidden_graphs, hidden_dim, penultimate_dim, normalize,
                 num_classes, dropout, device):
        super().__init__()
        self.max_step = max_step
        self.hidden_graphs = hidden_graphs
        self.size_hidden_graphs = size_hidden_graphs
        self.normalize = normalize
        self.device = device
        self.penultimate_dim = penultimate_dim
        self.num_classes = num_classes
        self.adj_hidden = nn.Parameter(torch.empty(hidden_graphs, (size_hidden_graphs * (size_hidden_graphs - 1)) // 2))
        self.features_hidden = nn.Parameter(torch.empty(hidden_graphs, size_hidden_graphs, hidden_dim))
        self.fc = nn.Linear(input_dim, hidden_dim)
        self.rw_layers = nn.ModuleList([RWConv(hidden_dim, hidden_dim) for _ in range(max_step)])
        self.bn = None
        self.fc1 = None
        self.fc2 = None
        self.dropout = nn.Dropout(p=dropout)
        self.relu = nn.ReLU()
        self.sigmoid = nn.Sigmoid()
        self.init_weights()
        self._rng_state = None
        self.fused_head = True           # False: the head runs on the torch modules' own operators

    def init_weights(self):                                   # rwgnn_pyg.py:90-92
        self.adj_hidden.data.uniform_(-1, 1)
        self.features_hidden.data.uniform_(0, 1)

    def seed_dropout(self, seed: int, offset: int = 0):
        dev = self.adj_hidden.device
        self._rng_state = torch.tensor([int(seed) & 0x7FFFFFFFFFFFFFFF, int(offset)], dtype=torch.int64, device=dev)

    def _draw_rng(self):
        if self._rng_state is None or self._rng_state.device != self.adj_hidden.device:
            self.seed_dropout(int(torch.randint(0, 2 ** 62, (1,)).item()))
        used = self._rng_state.clone()
        self._rng_state[1] += 1
        return used

    def rw_features(self, data):
        """The concatenated per-step features before batch-norm (``rwgnn_pyg.py:94-153``), (G, H*S*max_step)."""
        x, edge_index, batch = data.x, data.edge_index, data.batch
        st = propagation_structure(edge_index, x.shape[0], batch)
        g, h, s = st.n_graphs, self.hidden_graphs, self.size_hidden_graphs
        x = self.sigmoid(narrow_linear(x, self.fc.weight, self.fc.bias))     # :113
        steps = [x]
        for i in range(1, self.max_step):
            steps.append(self.rw_layers[i - 1](steps[-1], edge_index, st))    # :128
        xs = torch.stack(steps)
        z = hidden_chain(self.adj_hidden, self.features_hidden, self.max_step)
        p = self.dropout.p if self.training else 0.0
        k = pair_kernel_features(xs, z, st, p, self._draw_rng() if p > 0 else None)      # (G, L, H*S)
        out = []
        for i in range(self.max_step):
            t = k[:, i].reshape(g, h, s).permute(2, 1, 0)                     # (S, H, G): :144 after :138-141
            if self.normalize:                                                # :145-146, as written there
                norm = st.counts.unsqueeze(1).repeat(1, h)
                t = t / norm.unsqueeze(2)
            t = torch.transpose(t, 1, 2)                                      # :148
            out.append(t.reshape(g, -1))                                      # :149
        return torch.cat(out, dim=1)                                          # :153

    def forward(self, data):
        out = self.rw_features(data)
        dev = out.device
        if self.bn is None or self.bn.num_features != out.size(1):            # :156-164
            self.bn = nn.BatchNorm1d(out.size(1)).to(dev)
        if self.fc1 is None or self.fc1.in_features != out.size(1):
            self.fc1 = nn.Linear(out.size(1), self.penultimate_dim).to(dev)
            self.fc2 = nn.Linear(self.penultimate_dim, self.num_classes).to(dev)
        if self.fused_head and out.is_cuda and isinstance(self.bn.momentum, float) and \
                head_supported(out.size(0), out.size(1), self.penultimate_dim, self.num_classes):
            # :159,166-170 in the fused head kernels (same modules and state_dict; own dropout generator)
            p = self.dropout.p if self.training else 0.0
            if self._rng_state is None or self._rng_state.device != dev:
                self.seed_dropout(int(torch.randint(0, 2 ** 62, (1,)).item()))
            return fused_head(out, self.bn, self.fc1, self.fc2, p, self.training, self._rng_state if p > 0 else None)
        out = self.bn(out)
        out = self.relu(self.fc1(out))
        out = self.dropout(out)
        out = self.fc2(out)
        return F.log_softmax(out, dim=1)
