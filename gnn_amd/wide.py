"""The random-walk-kernel layer for hyper-parameters beyond the direct range of the fused kernels
(``hidden_graphs`` > 64, ``hidden_dim`` > 64, ``max_step`` > 4), composed from the same kernels.

``rwgnn2/main.py:22-27`` takes any integer for ``--hidden-graphs``, ``--hidden-dim`` and ``--max-step`` and
``rwgnn2/model.py:48`` loops over any number of steps; a drop-in must not refuse them.  The fused kernels hold one
lane per hidden graph (at most 64), register or LDS tiles of at most 64 channels, and at most four steps per launch.
Everything larger is a sum of pieces they can do, because column ``i*H + h`` of K (``model.py:45-67``),

    K[g, i*H+h] = sum_{n in g} sum_s keep[i,h,s,n] (Z_0[h,s] . X_0[n]) (Z_i[h,s] . X_i[n]),

(a) depends on hidden graph h alone - hidden graphs go in chunks of 64; (b) involves two steps only, 0 and i - every
step is a two-"step" launch on the stacks (X_0, X_i), (Z_0, Z_i), whose second column block is the one wanted;
(c) is bilinear in the two dot products - with the channels cut into tiles, ``(sum_A a_A)(sum_B b_B) = sum_{A,B} a_A
b_B``, every pair of tiles is one such launch (all pairs of a (chunk, step) under the same mask).  The propagation
``X_i = A X_{i-1}`` runs on the SpMM kernels in blocks of 64 channels; the hidden-graph chain (a few thousand
multiply-adds) and the glue between the pieces are torch operators under autograd.

This path is general, not fast: (tiles^2) x steps x chunks launches that each spend half their work on a column block
nobody reads.  Dropout masks follow the kernels' specification with one generator stream per (chunk, step); they are
not the masks a single launch over all hidden graphs would draw (no such launch exists), so parity with dropout is
statistical here; without dropout the result is the reference's to rounding (tests/test_gpu_wide.py, 1e-5).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import _lib
from .function import _reduce_rows, choose_stream_len
from .rwconv import _Propagate
from .structure import GraphStructure

__all__ = ["needs_composition", "rw_kernel_layer_composed"]

MAX_HIDDEN_GRAPHS = 64
MAX_CHANNELS = 64
MAX_STEPS = 4


def needs_composition(n_hidden: int, d: int, steps: int) -> bool:
    return n_hidden > MAX_HIDDEN_GRAPHS or d > MAX_CHANNELS or steps > MAX_STEPS


class _Contract2(torch.autograd.Function):
    """(xa, xb (N, D), za, zb (H, S, D)) -> (G, H):  sum_{n in g} sum_s keep (za[h,s] . xa[n]) (zb[h,s] . xb[n]),
    divided by the node count when ``normalize`` and scaled by 1/(1-p) under dropout.  H <= 64, D <= 64."""

    @staticmethod
    def forward(ctx, xa, xb, za, zb, st: GraphStructure, normalize: bool, drop: float, rng):
        lib = _lib.load()
        dev = xa.device
        n, d = xa.shape
        h, s, _ = za.shape
        g = st.n_graphs
        xs = torch.stack([xa.detach(), xb.detach()]).contiguous().float()
        z = torch.stack([za.detach(), zb.detach()]).contiguous().float()
        with _lib.device_guard(dev):
            stream = _lib.current_stream(dev)
            stream_len = choose_stream_len(n, d, h, s, 2)
            k_part = torch.empty(lib.rwk_fused_k_partials_size(n, stream_len, g, h, 2), dtype=torch.float32, device=dev)
            _lib.check(lib.rwk_fused_forward(xs.data_ptr(), z.data_ptr(), st.gid.data_ptr(), _lib.ptr(st.perm), n, d, h,
                                             s, 2, stream_len, drop, _lib.ptr(rng) if drop else None,
                                             k_part.data_ptr(), stream), "rwk_fused_forward")
            k2 = torch.empty((g, 2 * h), dtype=torch.float32, device=dev)
            _lib.check(lib.rwk_fused_reduce(k_part.data_ptr(), st.seg_ptr.data_ptr(), st.counts_f32.data_ptr(), g, h, 2,
                                            stream_len, drop, int(bool(normalize)), k2.data_ptr(), stream),
                       "rwk_fused_reduce")
        ctx.st, ctx.cfg = st, (bool(normalize), drop)
        ctx.save_for_backward(xs, z, rng if drop else None)
        return k2[:, h:].contiguous()

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_k):
        lib = _lib.load()
        xs, z, rng = ctx.saved_tensors
        st = ctx.st
        normalize, drop = ctx.cfg
        _, n, d = xs.shape
        _, h, s, _ = z.shape
        dev, g = xs.device, st.n_graphs
        with _lib.device_guard(dev):
            stream = _lib.current_stream(dev)
            d_k2 = torch.cat([torch.zeros_like(d_k), d_k.float()], dim=1).contiguous()   # the unread block: no gradient
            stream_len = choose_stream_len(n, d, h, s, 2, backward=True)
            n_part = lib.rwk_fused_dz_partials_count(n, stream_len, d, h, s, 2)
            if n_part <= 0:
                raise NotImplementedError("unsupported hyper-parameters for the fused backward kernel")
            dx = torch.empty_like(xs)
            dz_part = torch.empty((n_part, 2 * h * s * d), dtype=torch.float32, device=dev)
            ws_n = lib.rwk_fused_backward_workspace(n, d, h, s, 2)
            ws = torch.empty(ws_n, dtype=torch.float32, device=dev) if ws_n else None
            _lib.check(lib.rwk_fused_backward(xs.data_ptr(), z.data_ptr(), st.gid.data_ptr(), _lib.ptr(st.perm),
                                              d_k2.data_ptr(), st.counts_f32.data_ptr(), n, d, h, s, 2, stream_len, drop,
                                              int(normalize), _lib.ptr(rng), dx.data_ptr(), dz_part.data_ptr(),
                                              _lib.ptr(ws), stream), "rwk_fused_backward")
            dz = _reduce_rows(lib, dz_part, stream).view(2, h, s, d)
        return dx[0], dx[1], dz[0], dz[1], None, None, None, None


def _propagate(x, st: GraphStructure):
    """A x on the SpMM kernels, 64 channels at a time."""
    if x.shape[1] <= MAX_CHANNELS:
        return _Propagate.apply(x, st)
    return torch.cat([_Propagate.apply(x[:, a:a + MAX_CHANNELS].contiguous(), st)
                      for a in range(0, x.shape[1], MAX_CHANNELS)], dim=1)


def _tiles(t, width, n_tiles):
    """The last axis cut into ``n_tiles`` tiles of ``width`` channels (zero-padded: padding adds nothing to a dot
    product)."""
    pad = width * n_tiles - t.shape[-1]
    if pad:
        t = F.pad(t, (0, pad))
    return [t[..., a * width:(a + 1) * width].contiguous() for a in range(n_tiles)]


def rw_kernel_layer_composed(pre, adj_hidden, features_hidden, st: GraphStructure, max_step: int, normalize: bool,
                             drop_p: float = 0.0, rng_state=None):
    """(G, max_step*H) random-walk-kernel features for any hidden_graphs, hidden_dim, max_step; ``pre`` is
    ``fc(features)`` before the sigmoid."""
    if not pre.is_cuda:
        raise RuntimeError("gnn_amd runs on HIP devices only (no CPU fallback)")
    n_hidden, s, d = features_hidden.shape
    steps = int(max_step)
    if pre.shape != (st.n, d):
        raise ValueError(f"input has shape {tuple(pre.shape)} for {st.n} nodes and hidden_dim {d}")
    drop = float(drop_p) if (drop_p and rng_state is not None) else 0.0
    dev = pre.device
    # model.py:39-44,56: the hidden graphs' chain (tiny: torch operators, autograd)
    n_pairs = s * (s - 1) // 2
    zs = [features_hidden]
    if steps > 1:
        iu = torch.triu_indices(s, s, 1, device=dev)
        ah = torch.zeros(n_hidden, s, s, dtype=features_hidden.dtype, device=dev)
        if n_pairs:
            ah[:, iu[0], iu[1]] = torch.relu(adj_hidden)
        ah = ah + ah.transpose(1, 2)
        for _ in range(1, steps):
            zs.append(torch.einsum("abc,acd->abd", ah, zs[-1]))
    # model.py:43,55: node features of every step
    xs = [torch.sigmoid(pre)]
    for _ in range(1, steps):
        xs.append(_propagate(xs[-1], st))
    n_tiles = -(-d // MAX_CHANNELS)
    width = -(-d // n_tiles)
    width = (width + 3) // 4 * 4 if d > 4 else width
    x_t = [_tiles(x, width, n_tiles) for x in xs]
    rng_used = None
    if drop:
        rng_used = rng_state.clone()
        with torch.no_grad():
            rng_state[1] += 1                                    # the next call draws other masks
    cols = []
    for i in range(steps):
        for c0 in range(0, n_hidden, MAX_HIDDEN_GRAPHS):
            c1 = min(c0 + MAX_HIDDEN_GRAPHS, n_hidden)
            z0_t = _tiles(zs[0][c0:c1], width, n_tiles)
            zi_t = _tiles(zs[i][c0:c1], width, n_tiles)
            rng = None
            if drop:                                             # one generator stream per (chunk, step)
                rng = rng_used.clone()
                rng[1] += (i * ((n_hidden + MAX_HIDDEN_GRAPHS - 1) // MAX_HIDDEN_GRAPHS) + c0 // MAX_HIDDEN_GRAPHS
                           + 1) << 40
            acc = None
            for a in range(n_tiles):
                for b in range(n_tiles):
                    part = _Contract2.apply(x_t[0][a], x_t[i][b], z0_t[a], zi_t[b], st, normalize, drop, rng)
                    acc = part if acc is None else acc + part
            cols.append(acc)
    return torch.cat(cols, dim=1)                                # column i*H + h (model.py:67)
