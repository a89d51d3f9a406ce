"""CPU restatement (numpy, float64 by default) of the random-walk-kernel layer of RW_NN.

TEST INFRASTRUCTURE ONLY.  Nothing under ``gnn_amd/`` may import this module: it is
the checker for the HIP kernels (``tests/``, ``__graft_entry__.smoke`` and the
``cpu_baseline`` leg of ``bench.py`` are its only callers), never the thing shipped or
measured.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function here against
golden vectors that ``oracle/make_golden.py`` produced by importing and running the
reference's own ``RW_NN`` (``rwgnn2/model.py``) in the build container.

What is restated (citations are to ``/root/reference/rwgnn2/model.py``, identical to
``rwgnn/model.py``):

* ``segment_counts``      -- ``torch.unique(graph_indicator, return_counts=True)``  (:32-37)
* ``hidden_adjacency``    -- relu + triu scatter + symmetrise                        (:39-42)
* ``coo_to_csr``          -- the CSR that ``torch.spmm`` multiplies by (coalesced COO) (:55)
* ``rw_kernel_forward``   -- lines :44-67 (zx, per-step propagation, product, segment sum,
                             sum over hidden nodes, normalisation, concatenation)
* ``rw_kernel_backward``  -- the analytic vector-Jacobian product of the same lines
* ``rw_kernel_gram``      -- an independent closed form, valid without dropout
* ``philox4x32`` / ``dropout_keep``  -- the counter-based dropout mask the HIP kernels
                             generate, so a masked forward/backward can be replayed here.
"""
from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------------------
# integer structure (must be reproduced bit-exactly by the device code)
# --------------------------------------------------------------------------------------

def segment_counts(graph_indicator: np.ndarray):
    """model.py:32-33 -- number of graphs and nodes per graph.

    The reference takes ``n_graphs = len(unique)`` and later indexes a length-``n_graphs``
    axis with the raw ids (:60), so ids must be exactly 0..G-1 with no gap; anything
    else makes ``index_add_`` raise.  The same rule is enforced here.
    """
    gi = np.asarray(graph_indicator, dtype=np.int64)
    if gi.size == 0:
        raise ValueError("graph_indicator is empty")
    uniq, counts = np.unique(gi, return_counts=True)
    g = int(uniq.size)
    if uniq[0] != 0 or uniq[-1] != g - 1:
        raise IndexError("graph_indicator ids must cover 0..G-1 without gaps")
    return g, counts.astype(np.int64)


def coo_to_csr(indices: np.ndarray, values: np.ndarray, n: int):
    """Row-major sorted, duplicate-summed CSR of a COO matrix (what ``torch.spmm`` sees after
    coalescing, model.py:55).  Returns (crow int64 (n+1), col int64 (nnz), val float32).

    Duplicates are summed in their original order of appearance (stable sort), in float32,
    which is what ``Tensor.coalesce()`` does on CPU.
    """
    row = np.asarray(indices[0], dtype=np.int64)
    col = np.asarray(indices[1], dtype=np.int64)
    val = np.asarray(values, dtype=np.float32)
    key = row * n + col
    order = np.argsort(key, kind="stable")
    key_s, val_s = key[order], val[order]
    if key_s.size == 0:
        return np.zeros(n + 1, np.int64), np.zeros(0, np.int64), np.zeros(0, np.float32)
    first = np.ones(key_s.size, dtype=bool)
    first[1:] = key_s[1:] != key_s[:-1]
    uniq_keys = key_s[first]
    out_val = np.zeros(uniq_keys.size, dtype=np.float32)
    grp = np.cumsum(first) - 1
    # sequential float32 accumulation in stable order
    for g, v in zip(grp, val_s):
        out_val[g] = np.float32(out_val[g] + v)
    r = uniq_keys // n
    c = uniq_keys % n
    crow = np.zeros(n + 1, dtype=np.int64)
    np.add.at(crow, r + 1, 1)
    crow = np.cumsum(crow)
    return crow, c.astype(np.int64), out_val


def csr_transpose(crow, col, val, n: int):
    """CSR of A^T with each transposed row ordered by ascending original row."""
    crow = np.asarray(crow, np.int64)
    col = np.asarray(col, np.int64)
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(crow))
    order = np.argsort(col * n + rows, kind="stable")
    t_rows = col[order]
    t_cols = rows[order]
    t_val = np.asarray(val)[order]
    t_crow = np.zeros(n + 1, np.int64)
    np.add.at(t_crow, t_rows + 1, 1)
    return np.cumsum(t_crow), t_cols, t_val


def csr_matmul(crow, col, val, x):
    """y = A @ x for CSR A, accumulating each row in CSR order."""
    n = crow.size - 1
    y = np.zeros((n, x.shape[1]), dtype=x.dtype)
    rows = np.repeat(np.arange(n), np.diff(crow))
    np.add.at(y, rows, val[:, None].astype(x.dtype) * x[col])
    return y


# --------------------------------------------------------------------------------------
# hidden graphs
# --------------------------------------------------------------------------------------

def triu_pairs(s: int):
    """Row-major strict upper-triangle index pairs, the order of ``torch.triu_indices(S,S,1)``
    used at model.py:40."""
    r, c = np.triu_indices(s, 1)
    return r, c


def hidden_adjacency(adj_hidden: np.ndarray, s: int) -> np.ndarray:
    """model.py:39-42.  (H, S(S-1)/2) -> symmetric (H, S, S) with zero diagonal."""
    h = adj_hidden.shape[0]
    ah = np.zeros((h, s, s), dtype=adj_hidden.dtype)
    r, c = triu_pairs(s)
    w = np.maximum(adj_hidden, 0)
    ah[:, r, c] = w
    ah[:, c, r] = w
    return ah


# --------------------------------------------------------------------------------------
# the random-walk kernel layer
# --------------------------------------------------------------------------------------

def _propagate(csr, x0, z0, ah, max_step):
    crow, col, val = csr
    xs = [x0]
    zs = [z0]
    for _ in range(1, max_step):
        xs.append(csr_matmul(crow, col, val, xs[-1]))               # model.py:55
        zs.append(np.einsum("abc,acd->abd", ah, zs[-1]))             # model.py:56
    return xs, zs


def rw_kernel_forward(csr, x0, adj_hidden, features_hidden, graph_indicator, max_step,
                      normalize=False, keep=None, drop_p=0.0, dtype=np.float64):
    """model.py:44-67.  Returns K of shape (G, max_step*H); column ``i*H + h``.

    ``keep``: optional boolean array (max_step, H, S, N); the product of step i is multiplied
    by ``keep[i] / (1 - drop_p)`` exactly where the reference applies ``dropout`` (:58).
    """
    x0 = np.asarray(x0, dtype)
    z0 = np.asarray(features_hidden, dtype)
    h, s, _ = z0.shape
    ah = hidden_adjacency(np.asarray(adj_hidden, dtype), s)
    g, counts = segment_counts(graph_indicator)
    gi = np.asarray(graph_indicator, np.int64)
    xs, zs = _propagate(csr, x0, z0, ah, max_step)
    zx = np.einsum("abc,dc->abd", z0, x0)                            # :45
    scale = 1.0 / (1.0 - drop_p) if keep is not None else 1.0
    out = []
    for i in range(max_step):
        t = np.einsum("abc,dc->abd", zs[i], xs[i])                   # :53 / :57
        if keep is not None:
            t = t * (keep[i].astype(dtype) * scale)                  # :58
        t = zx * t                                                   # :59
        seg = np.zeros((h, s, g), dtype)
        np.add.at(seg, (slice(None), slice(None), gi), t)            # :60
        k = seg.sum(axis=1).T                                        # :61-62
        if normalize:
            k = k / counts[:, None].astype(dtype)                    # :63-64
        out.append(k)
    return np.concatenate(out, axis=1)                               # :67


def rw_kernel_backward(csr, x0, adj_hidden, features_hidden, graph_indicator, max_step,
                       d_k, normalize=False, keep=None, drop_p=0.0, dtype=np.float64):
    """Vector-Jacobian product of ``rw_kernel_forward`` for a cotangent ``d_k`` (G, L*H).

    Returns (d_x0 (N,D), d_adj_hidden (H, S(S-1)/2), d_features_hidden (H,S,D)).
    """
    x0 = np.asarray(x0, dtype)
    z0 = np.asarray(features_hidden, dtype)
    adj_hidden = np.asarray(adj_hidden, dtype)
    h, s, d = z0.shape
    ah = hidden_adjacency(adj_hidden, s)
    g, counts = segment_counts(graph_indicator)
    gi = np.asarray(graph_indicator, np.int64)
    crow, col, val = csr
    xs, zs = _propagate(csr, x0, z0, ah, max_step)
    zx = np.einsum("abc,dc->abd", z0, x0)
    scale = 1.0 / (1.0 - drop_p) if keep is not None else 1.0
    d_k = np.asarray(d_k, dtype).reshape(g, max_step, h)
    if normalize:
        d_k = d_k / counts[:, None, None].astype(dtype)
    d_zx = np.zeros_like(zx)
    d_xs = [np.zeros_like(x0) for _ in range(max_step)]
    d_zs = [np.zeros_like(z0) for _ in range(max_step)]
    for i in range(max_step):
        t = np.einsum("abc,dc->abd", zs[i], xs[i])
        w = d_k[gi, i, :].T[:, None, :] * np.ones((1, s, 1), dtype)   # (H,S,N)
        if keep is not None:
            w = w * (keep[i].astype(dtype) * scale)
        d_zx += w * t
        d_t = w * zx
        d_zs[i] += np.einsum("abn,nd->abd", d_t, xs[i])
        d_xs[i] += np.einsum("abn,abd->nd", d_t, zs[i])
    d_zs[0] += np.einsum("abn,nd->abd", d_zx, x0)
    d_xs[0] += np.einsum("abn,abd->nd", d_zx, z0)
    d_ah = np.zeros_like(ah)
    t_crow, t_col, t_val = csr_transpose(crow, col, val, x0.shape[0])
    for i in range(max_step - 1, 0, -1):
        d_xs[i - 1] += csr_matmul(t_crow, t_col, t_val, d_xs[i])
        d_ah += np.einsum("abd,acd->abc", d_zs[i], zs[i - 1])
        d_zs[i - 1] += np.einsum("acb,acd->abd", ah, d_zs[i])
    r, c = triu_pairs(s)
    d_adj_hidden = (d_ah[:, r, c] + d_ah[:, c, r]) * (adj_hidden > 0)
    return d_xs[0], d_adj_hidden, d_zs[0]


def rw_kernel_gram(csr, x0, adj_hidden, features_hidden, graph_indicator, max_step,
                   normalize=False, dtype=np.float64):
    """Independent closed form valid without dropout: per-graph D x D Gram matrices
    C[g,i] = sum_{n in g} x0[n] (x) x_i[n];  K[g, iH+h] = sum_s z0[h,s]^T C[g,i] z_i[h,s]."""
    x0 = np.asarray(x0, dtype)
    z0 = np.asarray(features_hidden, dtype)
    h, s, d = z0.shape
    ah = hidden_adjacency(np.asarray(adj_hidden, dtype), s)
    g, counts = segment_counts(graph_indicator)
    gi = np.asarray(graph_indicator, np.int64)
    xs, zs = _propagate(csr, x0, z0, ah, max_step)
    out = np.zeros((g, max_step * h), dtype)
    for i in range(max_step):
        c = np.zeros((g, d, d), dtype)
        if d <= 8:
            np.add.at(c, gi, x0[:, :, None] * xs[i][:, None, :])
        else:                                   # one small matrix product per graph (no N x D x D temporary)
            order = np.argsort(gi, kind="stable")
            ends = np.cumsum(counts)
            a0, ai = x0[order], xs[i][order]
            for gg in range(g):
                lo = ends[gg] - counts[gg]
                c[gg] = a0[lo:ends[gg]].T @ ai[lo:ends[gg]]
        # two steps so that the D x D contraction is one matrix product: m[g,a,(h,s)] = sum_b c[g,a,b] z_i[h,s,b]
        m = (c.reshape(g * d, d) @ zs[i].reshape(h * s, d).T).reshape(g, d, h, s)
        k = np.einsum("hsa,gahs->gh", z0, m)
        if normalize:
            k = k / counts[:, None].astype(dtype)
        out[:, i * h:(i + 1) * h] = k
    return out


# --------------------------------------------------------------------------------------
# counter-based dropout mask (restates gnn_amd/csrc/rwk_common.h so masks can be replayed)
# --------------------------------------------------------------------------------------

_PHILOX_M0 = np.uint64(0xD2511F53)
_PHILOX_M1 = np.uint64(0xCD9E8D57)
_PHILOX_W0 = np.uint32(0x9E3779B9)
_PHILOX_W1 = np.uint32(0xBB67AE85)


#: rounds of the random-walk layer's mask generator (gnn_amd/csrc/rwk_common.h: RWK_MASK_ROUNDS)
MASK_ROUNDS = 7


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32 with the paper's default 10 rounds."""
    return philox4x32(c0, c1, c2, c3, k0, k1, 10)


def philox4x32(c0, c1, c2, c3, k0, k1, rounds):
    """Philox4x32 with ``rounds`` rounds (Salmon et al., SC'11); all arguments uint32 arrays
    (broadcastable).  Returns four uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(v, np.uint32) for v in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    mask32 = np.uint64(0xFFFFFFFF)
    with np.errstate(over="ignore"):
        for _ in range(rounds):
            p0 = _PHILOX_M0 * c0.astype(np.uint64)
            p1 = _PHILOX_M1 * c2.astype(np.uint64)
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = (p0 & mask32).astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = (p1 & mask32).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32(k0 + _PHILOX_W0)
            k1 = np.uint32(k1 + _PHILOX_W1)
    return c0, c1, c2, c3


def dropout_thresholds(drop_p: float):
    """(T0, T1): keep iff byte < T0, or byte == T0 and the call's tie byte < T1."""
    q = min(1.0, max(0.0, 1.0 - float(drop_p)))
    a = q * 256.0
    t0 = min(256, int(a))
    t1 = min(255, int((a - t0) * 256.0))
    return t0, t1


def dropout_keep(seed: int, offset: int, n_nodes: int, h: int, s: int, max_step: int, drop_p: float):
    """Boolean keep-mask (max_step, H, S, N) exactly as the HIP kernels draw it
    (specification in gnn_amd/csrc/rwk_common.h).

    One Philox4x32-7 stream (``MASK_ROUNDS``) per (node n, hidden graph h): call c has counter
    (n, h, c, offset_lo) and key (seed_lo, seed_hi ^ offset_hi).  Element (step i, hidden node s)
    is sample e = s*L + i: byte (e % 15) of call (e // 15), byte j being bits [8(j&3), 8(j&3)+8) of
    output word j>>2.  Byte 15 of each call breaks ties.
    """
    t0, t1 = dropout_thresholds(drop_p)
    n_samples = s * max_step
    n_calls = (n_samples + 14) // 15
    nn = np.arange(n_nodes, dtype=np.uint32)[None, :, None]
    hh = np.arange(h, dtype=np.uint32)[:, None, None]
    cc = np.arange(n_calls, dtype=np.uint32)[None, None, :]
    zero = np.uint32(0) * (nn + hh + cc)
    off_lo = np.uint32(offset & 0xFFFFFFFF)
    off_hi = np.uint32((offset >> 32) & 0xFFFFFFFF)
    k0 = np.uint32(seed & 0xFFFFFFFF)
    k1 = np.uint32((seed >> 32) & 0xFFFFFFFF) ^ off_hi
    w = philox4x32(nn + zero, hh + zero, cc + zero, off_lo + zero, k0, k1, MASK_ROUNDS)
    words = np.stack(w, axis=-1)                                   # (H, N, calls, 4)
    shifts = np.array([0, 8, 16, 24], dtype=np.uint32)
    by = ((words[..., :, None] >> shifts) & np.uint32(0xFF)).reshape(h, n_nodes, n_calls, 16)
    tie = by[..., 15:16] < t1
    keep = (by[..., :15] < t0) | ((by[..., :15] == t0) & tie)       # (H, N, calls, 15)
    keep = keep.reshape(h, n_nodes, n_calls * 15)[:, :, :n_samples].reshape(h, n_nodes, s, max_step)
    return np.ascontiguousarray(keep.transpose(3, 0, 2, 1))         # (L, H, S, N)
