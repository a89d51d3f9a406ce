"""TEST INFRASTRUCTURE ONLY - CPU restatement of the distance-biased transition matrix of the reference's
distance-biased walk models, dense, one graph at a time.  Only tests/ may import this file; nothing in gnn_amd/
does.

PARITY UNPINNED: the reference module (rwgnn2/rwgnn_dis_bias6.py) imports torch_geometric, which is not installed
here, and the reference holds no fixture for these matrices; this file follows the reference's text.

  hop_distances            rwgnn2/rwgnn_dis_bias6.py:109-128 (_compute_simple_distance_matrix: 1, 2 and 3 hops from
                           dense powers of the adjacency matrix, everything else at max_distance + 1), continued to
                           any max_distance the same way; equal to scipy's dijkstra(unweighted=True,
                           limit=max_distance) with inf -> max_distance + 1 of :130-154
  walk_probabilities       :222-250 (_compute_walk_probabilities: 'exp' and 'levy' weights, zero diagonal, row
                           normalisation with empty rows left at zero, uniform jump with probability c)
  apply_batch              P x per graph of a block-diagonal batch (what a dense ``walk_probs @ x`` gives)
"""
from __future__ import annotations

import numpy as np


def hop_distances(adj: np.ndarray, max_distance: int = 3) -> np.ndarray:
    n = adj.shape[0]
    dist = np.full((n, n), max_distance + 1, dtype=np.float32)
    np.fill_diagonal(dist, 0)
    a = (adj > 0).astype(np.float64)
    dist[adj > 0] = np.minimum(dist[adj > 0], 1)           # :115 (a self-loop keeps distance 0 as in :112,:119)
    np.fill_diagonal(dist, 0)
    power = a.copy()
    for d in range(2, max_distance + 1):
        power = (power @ a > 0).astype(np.float64)         # :118, :124: walks of length d exist
        mask = (power > 0) & (dist > d)                     # :119, :125
        dist[mask] = d
    return dist


def walk_probabilities(dist: np.ndarray, walk_type: str = "exp", alpha: float = 1.0, c: float = 0.1) -> np.ndarray:
    n = dist.shape[0]
    d64 = dist.astype(np.float64)
    if walk_type == "exp":
        w = np.exp(-alpha * d64)                            # :230
    elif walk_type == "levy":
        w = np.power(d64 + 1e-6, -alpha)                    # :233-234
    else:
        raise ValueError("Unsupported types of random walk: %s" % walk_type)
    np.fill_diagonal(w, 0)                                  # :231, :235
    rs = w.sum(axis=1, keepdims=True)
    rs[rs == 0] = 1                                         # :240
    g = w / rs
    return (1 - c) * g + c * np.ones((n, n)) / n            # :243-244


def apply_batch(adj: np.ndarray, graph_sizes, x: np.ndarray, max_distance=3, walk_type="exp", alpha=1.0, c=0.1):
    """adj: dense (N, N) block-diagonal adjacency; graph_sizes: nodes per graph in order; x: (N, D).
    Returns (P x, list of per-graph distance matrices)."""
    out = np.zeros_like(x, dtype=np.float64)
    dists = []
    lo = 0
    for ng in graph_sizes:
        sub = adj[lo:lo + ng, lo:lo + ng]
        dist = hop_distances(sub, max_distance)
        dists.append(dist)
        out[lo:lo + ng] = walk_probabilities(dist, walk_type, alpha, c) @ x[lo:lo + ng].astype(np.float64)
        lo += ng
    return out, dists
