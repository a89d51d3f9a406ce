"""Times the distance-biased operator's structure build and application: PYTHONPATH=. python scratch/dbias_time.py"""
import time
import torch
from gnn_amd.synth import config_batch
from gnn_amd.distance_bias import distance_bias_structure, distance_bias_propagate
from gnn_amd.structure import clear_structure_cache

dev = torch.device("cuda:0")
for name, d in [("proteins1024", 4), ("stream_h64", 64), ("single_graph", 64), ("powerlaw", 4)]:
    b = config_batch(name, 1)
    adj, feats, gi, y = b.to_torch(dev)
    st = distance_bias_structure(adj, gi, 3)          # warm-up (also builds the base structure)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = distance_bias_structure(adj, gi, 3)
    torch.cuda.synchronize()
    t_build = (time.perf_counter() - t0) * 1e3
    x = torch.randn(b.n_nodes, d, device=dev)
    for _ in range(3):
        distance_bias_propagate(x, st)
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        distance_bias_propagate(x, st)
    e.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(e) / 20 * 1e3
    nnz = st.col.numel()
    bytes_ = nnz * 8 + (b.n_nodes + 1) * 4 + 4 * b.n_nodes * d * 4
    print("%s: N %d, A nnz %d, pairs within 3 hops %d (%.1f per node); build %.2f ms; P x (D=%d) %.1f us, %.0f GB/s on %d MB"
          % (name, b.n_nodes, b.nnz, nnz, nnz / b.n_nodes, t_build, d, us, bytes_ / us / 1e3, bytes_ >> 20), flush=True)
