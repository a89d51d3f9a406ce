"""Durations of the per-graph chain launches (csrc/rwk_chain.hip) by number of steps, on graphs with and without hubs.
Usage on the GPU box: python scratch/chain_time.py"""
import sys
import torch
from gnn_amd import _lib
from gnn_amd.eval import _facts
from gnn_amd.structure import get_structure
from gnn_amd.synth import powerlaw_batch, random_batch

dev = torch.device("cuda:0")
lib = _lib.load()


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


cases = {"powerlaw64": powerlaw_batch(64, 2000, 4000, 2.1, 1), "sparse64": random_batch(64, 2000, 4000, 3.3, 1),
         "powerlaw256": powerlaw_batch(256, 2000, 4000, 2.1, 1)}
for name, b in cases.items():
    adj, feats, gi, _ = b.to_torch(dev)
    st = get_structure(adj, gi)
    max_nodes, _ = _facts(st)
    n, f = feats.shape
    w = torch.randn(4, f, device=dev) * 0.3
    bias = torch.zeros(4, device=dev)
    h, s = 16, 5
    adj_h = torch.rand(h, 10, device=dev)
    feat_h = torch.rand(h, s, 4, device=dev)
    stream = _lib.current_stream(dev)
    csr, csr_t = st.csr, st.csr_t
    deg = (csr.crow[1:] - csr.crow[:-1])
    print(name, "nodes", n, "nnz", st.nnz, "max_nodes", max_nodes, "max_degree", csr.max_degree,
          "rows>16", int((deg > 16).sum()), "rows>128", int((deg > 128).sum()), flush=True)
    for steps in (1, 2, 3, 4):
        xs = torch.empty(steps, n, 4, device=dev)
        ah = torch.empty(h, s, s, device=dev)
        z = torch.empty(steps, h, s, 4, device=dev)

        def fwd():
            _lib.check(lib.rwk_graph_chain_forward(
                feats.data_ptr(), w.data_ptr(), bias.data_ptr(), n, f, csr.crow.data_ptr(), csr.col.data_ptr(),
                None if csr.unit_values else csr.val.data_ptr(), st.seg_ptr.data_ptr(), st.n_graphs, max_nodes, steps,
                xs.data_ptr(), adj_h.data_ptr(), feat_h.data_ptr(), h, s, ah.data_ptr(), z.data_ptr(), None, None,
                stream), "fwd")

        def fwd_launches():
            _lib.check(lib.rwk_project_hidden_forward(feats.data_ptr(), w.data_ptr(), bias.data_ptr(), n, f, 4,
                                                      xs.data_ptr(), adj_h.data_ptr(), feat_h.data_ptr(), h, s, steps,
                                                      ah.data_ptr(), z.data_ptr(), None, None, stream), "p")
            for i in range(1, steps):
                st.spmm(xs[i - 1], xs[i])

        line = "  steps %d: forward chain %6.1f us, launches %6.1f us" % (steps, timed(fwd), timed(fwd_launches))
        if steps >= 2:
            dx = torch.randn(steps, n, 4, device=dev)
            x0 = torch.rand(n, 4, device=dev)
            partials = torch.empty(st.n_graphs, 4 * f + 4, device=dev)

            def bwd():
                _lib.check(lib.rwk_graph_chain_backward(
                    dx.data_ptr(), x0.data_ptr(), feats.data_ptr(), w.data_ptr(), n, f, csr_t.crow.data_ptr(),
                    csr_t.col.data_ptr(), None if csr_t.unit_values else csr_t.val.data_ptr(), st.seg_ptr.data_ptr(),
                    st.n_graphs, max_nodes, steps, partials.data_ptr(), None, None, 0, 0, None, stream), "bwd")
            line += ", backward chain %6.1f us" % timed(bwd)
        print(line, flush=True)
