"""What gathering through LDS costs at the roofline size: the one-launch X chain with one and with two steps on the
protein-sized batch tiled 64 times (19.7 M nodes); the difference is one propagation step with the graph's rows in LDS,
to hold against the tile SpMM on the same matrix."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gnn_amd import _lib, get_structure
from gnn_amd.eval import _facts
from gnn_amd.synth import config_batch
lib = _lib.load()
dev = torch.device("cuda:0")
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 64
adj, feats, gi, y = bench.tile_batch(config_batch("proteins1024", 1).to_torch(dev), scale)
st = get_structure(adj, gi)
n, g = st.n, st.n_graphs
h, s, d = 16, 5, 4
w = torch.randn(4, 1, device=dev); b = torch.randn(4, device=dev)
adj_h = torch.rand(h, 10, device=dev); fh = torch.rand(h, s, d, device=dev)
ah = torch.empty(h, s, s, device=dev)
max_nodes = _facts(st)[0]


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


res = {}
for steps in (1, 2):
    xs = torch.empty(steps, n, 4, device=dev); z = torch.empty(steps, h, s, d, device=dev)
    def chain():
        _lib.check(lib.rwk_x_chain_hidden_forward(feats.data_ptr(), w.data_ptr(), b.data_ptr(), n, 1, st.csr.crow.data_ptr(),
            st.csr.col.data_ptr(), None, st.seg_ptr.data_ptr(), g, max_nodes, steps, xs.data_ptr(), adj_h.data_ptr(),
            fh.data_ptr(), h, s, ah.data_ptr(), z.data_ptr(), None, None, _lib.current_stream(dev)), "x")
    res[steps] = timed(chain)
    print("x chain, steps=%d: %.1f us" % (steps, res[steps]))
xs = torch.empty(2, n, 4, device=dev)
print("one step through LDS: %.1f us; tile SpMM: %.1f us" % (res[2] - res[1], timed(lambda: st.spmm(xs[0], xs[1]))))
