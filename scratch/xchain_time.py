"""The one-launch X chain against the launches it replaces, each replayed from a HIP graph (proteins1024)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_amd import _lib, get_structure
from gnn_amd.eval import _facts
from gnn_amd.synth import config_batch
lib = _lib.load()
dev = torch.device("cuda:0")
adj, feats, gi, y = config_batch("proteins1024", 1).to_torch(dev)
st = get_structure(adj, gi)
n, g = st.n, st.n_graphs
h, s, d = 16, 5, 4
w = torch.randn(4, 1, device=dev); b = torch.randn(4, device=dev)
adj_h = torch.rand(h, 10, device=dev); fh = torch.rand(h, s, d, device=dev)
ah = torch.empty(h, s, s, device=dev)
max_nodes = _facts(st)[0]

def timed(fn, name):
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        fn(); fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(20):
            fn()
    for _ in range(3): gr.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): gr.replay()
    e1.record(); torch.cuda.synchronize()
    print("%-60s %.2f us" % (name, e0.elapsed_time(e1) * 1e3 / 200))

for steps in (1, 2):
    xs = torch.empty(steps, n, 4, device=dev); z = torch.empty(steps, h, s, d, device=dev)
    def chain(hh=h):
        _lib.check(lib.rwk_x_chain_hidden_forward(feats.data_ptr(), w.data_ptr(), b.data_ptr(), n, 1, st.csr.crow.data_ptr(),
            st.csr.col.data_ptr(), None, st.seg_ptr.data_ptr(), g, max_nodes, steps, xs.data_ptr(), adj_h.data_ptr(),
            fh.data_ptr(), hh, s, ah.data_ptr(), z.data_ptr(), None, None, _lib.current_stream(dev)), "x")
    timed(chain, "x chain + hidden graphs, steps=%d" % steps)
xs = torch.empty(2, n, 4, device=dev); z = torch.empty(2, h, s, d, device=dev)
def proj():
    _lib.check(lib.rwk_project_hidden_forward(feats.data_ptr(), w.data_ptr(), b.data_ptr(), n, 1, 4, xs.data_ptr(),
        adj_h.data_ptr(), fh.data_ptr(), h, s, 2, ah.data_ptr(), z.data_ptr(), None, None, _lib.current_stream(dev)), "p")
timed(proj, "projection + hidden graphs")
timed(lambda: st.spmm(xs[0], xs[1]), "SpMM")
def both():
    proj(); st.spmm(xs[0], xs[1])
timed(both, "projection + hidden graphs, then SpMM")
def hid():
    _lib.check(lib.rwk_hidden_forward(adj_h.data_ptr(), fh.data_ptr(), h, s, d, 2, ah.data_ptr(), z.data_ptr(), None, None, _lib.current_stream(dev)), "h")
timed(hid, "hidden graphs alone")
