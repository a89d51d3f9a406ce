"""Times the projection kernels alone (HIP events on the current stream): python scratch/proj_time.py"""
import torch
from gnn_amd import _lib

import os
if os.environ.get("RWK_LIB"):
    _lib.LIB_PATH = os.environ["RWK_LIB"]
lib = _lib.load()
dev = torch.device("cuda:0")
stream = _lib.current_stream(dev)


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for n, f, d in [(102962, 16, 64), (102962, 1, 64), (102962, 16, 4), (102962, 16, 16), (307842, 1, 4)]:
    feat = torch.randn(n, f, device=dev)
    w = torch.randn(d, f, device=dev)
    b = torch.randn(d, device=dev)
    x0 = torch.empty(n, d, device=dev)
    dx0 = torch.randn(n, d, device=dev)
    blocks = lib.rwk_project_backward_blocks(n, d)
    partials = torch.empty(blocks, d * f + d, device=dev)
    out = {}
    out["fwd_sigmoid"] = timed(lambda: lib.rwk_project_forward(feat.data_ptr(), w.data_ptr(), b.data_ptr(), n, f, d,
                                                               x0.data_ptr(), stream))
    out["fwd_linear"] = timed(lambda: lib.rwk_linear_forward(feat.data_ptr(), w.data_ptr(), b.data_ptr(), n, f, d,
                                                             x0.data_ptr(), stream))
    out["bwd_sigmoid"] = timed(lambda: lib.rwk_project_backward(dx0.data_ptr(), x0.data_ptr(), feat.data_ptr(),
                                                                w.data_ptr(), n, f, d, partials.data_ptr(), None,
                                                                stream))
    out["bwd_linear"] = timed(lambda: lib.rwk_linear_backward(dx0.data_ptr(), feat.data_ptr(), w.data_ptr(), n, f, d,
                                                              partials.data_ptr(), None, stream))
    fill = timed(lambda: x0.fill_(1.0))
    print(n, f, d, "blocks", blocks, {k: round(v, 1) for k, v in out.items()}, "fill(n*d) %.1f us" % fill, flush=True)
