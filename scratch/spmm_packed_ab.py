"""Same-process A/B of the packed SpMM kernel (rwk_spmm_csr with row_lanes >= 1) of two builds of the library:
python scratch/spmm_packed_ab.py <other library>.  Prints microseconds per launch and whether the bits agree."""
import ctypes
import sys

import numpy as np
import torch

libs = {"this": ctypes.CDLL("gnn_amd/librwk.so"), "other": ctypes.CDLL(sys.argv[1])}
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)


def case(n, mean, d, lanes):
    deg = rng.poisson(mean, n).clip(1, None)
    crow = np.zeros(n + 1, np.int32)
    crow[1:] = np.cumsum(deg)
    nnz = int(crow[-1])
    rows = np.repeat(np.arange(n), deg)
    col = ((rows + rng.integers(-200, 200, nnz)) % n).astype(np.int32)
    val = rng.standard_normal(nnz).astype(np.float32)
    x = torch.randn(n, d, device=dev)
    t = [torch.from_numpy(a).to(dev) for a in (crow, col, val)]
    outs = {}
    for name, lib in libs.items():
        f = lib.rwk_spmm_csr
        f.restype = ctypes.c_int
        y = torch.zeros(n, d, device=dev)
        args = [ctypes.c_void_p(t[0].data_ptr()), ctypes.c_void_p(t[1].data_ptr()), ctypes.c_void_p(t[2].data_ptr()),
                ctypes.c_int64(n), ctypes.c_int64(d), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()),
                ctypes.c_int(0), ctypes.c_int32(lanes), ctypes.c_void_p(0), ctypes.c_int64(0), ctypes.c_int32(0),
                ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)]
        for _ in range(5):
            assert f(*args) == 0
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                f(*args)
            b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) / 20 * 1e3)
        outs[name] = (best, y.clone())
    same = torch.equal(outs["this"][1], outs["other"][1])
    print("n=%d mean=%g d=%d lanes=%d: this %.1f us, other %.1f us, same bits %s"
          % (n, mean, d, lanes, outs["this"][0], outs["other"][0], same), flush=True)


for n, mean, d, lanes in ((500000, 4.6, 64, 1), (500000, 4.6, 16, 1), (500000, 4.6, 4, 1), (2000000, 4.6, 4, 1),
                          (100000, 40, 16, 8), (100000, 40, 4, 8), (40000, 4.6, 64, 1), (100000, 12, 6, 2)):
    case(n, mean, d, lanes)
