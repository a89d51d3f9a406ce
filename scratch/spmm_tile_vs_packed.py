"""Tile kernel (row_lanes = 0, int32 columns) against the packed kernel (row_lanes = 1) of the same library on
protein-like matrices: python scratch/spmm_tile_vs_packed.py.  Microseconds per launch."""
import ctypes

import numpy as np
import torch

lib = ctypes.CDLL("gnn_amd/librwk.so")
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
f = lib.rwk_spmm_csr
f.restype = ctypes.c_int


def case(n, mean, d):
    deg = rng.poisson(mean, n).clip(1, None)
    crow = np.zeros(n + 1, np.int32)
    crow[1:] = np.cumsum(deg)
    nnz = int(crow[-1])
    rows = np.repeat(np.arange(n), deg)
    col = ((rows + rng.integers(-150, 150, nnz)) % n).astype(np.int32)
    val = np.ones(nnz, np.float32)
    x = torch.randn(n, d, device=dev)
    t = [torch.from_numpy(a).to(dev) for a in (crow, col, val)]
    res = {}
    for name, lanes in (("tile", 0), ("packed", 1)):
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
        res[name] = (best, y)
    err = (res["tile"][1] - res["packed"][1]).abs().max().item()
    print("n=%d mean=%g d=%d: tile %.1f us, packed %.1f us (max diff %.2g)"
          % (n, mean, d, res["tile"][0], res["packed"][0], err), flush=True)


for n in (40000, 300000, 2000000):
    for d in (4, 8, 16, 32, 64):
        case(n, 3.7, d)
