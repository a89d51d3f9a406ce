"""A/B of the SpMM schedules: packed row-lane kernel vs tile kernel (row_lanes = 0)."""
import sys
import torch
sys.path.insert(0, '.')
import bench
from gnn_amd import get_structure
from gnn_amd.synth import CONFIGS, config_batch

dev = torch.device('cuda:0')
for spec in sys.argv[1:]:
    parts = spec.split('@')
    name, scale = parts[0], parts[1]
    hp = CONFIGS[name]['hp']
    bt = bench.tile_batch(config_batch(name, 1).to_torch(dev), int(scale))
    st = get_structure(bt[0], bt[2])
    d = int(parts[2]) if len(parts) > 2 else hp['hidden_dim']
    x = torch.rand((st.n, d), device=dev)
    outs = {}
    sched = (st.csr.row_lanes, st.csr.long_rows, st.csr.long_threshold)
    for label in ('packed', 'tile'):
        if label == 'tile':
            st.csr.row_lanes, st.csr.long_rows, st.csr.long_threshold = 0, None, 0
        else:
            st.csr.row_lanes, st.csr.long_rows, st.csr.long_threshold = sched if sched[0] else (1, None, 0)
        y = torch.zeros((st.n, d), device=dev)
        for _ in range(3):
            st.spmm(x, y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        it = 30
        e0.record()
        for _ in range(it):
            st.spmm(x, y)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / it
        outs[label] = y
        print(f"{spec} d={d} {label}: {us:.1f} us  {st.spmm_bytes(d) / us / 1e3:.0f} GB/s", flush=True)
    ref = torch.sparse.mm(bt[0].coalesce(), x) if st.n < 2_000_000 else outs['packed']
    for label, y in outs.items():
        err = (y - ref).abs().max().item() / ref.abs().max().item()
        print(f"   {label} rel err vs torch.sparse.mm: {err:.2e}; accumulate check:", end=' ')
        y2 = y.clone()
        st.spmm(x, y2, accumulate=True)
        print(f"{((y2 - 2 * y).abs().max().item()):.2e}", flush=True)
