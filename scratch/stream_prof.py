"""Where a streamed step spends its time: host time of batch assembly and of the eager step, device time of the
assembly launch, eager resident step against streamed step (stream_h64)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from gnn_amd.synth import CONFIGS
dev = torch.device("cuda:0")
cfg = sys.argv[1] if len(sys.argv) > 1 else "stream_h64"
hp = dict(CONFIGS[cfg]["hp"])
sb = bench.StreamBench(cfg, hp, dev, 31, seed=1)
sb.run(10)
torch.cuda.synchronize()
# host time of assembling one batch
t0 = time.perf_counter()
for i in range(50):
    b = sb.ds.batch(sb.id_lists[i % 31], slot=sb.tr.slots[0])
t1 = time.perf_counter()
torch.cuda.synchronize()
print("batch() host us:", (t1 - t0) / 50 * 1e6)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(50):
    b = sb.ds.batch(sb.id_lists[i % 31], slot=sb.tr.slots[0])
e1.record(); torch.cuda.synchronize()
print("batch() device us (back to back):", e0.elapsed_time(e1) * 1e3 / 50)
# eager step on one resident batch: host time without waiting, then device time
tr = sb.tr
tr.adj, tr.feats, tr.gi, tr.y = sb.ds.batch(sb.id_lists[0])
for _ in range(5):
    tr.step_eager()
torch.cuda.synchronize()
t0 = time.perf_counter()
e0.record()
for _ in range(50):
    tr.step_eager()
t1 = time.perf_counter()
e1.record(); torch.cuda.synchronize()
t2 = time.perf_counter()
print("eager resident: host issue us/step", (t1 - t0) / 50 * 1e6, "wall", (t2 - t0) / 50 * 1e6, "device", e0.elapsed_time(e1) * 1e3 / 50)
plan = sb.ds.plan(sb.id_lists)
torch.cuda.synchronize()
t0 = time.perf_counter()
e0.record()
for i in range(62):
    b = sb.ds.batch_planned(plan, i % 31, slot=sb.tr.slots[0])
t1 = time.perf_counter()
e1.record(); torch.cuda.synchronize()
print("batch_planned(): host us", (t1 - t0) / 62 * 1e6, "device us", e0.elapsed_time(e1) * 1e3 / 62)
for side in (True, False):
    sb2 = bench.StreamBench(cfg, hp, dev, 31, seed=1, side_stream=side)
    sb2.run(8)
    for _ in range(3):
        el, dm = bench._timed(sb2.run, 62)
        print("streamed, side_stream=%s: wall ms/step %.4f device %.4f" % (side, el / 62 * 1e3, dm / 62))
