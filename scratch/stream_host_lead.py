"""Where a streamed pass spends its host time: per pass the host duration of run_pass (launches only), the wait in the
synchronize behind it, and the host time of DeviceDataset.plan alone.  python scratch/stream_host_lead.py [config] [serial]"""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from gnn_amd.synth import CONFIGS  # noqa: E402

config = sys.argv[1] if len(sys.argv) > 1 else "stream_h64"
dev = torch.device("cuda:0")
hp = dict(CONFIGS[config]["hp"])
side = {"serial": False, "side": True}.get(sys.argv[2] if len(sys.argv) > 2 else "", None)      # "serial": the assembly on the step's own stream
if len(sys.argv) > 2 and sys.argv[2] == "auto_launch":       # the assembly launch in front of every step
    import gnn_amd.composite
    gnn_amd.composite.ASSEMBLE_IN_FORWARD = False
sb = bench.StreamBench(config, hp, dev, 31, seed=4321, side_stream=side)
print("assembly on", {True: "a side stream", False: "the launch stream", None: "the stream the trainer picks"}[side])
sb.run(31)
torch.cuda.synchronize()
for p in range(4):
    t0 = time.perf_counter()
    plan = sb.ds.plan(sb.id_lists)
    t1 = time.perf_counter()
    del plan
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    sb.run(31)
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print("pass %d: plan alone %.2f ms | run_pass host %.2f ms, sync wait %.2f ms, total %.3f ms/step" % (
        p, (t1 - t0) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t4 - t2) * 1e3 / 31), flush=True)
# back to back passes, one sync at the end
t0 = time.perf_counter()
sb.run(31 * 4)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("4 passes back to back: host %.2f ms, sync wait %.2f ms, %.3f ms/step" % (
    (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) * 1e3 / 124), flush=True)
