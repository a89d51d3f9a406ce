"""Streamed passes with the steps on a high-priority stream (the assembly's side stream keeps the lowest priority):
does the assembly launch, which shares the chip with the step's short launches, get out of their way?
python scratch/stream_priority.py [config]"""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from gnn_amd.synth import CONFIGS  # noqa: E402

config = sys.argv[1] if len(sys.argv) > 1 else "proteins1024"
dev = torch.device("cuda:0")
hp = dict(CONFIGS[config]["hp"])
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "?")


def timed(sb, passes):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sb.run(31 * passes)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / (31 * passes)


for name, prio in (("default stream", None), ("priority -1", -1), ("default stream", None), ("priority -1", -1)):
    sb = bench.StreamBench(config, hp, dev, 31, seed=4321)
    if prio is None:
        sb.run(62)
        ms = [timed(sb, 4) for _ in range(3)]
    else:
        hi = torch.cuda.Stream(device=dev, priority=prio)
        hi.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(hi):
            sb.run(62)
            ms = [timed(sb, 4) for _ in range(3)]
    print("%-16s %s ms/step" % (name, " ".join("%.4f" % v for v in ms)), flush=True)
    del sb
