"""Host-side profile of a streamed pass at the proteins1024 shape (StreamTrainer.run_pass)."""
import cProfile, pstats, io, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from gnn_amd.synth import CONFIGS

dev = torch.device("cuda:0")
hp = dict(CONFIGS["proteins1024"]["hp"])
sb = bench.StreamBench("proteins1024", hp, dev, 31, seed=1, optimizer="flat")
sb.run(31)
torch.cuda.synchronize()
t = time.perf_counter(); sb.run(62); torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t) / 62 * 1e3)
pr = cProfile.Profile(); pr.enable(); sb.run(62); torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(40); print(s.getvalue()[:7000])
