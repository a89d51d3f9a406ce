"""Evaluation over batches that are never assembled: eval_forward_planned (one launch per batch, reading the data set)
against batch assembly + model(...) per batch, and against one resident batch.  python scratch/eval_stream_time.py"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from gnn_amd import RW_NN  # noqa: E402
from gnn_amd.data import DeviceDataset  # noqa: E402
from gnn_amd.eval import eval_forward_planned  # noqa: E402
from gnn_amd.synth import CONFIGS, config_batch  # noqa: E402

dev = torch.device("cuda:0")
hp = dict(CONFIGS["proteins1024"]["hp"])
pool = 31
whole = config_batch("proteins1024", seed=4321, scale=pool)
per = whole.n_graphs // pool
ds = DeviceDataset.from_batch(whole, dev)
id_lists = [np.arange(i * per, (i + 1) * per) for i in range(pool)]
torch.manual_seed(0)
m = RW_NN(hp["input_dim"], hp["max_step"], hp["hidden_graphs"], hp["size_hidden_graphs"], hp["hidden_dim"],
          hp["penultimate_dim"], hp["normalize"], hp["n_classes"], hp["dropout"], dev).to(dev).eval()
plan = ds.plan(id_lists)
outs = [torch.empty(per, hp["n_classes"], device=dev) for _ in range(pool)]


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / pool)
    return best * 1e6


def gather_pass():
    for i in range(pool):
        assert eval_forward_planned(m, ds, plan, i, outs[i]) is not None


slots = ds.make_slots(3, per, max(ds.batch_shape(i)[0] for i in id_lists), max(ds.batch_shape(i)[1] for i in id_lists))


def assembled_pass():
    with torch.no_grad():
        for i in range(pool):
            adj, fe, gi, _ = ds.batch_planned(plan, i, slots[i % 3])
            m(adj, fe, gi)


res = ds.batch_planned(plan, 0)


def resident_pass():
    with torch.no_grad():
        for i in range(pool):
            m(res[0], res[1], res[2])


print("per 1024-graph batch, us (host clock over a pass of %d batches, best of 5):" % pool)
print("  straight from the data set (eval_forward_planned): %.1f" % timed(gather_pass))
print("  batch assembly + model(...):                       %.1f" % timed(assembled_pass))
print("  model(...) on one resident batch:                  %.1f" % timed(resident_pass))
