"""One rank of the two-process data-parallel check on the HIP kernels (started by tests/conftest.py through
``python -m torch.distributed.run --nproc-per-node 2``, before the test process touches the GPU; the ranks import what
they need and then wait for the test's go-ahead; both ranks share GPU 0, the gradient all-reduce goes over gloo).  Each rank takes its ``gnn_amd.dp.shard_ranges`` share of ONE 128-graph batch.

1. Layer level (no batch-norm, no dropout): the summed per-shard gradients of ``sum(K * W)`` equal the gradient one
   process computes on the whole batch, to 1e-5 (K rows depend on their own graph only, so the sum is exact up to
   rounding).
2. Training steps (``Trainer(world=2)``, dropout on, per-rank masks): three eager steps, a capture, two replayed
   steps; the ranks must end with bit-identical parameters and moments.
Writes ``rank<r>.json`` into the directory given as the only argument."""
import json
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(outdir):
    import numpy as np
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    result = {"rank": rank, "ok": False}
    import faulthandler
    try:
        # Everything is imported; the GPU work waits until the test asks for it (the file "go"), so that the ranks do
        # not share the device with whatever other tests the session is running meanwhile.
        import time
        t_end = time.time() + 1800
        while not os.path.exists(os.path.join(outdir, "go")):
            if time.time() > t_end:
                result.update(ok=True, skipped="the test that starts the ranks never ran")
                return result
            time.sleep(0.05)
        # the GPU part takes seconds: a rank still here after four minutes writes every thread's stack to the log
        # (stderr) and exits, so that a hang names the call it sits in
        faulthandler.dump_traceback_later(240, exit=True)
        if not torch.cuda.is_available():
            result.update(ok=True, skipped="no GPU")
            return result
        from gnn_amd import RW_NN
        from gnn_amd.dp import FlatGradBucket, broadcast_parameters, shard_ranges
        from gnn_amd.synth import random_batch
        from gnn_amd.trainer import Trainer
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group("gloo")
        whole = random_batch(128, 20, 60, 3.0, seed=21)
        ranges = shard_ranges(whole.sizes, world)
        lo, hi = ranges[rank]
        shard = whole.slice_graphs(lo, hi)
        result["range"] = [lo, hi]
        hp = dict(input_dim=1, max_step=2, hidden_graphs=16, size_hidden_graphs=5, hidden_dim=4)

        def model(drop):
            torch.manual_seed(5)
            m = RW_NN(hp["input_dim"], hp["max_step"], hp["hidden_graphs"], hp["size_hidden_graphs"],
                      hp["hidden_dim"], 32, False, 2, drop, dev).to(dev).train()
            m.seed_dropout(1000 + rank)
            return m

        # ---- 1. layer level: all-reduced shard gradients == whole-batch gradient
        w_all = torch.from_numpy(np.random.default_rng(9).standard_normal((128, 32)).astype(np.float32)).to(dev)

        def layer_grads(m, batch, w):
            bucket = FlatGradBucket(m)
            bucket.release()
            adj, feats, gi, _ = batch.to_torch(dev)
            (m.rw_features(adj, feats, gi) * w).sum().backward()
            bucket.gather()
            torch.cuda.synchronize()
            return bucket

        m = model(0.0)
        broadcast_parameters(m)
        bucket = layer_grads(m, shard, w_all[lo:hi])
        bucket.all_reduce(average=False)
        torch.cuda.synchronize()
        summed = bucket.flat.clone()
        ref = layer_grads(model(0.0), whole, w_all).flat
        n_layer = sum(p.numel() for p in (m.adj_hidden, m.features_hidden, m.fc.weight, m.fc.bias))
        off, errs = 0, {}
        for name, p in (("adj_hidden", m.adj_hidden), ("features_hidden", m.features_hidden),
                        ("fc.weight", m.fc.weight), ("fc.bias", m.fc.bias)):
            a, b = summed[off:off + p.numel()].double(), ref[off:off + p.numel()].double()
            errs[name] = float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
            off += p.numel()
        result["layer_grad_rel_err"] = errs
        layer_ok = max(errs.values()) < 1e-5 and float(ref[:n_layer].abs().max()) > 0

        # ---- 2. training steps on two real processes
        m = model(0.2)
        broadcast_parameters(m)
        tr = Trainer(m, shard.to_torch(dev), world=world, sync_before_all_reduce=True)
        for _ in range(3):
            tr.step_eager()
        tr.capture()
        for _ in range(2):
            tr.step_graph()
        torch.cuda.synchronize()
        result["loss"] = float(tr.loss)
        same = True
        for t in (tr.opt.flat_param, tr.opt.exp_avg, tr.opt.exp_avg_sq):
            mine = t.detach().cpu()
            both = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(both, mine)
            same = same and all(torch.equal(both[0], x) for x in both[1:])
        result["params_bit_identical"] = bool(same)
        result["finite"] = bool(torch.isfinite(tr.opt.flat_param).all()) and bool(np.isfinite(result["loss"]))
        result["steps"] = int(tr.opt.step_counters[0])
        result["ok"] = bool(layer_ok and same and result["finite"] and result["steps"] == 3 + 3 + 2)
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        result["error"] = traceback.format_exc()
    return result


if __name__ == "__main__":
    out = sys.argv[1]
    res = main(out)
    with open(os.path.join(out, "rank%d.json" % res["rank"]), "w") as f:
        json.dump(res, f)
    sys.exit(0 if res["ok"] else 1)
