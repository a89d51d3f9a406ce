#!/usr/bin/env python3
"""Benchmark of the RW_NN training step on MI355X.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step is the reference's ``train()`` (rwgnn2/main.py:82-88) on one resident batch: zero the gradients,
``model(adj, features, graph_indicator)``, cross-entropy, backward, (gradient all-reduce over the ranks),
Adam update.  The workload is BASELINE.json configs[1]: 1024 PROTEINS/D&D-scale graphs (100..500 nodes, mean
degree 3.7) per GPU, main.py's default hyper-parameters (16 hidden graphs x 5 nodes, hidden_dim 4, max_step 2,
dropout 0.2).  Each rank holds its own batch (weak scaling); value = graphs of all ranks / slowest rank's time.

Rank 0 prints ONE JSON line.  Besides the contract keys it carries
  "roofline":     the dominant hand-written kernel over its launch duration (HIP events on the launch stream,
                  inside this process): algorithmic flops against the dense fp32 peak when its arithmetic
                  intensity is above the ridge point (the fused kernels), else algorithmic HBM bytes against 8 TB/s;
  "kernels":      the same figure for each kernel class (SpMM, fused forward, fused backward);
  "cpu_baseline": the operator-for-operator PyTorch-CPU port of the reference (oracle/rwnn_torch_port.py)
                  timed on this host on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0
FP32_PEAK_TFLOPS = 157.3      # dense fp32 MFMA peak = fp32 vector peak


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="proteins1024",
                    help="synthetic workload of gnn_amd.synth.CONFIGS: proteins1024 (the metric's), stream_h64, "
                         "powerlaw, single_graph (one graph: --kernels-only, a batch of one cannot train)")
    ap.add_argument("--scale", type=int, default=1, help="multiply the number of graphs per batch")
    ap.add_argument("--mode", choices=["graph", "eager"], default="graph",
                    help="replay the step from a captured HIP graph (default) or launch it eagerly")
    ap.add_argument("--optimizer", choices=["flat", "torch"], default="flat",
                    help="flat: gnn_amd.optim.FlatAdam (one launch); torch: torch.optim.Adam(capturable)")
    ap.add_argument("--steps-per-graph", type=int, default=5,
                    help="graph mode, one process: consecutive steps captured in one graph (whole replays first, the "
                         "remaining steps as single-step replays); 1 = one step per replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--cpu-sample-graphs", type=int, default=1024)
    ap.add_argument("--kernels-only", action="store_true",
                    help="no training loop: time the hand-written kernels on the (tiled) batch and print them")
    ap.add_argument("--roofline-scale", type=int, default=1,
                    help="with --kernels-only: tile the batch this many times on the device, so that the working "
                         "set leaves the 256 MB Infinity Cache")
    ap.add_argument("--dropout", type=float, default=None, help="override the configuration's dropout rate")
    ap.add_argument("--calibrate", action="store_true",
                    help="with --kernels-only: also launch a 1 GiB elementwise read+write, the known byte count "
                         "the FETCH_SIZE / WRITE_SIZE counter passes are calibrated on")
    ap.add_argument("--only", default="", help="with --kernels-only: comma-separated kernel classes to launch")
    ap.add_argument("--capture-allreduce", action="store_true",
                    help="(the default on RCCL; kept for older command lines) capture the gradient all-reduce inside "
                         "the step graph")
    ap.add_argument("--no-capture-allreduce", action="store_true",
                    help="--gpus N > 1: graph of forward + backward, eager all-reduce, Adam launch (the fall-back shape); "
                         "also RWK_CAPTURE_ALLREDUCE=0")
    ap.add_argument("--dp-shape", type=int, default=0,
                    help="one GPU only: time the step in the shape a rank of an N-rank run executes (two graphs and "
                         "the collective's slot between them, no process group) to see what the shape itself costs")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="--gpus N: weak = every rank its own batch of the configuration (default); strong = ONE batch "
                         "of the configuration cut into N contiguous graph ranges balanced by node count "
                         "(gnn_amd.dp.shard_ranges), total work fixed")
    ap.add_argument("--stream", action="store_true",
                    help="never-repeating batches: a pool of --pool-batches distinct batches of the configuration as "
                         "one device-resident data set, each batch assembled (DeviceDataset.batch, structure included) "
                         "inside the timed region right before its step, on a side stream under the previous step "
                         "(gnn_amd.trainer.StreamTrainer); steps are launched eagerly")
    ap.add_argument("--pool-batches", type=int, default=31,
                    help="--stream: distinct batches per pass (31 x 4096 = 127 k graphs: one GPU's share of the "
                         "1 M-graph stream of BASELINE.json configs[2])")
    ap.add_argument("--distinct-batches", type=int, default=0,
                    help="time an epoch over this many different resident batches of the configuration replayed as one "
                         "HIP graph (gnn_amd.trainer.EpochTrainer; 16 x 23 MB leaves the 256 MB Infinity Cache) "
                         "instead of one resident batch")
    ap.add_argument("--no-extras", action="store_true",
                    help="default run: skip the extra measurements (distinct batches, stream_h64 stream, drop-in)")
    return ap.parse_args()


from gnn_amd.trainer import Trainer  # noqa: E402  (the step being timed)


def tile_batch(batch_t, k):
    """k copies of a resident batch as one block-diagonal batch, built on the device (the roofline-sized
    variant of a configuration: same graphs, working set far beyond the 256 MB Infinity Cache)."""
    adj, feats, gi, y = batch_t
    if k <= 1:
        return batch_t
    dev = feats.device
    n = adj.shape[0]
    g = int(gi.max().item()) + 1
    idx, val = adj._indices(), adj._values()
    node_off = torch.arange(k, device=dev, dtype=torch.int64) * n
    idx_k = (idx[:, None, :] + node_off[None, :, None]).reshape(2, -1)          # stays row-major sorted
    adj_k = torch.sparse_coo_tensor(idx_k, val.repeat(k), (n * k, n * k))
    graph_off = torch.arange(k, device=dev, dtype=torch.int64) * g
    gi_k = (gi[None, :] + graph_off[:, None]).reshape(-1)
    return adj_k, feats.repeat(k, 1), gi_k, y.repeat(k)


def time_kernels(batch_t, hp, budget_ms=150.0, only=None, in_step=False):
    """Average launch duration of each hand-written kernel class, by launching it back to back through the
    C ABI between two events on the launch stream, with its algorithmic HBM bytes and flops."""
    from gnn_amd import _lib, get_structure
    from gnn_amd.function import choose_stream_len
    lib = _lib.load()
    adj, feats, gi, _ = batch_t
    dev = feats.device
    st = get_structure(adj, gi)
    n, g = st.n, st.n_graphs
    h, s, d, steps = hp["hidden_graphs"], hp["size_hidden_graphs"], hp["hidden_dim"], hp["max_step"]
    p = float(hp["dropout"])
    stream = _lib.current_stream(dev)
    path = lib.rwk_fused_path(d, h, s, steps)
    xs = torch.rand((steps, n, d), device=dev)
    z = torch.rand((steps, h, s, d), device=dev)
    dk = torch.randn((g, steps * h), device=dev)
    rng = torch.tensor([12345, 7], dtype=torch.int64, device=dev)
    stream_len = choose_stream_len(n, d, h, s, steps)
    stream_len_b = choose_stream_len(n, d, h, s, steps, backward=True)
    n_streams = lib.rwk_fused_n_streams(n, stream_len)
    k_part = torch.empty(lib.rwk_fused_k_partials_size(n, stream_len, g, h, steps), device=dev)
    n_part = lib.rwk_fused_dz_partials_count(n, stream_len_b, d, h, s, steps)
    dz_part = torch.empty((n_part, steps, h, s, d), device=dev)
    dx = torch.empty_like(xs)
    ws_n = lib.rwk_fused_backward_workspace(n, d, h, s, steps)
    ws = torch.empty(ws_n, device=dev) if ws_n else None
    y = torch.empty((n, d), device=dev)

    def run_spmm():
        st.spmm(xs[0], y)

    # the training step's form of the two launches: on the matrix-core path the forward leaves its dropout keep bits
    # in memory and the backward reads them (gnn_amd/function.py); elsewhere keep is None = the plain entry points
    keep_n = lib.rwk_fused_keep_words(n, d, h, s, steps) if p > 0 else 0
    keep = torch.empty(keep_n, dtype=torch.int64, device=dev) if keep_n else None

    def run_fwd():
        _lib.check(lib.rwk_fused_forward_bits(xs.data_ptr(), z.data_ptr(), st.gid.data_ptr(), _lib.ptr(st.perm), n, d,
                                              h, s, steps, stream_len, p, rng.data_ptr(), k_part.data_ptr(),
                                              _lib.ptr(keep), stream), "fwd")

    if keep is not None:
        run_fwd()                                             # the backward's timing loop reads what this one left

    def run_bwd():
        _lib.check(lib.rwk_fused_backward_bits(xs.data_ptr(), z.data_ptr(), st.gid.data_ptr(), _lib.ptr(st.perm),
                                               dk.data_ptr(), st.counts_f32.data_ptr(), n, d, h, s, steps,
                                               stream_len_b, p, 0, rng.data_ptr(), dx.data_ptr(), dz_part.data_ptr(),
                                               _lib.ptr(ws), _lib.ptr(keep), stream), "bwd")

    # The launch the single-process training step really runs at hidden_dim <= 4: the same backward kernel with the end
    # of the head's backward pass in its prologue (rwk_fused_backward_bn: it forms dK itself from what the head left).
    # Timed on buffers of the head's shapes; the roofline entry of the default line is taken from this launch.
    run_bwd_bn = None
    if in_step and p > 0 and "penultimate_dim" in hp and lib.rwk_fused_backward_bn_supported(d, h, s, steps):
        c_, p_, nc_ = steps * h, hp["penultimate_dim"], hp["n_classes"]
        if lib.rwk_head_train_supported(g, c_, p_, nc_):
            kk = torch.randn((g, c_), device=dev)
            dy = torch.randn((g, c_), device=dev) / g
            mean, invstd, bn_w = torch.zeros(c_, device=dev), torch.ones(c_, device=dev), torch.ones(c_, device=dev)
            hpart = torch.randn((lib.rwk_head_n_blocks(g), lib.rwk_head_grad_size(c_, p_, nc_)), device=dev) / g
            o_gamma = nc_ * p_ + nc_ + p_ * c_ + p_
            d_loss_one = torch.ones((), device=dev)

            def run_bwd_bn():
                _lib.check(lib.rwk_fused_backward_bn(
                    xs.data_ptr(), z.data_ptr(), st.gid.data_ptr(), _lib.ptr(st.perm), st.counts_f32.data_ptr(), n, d,
                    h, s, steps, stream_len_b, p, 0, rng.data_ptr(), dx.data_ptr(), dz_part.data_ptr(), kk.data_ptr(),
                    dy.data_ptr(), mean.data_ptr(), invstd.data_ptr(), bn_w.data_ptr(), hpart.data_ptr(),
                    hpart.shape[0], hpart.shape[1], o_gamma, d_loss_one.data_ptr(), g, stream), "bwd_bn")

    lh = steps * h
    # algorithmic HBM bytes per launch (fp32 and int32 are 4 bytes):
    #   SpMM:      values + columns + row pointers + x read once + y written once
    #   forward:   X_0..X_{L-1} read once, one graph id per node, K partial slots written, Z once
    #   backward:  X read + dX written, graph ids, dK, the per-workgroup dZ partial rows written
    bytes_ = {
        "spmm_csr": st.spmm_bytes(d),
        "fused_forward": steps * n * d * 4 + n * 4 + (n_streams + g) * lh * 4 + steps * h * s * d * 4 + keep_n * 8,
        "fused_backward": 2 * steps * n * d * 4 + n * 4 + g * lh * 4 + n_part * steps * h * s * d * 4 + keep_n * 8,
    }
    # algorithmic flops (the roofline's numerator): forward (L+1) contractions of 2 H S D flops per node (zx and the
    # L products T_i) plus 3 element-wise flops per (step, hidden node, node); backward twice the forward's
    # contractions.  "flops_executed" is what the kernels really issue: the forward shares zx with T_0 (L contractions),
    # the backward recomputes the forward's contractions and adds those of dX and dZ, plus 8 element-wise flops.
    contr = 2 * h * s * d * n
    flops = {
        "spmm_csr": 2 * st.nnz * d,
        "fused_forward": (steps + 1) * contr + 3 * steps * h * s * n,
        "fused_backward": 2 * (steps + 1) * contr,
    }
    flops_executed = {
        "spmm_csr": 2 * st.nnz * d,
        "fused_forward": steps * contr + 3 * steps * h * s * n,
        "fused_backward": 3 * steps * contr + 8 * steps * h * s * n,
    }
    # What limits a kernel: its arithmetic intensity (algorithmic flops / algorithmic bytes) against the ridge
    # point 157.3 TFLOP/s / 8 TB/s = 19.7 flop/B.  The SpMM (0.3..0.5 flop/B) is priced against HBM.  The fused
    # kernels sit at 40..70 flop/B (hidden_dim 4, vector ALU) or above 250 flop/B (matrix cores): fp32-compute
    # bound, priced against the dense fp32 peak (the fp32 MFMA peak and the fp32 vector peak are the same number).
    ridge = FP32_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
    bound = {name: ("mfma" if flops[name] / bytes_[name] > ridge else "hbm") for name in bytes_}
    unit_of = {"spmm_csr": "vector ALU", "fused_forward": "matrix cores" if path == 2 else "vector ALU",
               "fused_backward": "matrix cores" if path == 2 else "vector ALU"}
    out = {}
    launches = [("spmm_csr", run_spmm), ("fused_forward", run_fwd), ("fused_backward", run_bwd)]
    if run_bwd_bn is not None:
        launches.append(("fused_backward_in_step", run_bwd_bn))
        for tbl in (bytes_, flops, flops_executed, bound, unit_of):
            tbl["fused_backward_in_step"] = tbl["fused_backward"]
        bytes_["fused_backward_in_step"] += 2 * g * lh * 4          # k and dy read instead of d_k
    for name, fn in launches:
        if only and name not in only:
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        one = max(e0.elapsed_time(e1), 1e-3)
        iters = int(min(400, max(3, budget_ms / one)))
        for _ in range(min(5, iters)):
            fn()
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        gbs = bytes_[name] / (us * 1e-6) / 1e9
        tfs = flops[name] / (us * 1e-6) / 1e12
        out[name] = {"us": round(us, 3), "launches": iters, "bytes": int(bytes_[name]), "flops": int(flops[name]),
                     "GB/s": round(gbs, 1), "frac_hbm": round(gbs / HBM_PEAK_GBS, 4), "TFLOP/s": round(tfs, 3),
                     "frac_fp32_peak": round(tfs / FP32_PEAK_TFLOPS, 4), "bound": bound[name],
                     "flop_per_byte": round(flops[name] / bytes_[name], 1), "executes_on": unit_of[name],
                     "flops_executed": int(flops_executed[name]),
                     "TFLOP/s_executed": round(flops_executed[name] / (us * 1e-6) / 1e12, 3)}
        if name == "fused_backward":
            out[name]["launch"] = "rwk_fused_backward_bits, stand-alone (reads dK)"
        if name == "fused_backward_in_step":
            out[name]["launch"] = ("rwk_fused_backward_bn: the backward launch of the single-process training step "
                                   "(the head's batch-norm backward in its prologue)")
        if name == "spmm_csr":
            # the compact structure (16-bit column distances, no value array for unweighted graphs) moves fewer
            # bytes than the algorithmic count, which is defined on int32 columns and fp32 values
            moved = st.spmm_bytes_moved(d)
            out[name]["bytes_moved"] = int(moved)
            out[name]["GB/s_moved"] = round(moved / (us * 1e-6) / 1e9, 1)
    out["_shape"] = {"nodes": n, "nnz": st.nnz, "graphs": g, "stream_len": [stream_len, stream_len_b],
                     "fused_path": {0: "generic", 1: "register", 2: "mfma"}[path]}
    return out


def roofline_of(kern, name, traffic=None):
    k = kern[name]
    if k["bound"] == "mfma":
        return {"kernel": name, "bound": "mfma", "achieved": k["TFLOP/s"], "peak": FP32_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": k["frac_fp32_peak"], "traffic": traffic,
                "hbm_GB/s": k["GB/s"], "frac_hbm": k["frac_hbm"], "flop_per_byte": k["flop_per_byte"],
                "executes_on": k["executes_on"], "us": k["us"]}
    # HBM-bound kernels are priced on the bytes the scheduled kernel actually moves (the compact SpMM structure reads
    # 2 bytes per stored entry where the algorithmic count, defined on int32 columns and fp32 values, has 8); the
    # algorithmic figure is kept beside it
    moved = k.get("GB/s_moved", k["GB/s"])
    return {"kernel": name, "bound": "hbm", "achieved": moved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(moved / HBM_PEAK_GBS, 4), "traffic": traffic, "achieved_algorithmic": k["GB/s"],
            "frac_algorithmic": k["frac_hbm"], "bytes_moved": k.get("bytes_moved", k["bytes"]),
            "bytes_algorithmic": k["bytes"], "fp32_TFLOP/s": k["TFLOP/s"],
            "frac_fp32_peak": k["frac_fp32_peak"], "flop_per_byte": k["flop_per_byte"],
            "executes_on": k["executes_on"], "us": k["us"]}


def load_traffic(config, scale):
    """HBM bytes per launch from the committed counter passes (profiles/traffic.json: FETCH_SIZE and
    WRITE_SIZE collected in separate rocprofv3 --pmc runs of `bench.py --kernels-only`), or {}."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            return json.load(f).get(f"{config}@{scale}", {})
    except (OSError, ValueError):
        return {}


def drop_in_step(batch_t, hp, dev, steps=300, warmup=30, one_launch_adam=False):
    """The reference's training closure unchanged (rwgnn2/main.py:82-88: zero_grad, model(...), F.cross_entropy,
    backward, optimizer.step) on gnn_amd.RW_NN, launched eagerly: what a maintainer gets from the one-line import
    swap, without Trainer.capture or forward_with_loss - with torch.optim.Adam, or (``one_launch_adam``) with the second
    one-line swap ``optim.Adam`` -> ``gnn_amd.Adam`` at main.py:79."""
    import gnn_amd
    from gnn_amd import RW_NN
    adj, feats, gi, y = batch_t
    torch.manual_seed(0)
    model = RW_NN(hp["input_dim"], hp["max_step"], hp["hidden_graphs"], hp["size_hidden_graphs"], hp["hidden_dim"],
                  hp["penultimate_dim"], hp["normalize"], hp["n_classes"], hp["dropout"], dev).to(dev)
    model.train()
    opt = (gnn_amd.Adam if one_launch_adam else torch.optim.Adam)(model.parameters(), lr=1e-2)

    def train():
        opt.zero_grad()
        out = model(adj, feats, gi)
        loss = F.cross_entropy(out, y)
        loss.backward()
        opt.step()

    for _ in range(warmup):
        train()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        train()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    what = ("rwgnn2/main.py:82-88 unchanged after the import swap: eager launches, F.cross_entropy, "
            "torch.optim.Adam, per-parameter gradients (host-bound)")
    if one_launch_adam:
        what = ("the same loop with main.py:79's optim.Adam replaced by gnn_amd.Adam (a torch.optim.Optimizer whose "
                "step is one launch)")
    return {"ms_per_step": round(ms, 4), "graphs_per_s": round(int(y.numel()) / ms * 1e3, 1), "what": what}


def _new_model(hp, dev, seed=1234):
    from gnn_amd import RW_NN
    torch.manual_seed(0)
    model = RW_NN(hp["input_dim"], hp["max_step"], hp["hidden_graphs"], hp["size_hidden_graphs"], hp["hidden_dim"],
                  hp["penultimate_dim"], hp["normalize"], hp["n_classes"], hp["dropout"], dev).to(dev)
    model.train()
    model.seed_dropout(seed)
    return model


def _timed(run, steps):
    """Wall-clock and device time of ``run(steps)`` between two synchronisations."""
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    run(steps)
    ev1.record()
    while not ev1.query():
        pass
    torch.cuda.synchronize()
    return time.perf_counter() - t0, ev0.elapsed_time(ev1)


def distinct_batches_epoch(config, hp, dev, k, epochs=8, seed0=100):
    """The reference's epoch loop (rwgnn2/main.py:103-106) over ``k`` DIFFERENT resident batches of the configuration,
    replayed as one HIP graph per epoch (gnn_amd.trainer.EpochTrainer): every step meets a batch the caches have not
    seen for k - 1 steps."""
    from gnn_amd.synth import config_batch
    from gnn_amd.trainer import EpochTrainer
    batches = [config_batch(config, seed=seed0 + i).to_torch(dev) for i in range(k)]
    tr = EpochTrainer(_new_model(hp, dev), batches)
    tr.capture_epoch()
    tr.run_epoch()
    elapsed, dev_ms = _timed(lambda n: [tr.run_epoch() for _ in range(n)], epochs)
    graphs = sum(int(b[3].numel()) for b in batches)
    nbytes = sum(b[1].numel() * 4 + b[0]._values().numel() * 20 + b[2].numel() * 8 for b in batches)
    return {"batches": k, "epochs": epochs, "ms_per_step": round(elapsed / (epochs * k) * 1e3, 4),
            "device_ms_per_step": round(dev_ms / (epochs * k), 4),
            "graphs_per_s": round(graphs * epochs / elapsed, 1), "input_MB": round(nbytes / 1e6, 1),
            "final_loss": round(float(tr.epoch_losses[-1]), 5),
            "what": "EpochTrainer: one HIP-graph replay per epoch over %d different resident batches" % k}


class StreamBench:
    """A pool of distinct batches of a configuration as one device-resident data set, streamed batch by batch."""

    def __init__(self, config, hp, dev, pool_batches, seed, optimizer="flat", world=1, sync_before_all_reduce=False,
                 side_stream=None):
        from gnn_amd.data import DeviceDataset
        from gnn_amd.synth import config_batch
        from gnn_amd.trainer import StreamTrainer
        whole = config_batch(config, seed=seed, scale=pool_batches)
        self.per_batch = whole.n_graphs // pool_batches
        self.ds = DeviceDataset.from_batch(whole, dev)
        self.id_lists = [np.arange(i * self.per_batch, (i + 1) * self.per_batch) for i in range(pool_batches)]
        shapes = [self.ds.batch_shape(ids) for ids in self.id_lists]
        self.nodes = sum(s[0] for s in shapes)
        self.tr = StreamTrainer(_new_model(hp, dev, seed), self.ds, self.per_batch, max(s[0] for s in shapes),
                                max(s[1] for s in shapes), depth=3, side_stream=side_stream, world=world,
                                optimizer=optimizer,
                                sync_before_all_reduce=sync_before_all_reduce)
        self.cursor = 0

    def run(self, steps):
        """``steps`` training steps, each on the next batch of the pool (whole passes, then the rest)."""
        pool = len(self.id_lists)
        while steps > 0:
            take = min(steps, pool - self.cursor)
            self.tr.run_pass(self.id_lists[self.cursor:self.cursor + take], keep_losses=False)
            self.cursor = (self.cursor + take) % pool
            steps -= take

    def describe(self):
        return ("pool of %d distinct batches x %d graphs (%d nodes) resident as one data set; every batch assembled "
                "inside the timed region (one upload per pass + one launch per batch %s, structure "
                "included), steps launched eagerly%s" % (
                    len(self.id_lists), self.per_batch, self.nodes,
                    "on a side stream" if self.tr._side is not None else "on the step's stream",
                    ", a whole pass as one library call on the slots' buffers" if self.tr.raw_steps else ""))


def stream_extra(config, dev, pool_batches=31, steps=62, warmup=31):
    """The streamed pass of a configuration as an extra key of the default line: one whole pass over the pool untimed
    (every slot and workspace has met its largest batch, the clocks have settled), then whole passes timed."""
    from gnn_amd.synth import CONFIGS
    hp = dict(CONFIGS[config]["hp"])
    sb = StreamBench(config, hp, dev, pool_batches, seed=4321)
    sb.run(warmup)
    # A pass that steps through autograd allocates thousands of Python objects per step; by now the process holds
    # everything the earlier measurements built, and every full collection walks all of it.  Park those objects in
    # the permanent generation for the timed region (what a training script gets by calling gc.freeze() after set-up).
    import gc
    gc.collect()
    gc.freeze()
    try:
        elapsed, dev_ms = _timed(sb.run, steps)
    finally:
        gc.unfreeze()
    return {"config": config, "steps": steps, "ms_per_step": round(elapsed / steps * 1e3, 4),
            "device_ms_per_step": round(dev_ms / steps, 4),
            "graphs_per_s": round(sb.per_batch * steps / elapsed, 1), "what": sb.describe()}


def eager_step_extra(batch_t, hp, dev, steps=200, warmup=50):
    """Trainer.step_eager on the resident batch: the whole step as one library call, no graph replay."""
    from gnn_amd.trainer import Trainer
    tr = Trainer(_new_model(hp, dev), batch_t)
    for _ in range(warmup):
        tr.step_eager()
    elapsed, dev_ms = _timed(lambda n: [tr.step_eager() for _ in range(n)], steps)
    return {"steps": steps, "ms_per_step": round(elapsed / steps * 1e3, 4),
            "device_ms_per_step": round(dev_ms / steps, 4),
            "graphs_per_s": round(int(batch_t[3].numel()) * steps / elapsed, 1),
            "one_call": bool(tr._train_step is not None and tr.tail is not None and tr.tail.head_folds > 0),
            "what": "the headline step launched eagerly, not replayed: forward, loss, backward and Adam in one "
                    "library call (rwk_model_train_step), the same eight launches"}


def eval_stream_extra(config, dev, pool_batches=31, passes=4):
    """model.eval()'s forward over the batches of a pool without assembling them: one launch per batch that reads its
    graphs from the device-resident data set (gnn_amd.eval.eval_forward_planned)."""
    from gnn_amd.data import DeviceDataset
    from gnn_amd.eval import eval_forward_planned
    from gnn_amd.synth import CONFIGS, config_batch
    hp = dict(CONFIGS[config]["hp"])
    whole = config_batch(config, seed=4321, scale=pool_batches)
    per = whole.n_graphs // pool_batches
    ds = DeviceDataset.from_batch(whole, dev)
    id_lists = [np.arange(i * per, (i + 1) * per) for i in range(pool_batches)]
    model = _new_model(hp, dev).eval()
    plan = ds.plan(id_lists)
    outs = [torch.empty(per, hp["n_classes"], device=dev) for _ in range(pool_batches)]

    def run(n):
        for _ in range(n):
            for i in range(pool_batches):
                if eval_forward_planned(model, ds, plan, i, outs[i]) is None:
                    raise RuntimeError("the evaluation launch does not serve this configuration")
    run(1)
    elapsed, dev_ms = _timed(run, passes)
    batches = passes * pool_batches
    return {"config": config, "batches": batches, "us_per_batch": round(elapsed / batches * 1e6, 2),
            "device_us_per_batch": round(dev_ms / batches * 1e3, 2), "graphs_per_s": round(per * batches / elapsed, 1),
            "what": "evaluation-mode forward of %d distinct batches x %d graphs read straight from the resident data "
                    "set: one launch per batch, no batch assembled" % (pool_batches, per)}


def cpu_baseline(batch, hp, n_graphs, budget_s=20.0):
    """Forward + backward (train mode, dropout on) of the PyTorch-CPU port on the first n_graphs graphs."""
    from oracle import rwnn_torch_port as P
    sub = batch.slice_graphs(0, min(n_graphs, batch.n_graphs))
    adj, feats, gi, y = sub.to_torch("cpu")
    adj = adj.coalesce()
    torch.manual_seed(0)
    h, s, d, steps = hp["hidden_graphs"], hp["size_hidden_graphs"], hp["hidden_dim"], hp["max_step"]
    lh = h * steps
    params = {
        "adj_hidden": torch.empty(h, s * (s - 1) // 2).uniform_(-1, 1),
        "features_hidden": torch.empty(h, s, d).uniform_(0, 1),
        "fc.weight": torch.randn(d, hp["input_dim"]) * 0.5, "fc.bias": torch.zeros(d),
        "bn.weight": torch.ones(lh), "bn.bias": torch.zeros(lh),
        "fc1.weight": torch.randn(hp["penultimate_dim"], lh) * 0.1, "fc1.bias": torch.zeros(hp["penultimate_dim"]),
        "fc2.weight": torch.randn(hp["n_classes"], hp["penultimate_dim"]) * 0.1,
        "fc2.bias": torch.zeros(hp["n_classes"]),
    }
    for v in params.values():
        v.requires_grad_(True)

    def step():
        for v in params.values():
            v.grad = None
        out, _ = P.forward(params, adj, feats, gi, steps, hp["normalize"], hp["dropout"], True)
        F.nll_loss(out, y).backward()

    def timed(threads, max_steps, budget):
        torch.set_num_threads(threads)
        step()                                            # warm-up at this thread count
        times = []
        t_start = time.perf_counter()
        while len(times) < max_steps and (time.perf_counter() - t_start) < budget:
            t0 = time.perf_counter()
            step()
            times.append(time.perf_counter() - t0)
        times.sort()
        return times[len(times) // 2], len(times)

    # cores this process may run on (the box's CPU share), not the machine's core count
    try:
        share = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        share = os.cpu_count() or 1
    env_cap = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    threads = max(1, min(share, env_cap) if env_cap > 0 else min(share, 16))
    med, n_steps = timed(threads, 20, budget_s * 0.6)
    med1, n_steps1 = timed(1, 5, budget_s * 0.4)
    torch.set_num_threads(threads)
    return {"value": round(sub.n_graphs / med, 1), "unit": "graphs/s", "cores": threads, "kind": "port",
            "sample": f"first {sub.n_graphs} graphs ({sub.n_nodes} nodes, {sub.nnz} nnz) of the same batch, "
                      f"fwd+bwd train mode, median of {n_steps} steps on {threads} threads (the process may run on "
                      f"{share} of the host's {os.cpu_count()} logical CPUs)",
            "ms_per_step": round(med * 1e3, 2),
            "single_thread": {"value": round(sub.n_graphs / med1, 1), "unit": "graphs/s", "cores": 1,
                              "ms_per_step": round(med1 * 1e3, 2), "steps": n_steps1}}


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    # RWK_DIST_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks (ranks share devices,
    # the all-reduce goes through the host); the measured runs use nccl (= RCCL), one rank per GPU.
    import faulthandler
    if os.environ.get("RWK_BENCH_TRACE"):            # hang diagnosis: dump every thread's stack after N seconds
        os.makedirs("gpurun_out", exist_ok=True)
        faulthandler.dump_traceback_later(int(os.environ["RWK_BENCH_TRACE"]),
                                          file=open("gpurun_out/trace_rank%d.txt" % rank, "w"), exit=True)
    else:
        # a run still going after fifteen minutes writes every thread's stack to stderr once and carries on: a rank
        # stuck in a collective or a synchronize then names the call it sits in
        faulthandler.dump_traceback_later(900, exit=False)
    backend = os.environ.get("RWK_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # RWK_REHEARSE_RCCL=1 (with --dp-shape 2): a ONE-rank process group on the real collective library, and every
    # collective call of the multi-rank path issued on it - what a box with a single GPU can check of that path
    # (initialisation, the averaging all-reduce on the launch stream, its capture, the barriers, the shutdown).
    rehearse = world == 1 and os.environ.get("RWK_REHEARSE_RCCL", "0") == "1"
    dist_on = world > 1 or rehearse
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from gnn_amd import RW_NN
    from gnn_amd.dp import broadcast_parameters
    from gnn_amd.synth import CONFIGS, config_batch

    if args.config not in CONFIGS:
        raise SystemExit("unknown --config %r; choose from %s" % (args.config, ", ".join(CONFIGS)))
    cfg = CONFIGS[args.config]
    hp = dict(cfg["hp"])
    if args.dropout is not None:
        hp["dropout"] = args.dropout
    if args.scaling == "strong" and world > 1:
        from gnn_amd.dp import shard_ranges
        whole = config_batch(args.config, seed=1, scale=args.scale)         # the same batch on every rank ...
        lo, hi = shard_ranges(whole.sizes, world)[rank]                     # ... cut by node count
        batch = whole.slice_graphs(lo, hi)
    else:
        batch = config_batch(args.config, seed=1 + rank, scale=args.scale)  # each rank its own graphs
    batch_t = batch.to_torch(dev)

    if args.kernels_only:
        only = [k for k in args.only.split(",") if k] or None
        kern = time_kernels(tile_batch(batch_t, args.roofline_scale), hp, only=only)
        if args.calibrate:
            a = torch.zeros(1 << 28, device=dev)
            b = torch.empty_like(a)
            for _ in range(3):
                torch.add(a, 1.0, out=b)
            torch.cuda.synchronize()
        traffic = load_traffic(args.config, args.roofline_scale)
        shape = kern.pop("_shape")
        print(json.dumps({"config": args.config, "roofline_scale": args.roofline_scale, "shape": shape,
                          "dropout": hp["dropout"], "kernels": kern,
                          "roofline": [roofline_of(kern, k, traffic.get(k, {}).get("hbm_bytes")) for k in kern]}),
              flush=True)
        return

    if batch.n_graphs < 2:
        raise SystemExit("%s is a batch of one graph: BatchNorm1d cannot train on it (model.py:68 raises in the "
                         "reference too); run it with --kernels-only" % args.config)
    shape_world = args.dp_shape if (world == 1 and args.dp_shape > 1) else world
    # RCCL runs capture the gradient all-reduce inside the step's graph (one replay per step, several steps per
    # replay, nothing on the host in between: 0.099 against 0.110 ms per step in the one-rank rehearsal on RCCL); a
    # capture the runtime refuses, or one whose replay leaves different gradients on the ranks, falls back to
    # graph + eager all-reduce + Adam launch.  --no-capture-allreduce / RWK_CAPTURE_ALLREDUCE=0: that shape at once.
    capture_collective = (backend == "nccl" and not args.no_capture_allreduce
                          and os.environ.get("RWK_CAPTURE_ALLREDUCE", "1") != "0")
    launch = args.mode
    run_steps = None
    stream_bench = None
    graphs_per_step = batch.n_graphs
    if args.stream:
        # never-repeating batches: assembly inside the timed region, eager steps (StreamTrainer)
        stream_bench = StreamBench(args.config, hp, dev, args.pool_batches, seed=1 + rank, optimizer=args.optimizer,
                                   world=shape_world, sync_before_all_reduce=backend != "nccl")
        model, tr = stream_bench.tr.model, stream_bench.tr
        broadcast_parameters(model)
        run_steps = stream_bench.run
        graphs_per_step = stream_bench.per_batch
        launch = "eager, streamed: " + stream_bench.describe()
    elif args.distinct_batches > 1:
        if world > 1:
            raise SystemExit("--distinct-batches is a single-process measurement")
        from gnn_amd.trainer import EpochTrainer
        k = args.distinct_batches
        batches = [batch_t] + [config_batch(args.config, seed=100 + i, scale=args.scale).to_torch(dev)
                               for i in range(1, k)]
        model = _new_model(hp, dev, 1234)
        tr = EpochTrainer(model, batches, optimizer=args.optimizer)
        tr.capture_epoch()
        args.steps = max(k, args.steps // k * k)           # whole epochs: one replay each
        args.warmup = max(k, args.warmup // k * k)
        run_steps = lambda n: [tr.run_epoch() for _ in range(n // k)]   # noqa: E731
        graphs_per_step = sum(int(b[3].numel()) for b in batches) / k
        launch = "graph (one replay per epoch of %d different resident batches, EpochTrainer)" % k
    else:
        model = _new_model(hp, dev, 1234 + rank)
        broadcast_parameters(model)
        tr = Trainer(model, batch_t, shape_world, args.optimizer, sync_before_all_reduce=backend != "nccl",
                     capture_collective=capture_collective and dist_on)
        if args.mode == "graph":
            try:
                multi_ok = shape_world == 1 or (capture_collective and dist_on)
                spg = args.steps_per_graph if (multi_ok and args.steps_per_graph > 1) else 1
                tr.capture(steps_per_graph=spg)
                step = tr.step_graph
                if tr.multi_k > 1:
                    run_steps = tr.run_steps
                    launch = "graph (%d steps per replay)" % tr.multi_k
            except Exception as exc:        # a capture that the runtime refuses must not cost the whole measurement
                print(f"[bench] HIP-graph capture failed ({type(exc).__name__}: {exc}); launching eagerly",
                      file=sys.stderr, flush=True)
                torch.cuda.synchronize()
                launch = "eager (capture failed)"
                step = tr.step_eager
        else:
            step = tr.step_eager

    # The per-kernel timings of the roofline entry run before the timed steps, not after: they keep the device busy
    # for some 25 ms, and the step needs ~10 ms of load to reach its steady duration (the first 80 replays after an
    # idle period were measured up to 4 % slower, which a 20-step timed region would otherwise report as the step
    # time).
    kern = None
    if rank == 0 and not args.no_kernel_timing:
        kern = time_kernels(batch_t, hp, in_step=True)
    if run_steps is not None:
        run_steps(args.warmup)
    else:
        for _ in range(args.warmup):
            step()
    if dist_on:
        dist.barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    if run_steps is not None:
        run_steps(args.steps)
    else:
        for _ in range(args.steps):
            step()
    ev1.record()
    # the host polls for the end of the last step (a blocking wait is woken tens of microseconds late, which a short
    # timed region would carry as step time); the synchronize below then returns at once
    while not ev1.query():
        pass
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    device_ms = ev0.elapsed_time(ev1)
    n_graphs_total = graphs_per_step
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([graphs_per_step], dtype=torch.float64, device=dev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        n_graphs_total = float(c.item())
    loss = float(tr.loss.item())
    # where a data-parallel step spends its time (every rank runs it: the collective needs them all)
    phases = tr.phase_times() if (launch == "graph" and shape_world > 1) else None

    if rank == 0:
        result = {
            "metric": "train graphs/sec (fwd+bwd)",
            "value": round(n_graphs_total * args.steps / elapsed, 1),
            "unit": "graphs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": args.scaling if world > 1 else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.config}: {batch.n_graphs} graphs/GPU, {batch.n_nodes} nodes, {batch.nnz} nnz "
                            f"(rank 0), H{hp['hidden_graphs']} S{hp['size_hidden_graphs']} D{hp['hidden_dim']} "
                            f"L{hp['max_step']} dropout {hp['dropout']}",
                "step": "zero_grad + forward + cross_entropy + backward + grad all-reduce + Adam (%s)" % args.optimizer,
                "launch": launch,
                "parallelism": f"dp{world} (graphs sharded, one flat-bucket all-reduce)" + (
                    "; one batch cut into node-balanced graph ranges" if (args.scaling == "strong" and world > 1) else ""),
            },
            "final_loss": round(loss, 5),
            # the same K steps between two events on the launch stream (rank 0): what the host clock adds
            "device_ms_per_step": round(device_ms / args.steps, 4),
        }
        if shape_world != world:
            result["config"]["launch"] += f"; step in the shape of a {shape_world}-rank run (%s)" % (
                "one-rank %s process group, every collective issued" % backend if rehearse else "no process group")
        if phases is not None:
            result["phases"] = phases
        if tr.capture_note:
            result["config"]["launch"] += "; " + tr.capture_note
        if kern is not None:
            result["kernel_shape"] = kern.pop("_shape")
            # the dominant kernel, in the form the step launches it when that form was timed
            step_forms = {k: v for k, v in kern.items() if k != "fused_backward" or "fused_backward_in_step" not in kern}
            dom = max(step_forms, key=lambda k: step_forms[k]["us"])
            traffic = load_traffic(args.config, 1)
            result["kernels"] = kern
            result["roofline"] = roofline_of(kern, dom, traffic.get(dom.replace("_in_step", ""), {}).get("hbm_bytes"))
            if "launch" in kern[dom]:
                result["roofline"]["launch"] = kern[dom]["launch"]
        if world == 1 and not args.no_kernel_timing:
            result["drop_in_eager"] = drop_in_step(batch_t, hp, dev)
            result["drop_in_eager_one_launch_adam"] = drop_in_step(batch_t, hp, dev, one_launch_adam=True)
        plain = not (args.stream or args.distinct_batches > 1 or args.no_extras or args.no_kernel_timing)
        if world == 1 and plain and args.config == "proteins1024":
            # what no resident-batch figure covers: batches that change every step
            try:
                result["distinct_batches"] = distinct_batches_epoch(args.config, hp, dev, 16)
                result["stream_h64_stream"] = stream_extra("stream_h64", dev, steps=124, warmup=93)
                result["stream_proteins1024"] = stream_extra("proteins1024", dev, steps=372, warmup=279)
                result["eager_one_call_step"] = eager_step_extra(batch_t, hp, dev)
                result["eval_from_data_set"] = eval_stream_extra("proteins1024", dev)
            except Exception as exc:                      # the extras never cost the headline
                result["extras_error"] = "%s: %s" % (type(exc).__name__, str(exc)[:300])
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline(batch, hp, args.cpu_sample_graphs)
        print(json.dumps(result), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
