"""World-size-2 CPU (gloo) tests of the graph-sharded data-parallel path.

The HIP kernels cannot run here, so the layer is stood in for by the CPU oracle port (tests may use the oracle);
what is under test is gnn_amd.dp: the contiguous graph sharding, the flat gradient bucket and its single
all-reduce, and the identity "sum over ranks of per-shard gradients == gradient on the whole batch" for the
random-walk-kernel features (which are exactly additive over graphs).
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from gnn_amd.dp import FlatGradBucket, broadcast_parameters, shard_ranges  # noqa: E402
from gnn_amd.synth import random_batch  # noqa: E402


def test_shard_ranges_cover_and_balance():
    sizes = np.array([5, 1, 1, 1, 20, 3, 3, 3, 3, 10])
    for world in (1, 2, 3, 4, 10):
        r = shard_ranges(sizes, world)
        assert r[0][0] == 0 and r[-1][1] == sizes.size
        assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
        assert all(hi > lo for lo, hi in r)
    r = shard_ranges(np.ones(1024), 8)
    assert [hi - lo for lo, hi in r] == [128] * 8
    with pytest.raises(ValueError):
        shard_ranges([1, 2], 3)


def test_batch_slicing_renumbers_graphs():
    b = random_batch(12, 5, 9, 2.5, seed=3)
    s = b.slice_graphs(4, 9)
    assert s.n_graphs == 5 and s.graph_indicator.min() == 0 and s.graph_indicator.max() == 4
    assert s.n_nodes == int(b.sizes[4:9].sum())
    assert s.indices.min() >= 0 and s.indices.max() < s.n_nodes
    # an edge never leaves its graph
    assert np.array_equal(s.graph_indicator[s.indices[0]], s.graph_indicator[s.indices[1]])


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _layer_loss(params, batch, hp, weights):
    """sum_g w_g . K[g]  with K from the CPU port of the layer (no batch-norm: additive over graphs)."""
    from oracle import rwnn_torch_port as P
    adj, feats, gi, _ = batch.to_torch("cpu")
    k = P.rw_kernel_features(params, adj.coalesce(), feats, gi, hp["max_step"], hp["normalize"])
    return (k * weights).sum()


def _make_params(hp, seed):
    g = torch.Generator().manual_seed(seed)
    h, s, d = hp["hidden_graphs"], hp["size_hidden_graphs"], hp["hidden_dim"]
    return torch.nn.ParameterDict({
        "adj_hidden": torch.nn.Parameter(torch.empty(h, s * (s - 1) // 2).uniform_(-1, 1, generator=g)),
        "features_hidden": torch.nn.Parameter(torch.empty(h, s, d).uniform_(0, 1, generator=g)),
        "fc_weight": torch.nn.Parameter(torch.randn(d, hp["input_dim"], generator=g)),
        "fc_bias": torch.nn.Parameter(torch.randn(d, generator=g)),
    })


def _as_port_params(pd):
    return {"adj_hidden": pd["adj_hidden"], "features_hidden": pd["features_hidden"],
            "fc.weight": pd["fc_weight"], "fc.bias": pd["fc_bias"]}


HP = dict(input_dim=1, max_step=3, hidden_graphs=4, size_hidden_graphs=3, hidden_dim=4, normalize=True)


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    batch = random_batch(10, 4, 9, 2.5, seed=11)
    weights = torch.from_numpy(np.random.default_rng(5).normal(size=(10, HP["max_step"] * HP["hidden_graphs"])))
    weights = weights.float()
    # every rank starts from different parameters; rank 0's must win
    params = _make_params(HP, seed=100 + rank)
    broadcast_parameters(params)
    bucket = FlatGradBucket(params)
    lo, hi = shard_ranges(batch.sizes, world)[rank]
    bucket.zero()
    _layer_loss(_as_port_params(params), batch.slice_graphs(lo, hi), HP, weights[lo:hi]).backward()
    for p in params.values():                       # the gradients live in the flat buffer
        assert p.grad.data_ptr() >= bucket.flat.data_ptr()
    bucket.all_reduce(average=False)
    torch.save({"flat": bucket.flat.clone(), "params": {k: v.detach().clone() for k, v in params.items()},
                "range": (lo, hi)}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_gradients_equal_whole_batch_gradients(tmp_path, world):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(os.path.join(tmp_path, f"rank{r}.pt"), weights_only=True) for r in range(world)]
    # all ranks hold the same reduced bucket and rank 0's parameters
    for o in outs[1:]:
        assert torch.equal(o["flat"], outs[0]["flat"])
        for k in o["params"]:
            assert torch.equal(o["params"][k], outs[0]["params"][k])
    assert outs[0]["range"][0] == 0 and outs[-1]["range"][1] == 10
    # single-process gradient on the whole batch
    batch = random_batch(10, 4, 9, 2.5, seed=11)
    weights = torch.from_numpy(np.random.default_rng(5).normal(size=(10, HP["max_step"] * HP["hidden_graphs"]))).float()
    params = _make_params(HP, seed=100)
    bucket = FlatGradBucket(params)
    _layer_loss(_as_port_params(params), batch, HP, weights).backward()
    ref = bucket.flat
    assert torch.allclose(outs[0]["flat"], ref, rtol=1e-4, atol=1e-5 * float(ref.abs().max()))
