"""CPU-side checks: the C-ABI library loads and exports what include/rwk.h declares, the module mirrors the
reference's interface, and the product path refuses to run without a HIP device (no CPU fallback)."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "rwk.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(rwk_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    from gnn_amd import _lib
    lib = _lib.load()
    declared = _declared_functions()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(_lib.SIGNATURES) == declared
    assert lib.rwk_version() == 100
    assert lib.rwk_error_string(-3).decode().startswith("hyper-parameters")


def test_header_argument_counts_match_binding():
    from gnn_amd import _lib
    text = open(os.path.join(ROOT, "include", "rwk.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    for name, (_, argtypes) in _lib.SIGNATURES.items():
        m = re.search(r"\b%s\s*\(([^)]*)\)" % name, text)
        assert m, name
        args = m.group(1).strip()
        n_args = 0 if args in ("", "void") else len(args.split(","))
        assert n_args == len(argtypes), name


def test_workspace_size_functions_are_pure_host_code():
    from gnn_amd import _lib
    lib = _lib.load()
    assert lib.rwk_fused_n_streams(1000, 16) == 63
    assert lib.rwk_fused_k_partials_size(1000, 16, 10, 16, 2) == (63 + 10) * 32
    # 16 hidden graphs -> 16 lanes per stream, 256-thread workgroups -> 16 streams per workgroup
    assert lib.rwk_fused_dz_partials_count(1000, 16, 4, 16, 5, 2) == 4
    assert lib.rwk_fused_dz_partials_count(1000, 16, 4, 65, 5, 2) == 0     # unsupported: H > 64


def test_module_mirrors_reference_interface():
    from gnn_amd import RW_NN
    torch.manual_seed(0)
    m = RW_NN(1, 2, 16, 5, 4, 32, False, 2, 0.2, "cpu")
    keys = list(m.state_dict().keys())
    assert keys == ["adj_hidden", "features_hidden", "fc.weight", "fc.bias", "bn.weight", "bn.bias",
                    "bn.running_mean", "bn.running_var", "bn.num_batches_tracked", "fc1.weight", "fc1.bias",
                    "fc2.weight", "fc2.bias"]
    assert m.adj_hidden.shape == (16, 10) and m.features_hidden.shape == (16, 5, 4)
    assert float(m.adj_hidden.min()) >= -1 and float(m.adj_hidden.max()) <= 1
    assert float(m.features_hidden.min()) >= 0 and float(m.features_hidden.max()) <= 1
    assert m.max_step == 2 and m.hidden_graphs == 16 and m.size_hidden_graphs == 5 and m.normalize is False


def test_same_initial_weights_as_reference_rule():
    """Construction consumes the global RNG in the reference's order (model.py:16-29): parameters are
    created uninitialised, the Linear/BatchNorm modules draw theirs, then adj_hidden and features_hidden."""
    from gnn_amd import RW_NN
    torch.manual_seed(3)
    m = RW_NN(3, 2, 4, 3, 5, 6, True, 2, 0.0, "cpu")
    torch.manual_seed(3)
    fc = torch.nn.Linear(3, 5)
    torch.nn.BatchNorm1d(8)
    fc1 = torch.nn.Linear(8, 6)
    fc2 = torch.nn.Linear(6, 2)
    adj_hidden = torch.empty(4, 3).uniform_(-1, 1)
    feat_hidden = torch.empty(4, 3, 5).uniform_(0, 1)
    assert torch.equal(m.fc.weight, fc.weight) and torch.equal(m.fc1.weight, fc1.weight)
    assert torch.equal(m.fc2.bias, fc2.bias)
    assert torch.equal(m.adj_hidden, adj_hidden) and torch.equal(m.features_hidden, feat_hidden)


def test_no_cpu_fallback():
    from gnn_amd import RW_NN
    m = RW_NN(1, 2, 4, 3, 4, 8, False, 2, 0.0, "cpu")
    idx = torch.tensor([[0, 1], [1, 0]])
    adj = torch.sparse_coo_tensor(idx, torch.ones(2), (3, 3))
    with pytest.raises(RuntimeError, match="HIP devices only"):
        m(adj, torch.ones(3, 1), torch.tensor([0, 0, 1]))


def test_spmm_schedule_rule():
    from gnn_amd.structure import spmm_schedule
    assert spmm_schedule(1168, 2590, 4) == (1, 0)
    assert spmm_schedule(3000, 7000, 900) == (1, 128)
    lanes, thr = spmm_schedule(1000, 64000, 80)
    assert lanes == 16 and thr == 0
    from gnn_amd.structure import use_tile_kernel
    assert use_tile_kernel(307842, 1132004, 12, 4)            # protein-like batch, hidden_dim 4
    assert use_tile_kernel(190013, 622914, 1900, 64)          # hubs: always
    assert not use_tile_kernel(102962, 217050, 8, 64)         # 35 entries per 16-row workgroup
    assert use_tile_kernel(102962, 217050, 8, 16)
    assert not use_tile_kernel(1000, 64000, 80, 4)            # dense rows


def test_stream_length_rule():
    from gnn_amd.function import choose_stream_len
    assert choose_stream_len(1168, 4, 16, 5, 2) == 8
    assert choose_stream_len(307842, 4, 16, 5, 2) == 19          # 1013 workgroups: one round of the 1024 slots
    assert choose_stream_len(307842, 4, 16, 5, 2, backward=True) == 26   # 3 resident workgroups per CU
    assert choose_stream_len(50_000_000, 4, 16, 5, 2) == 20
    assert choose_stream_len(103008, 4, 64, 5, 2) == 26
    # matrix-core kernels: 248 forward workgroups x 4 blocks of 16 hidden graphs
    assert choose_stream_len(103008, 64, 64, 10, 2) == 13
    # backward: 16 streams per workgroup, all blocks of hidden graphs in one workgroup: one round of 252 workgroups
    assert choose_stream_len(103008, 64, 64, 10, 2, backward=True) == 26
    assert choose_stream_len(16 * 103008, 64, 64, 10, 2, backward=True) == 31       # 13 full rounds
    assert choose_stream_len(0, 4, 16, 5, 2) == 0

def test_rwconv_oracle_against_dense_matrix():
    """oracle/rwconv_oracle.py (edge-list form) against the dense matrix M[i, j] = #edges(j -> i) / deg[j] it
    restates, forward and VJP, on a graph with duplicate edges, self-loops and an isolated node."""
    from oracle import rwconv_oracle as R
    rng = np.random.default_rng(0)
    n = 12
    ei = np.vstack([rng.integers(0, n - 1, 30), rng.integers(0, n - 1, 30)])
    ei = np.concatenate([ei, ei[:, :4], np.array([[2, 2], [2, 2]])], axis=1)
    x, w, b = rng.normal(size=(n, 3)), rng.normal(size=(2, 3)), rng.normal(size=2)
    d = rng.normal(size=(n, 2))
    src = np.concatenate([ei[0], np.arange(n)])
    dst = np.concatenate([ei[1], np.arange(n)])
    deg = np.bincount(src, minlength=n)
    m = np.zeros((n, n))
    for s_, t_ in zip(src, dst):
        m[t_, s_] += 1.0 / deg[s_]
    out, agg = R.rwconv_forward(x, ei, w, b)
    assert np.abs(out - (m @ x @ w.T + b)).max() < 1e-12 and np.abs(agg - m @ x).max() < 1e-12
    assert np.abs(m.sum(0) - 1.0).max() < 1e-12          # every node hands out exactly its own value
    dx, dw, db = R.rwconv_backward(x, ei, w, d)
    assert np.abs(dx - m.T @ (d @ w)).max() < 1e-12
    assert np.abs(dw - d.T @ (m @ x)).max() < 1e-12 and np.abs(db - d.sum(0)).max() < 1e-12


def test_distance_bias_oracle_against_scipy_dijkstra():
    """The oracle's hop distances (dense powers, rwgnn2/rwgnn_dis_bias6.py:109-128) equal scipy's dijkstra with a
    limit and inf -> max_distance + 1 (:130-154), the reference's other way to the same matrix; rows of the
    transition matrix sum to one (:239-244)."""
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import dijkstra
    from oracle import distance_bias_oracle as o
    rng = np.random.default_rng(0)
    for n, p, md in [(30, 0.08, 3), (50, 0.05, 4), (12, 0.3, 2), (1, 0.0, 3), (40, 0.02, 3)]:
        a = (rng.random((n, n)) < p).astype(np.float32)
        a = np.triu(a, 1)
        a = a + a.T
        d = o.hop_distances(a, md)
        ref = dijkstra(csr_matrix(a), directed=False, limit=md, unweighted=True)
        ref[ref == np.inf] = md + 1
        assert np.array_equal(d, ref.astype(np.float32)), (n, md)
        for walk_type, alpha, c in [("exp", 1.0, 0.1), ("levy", 2.0, 0.25)]:
            pm = o.walk_probabilities(d, walk_type, alpha, c)
            if n > 1:
                assert np.allclose(pm.sum(1), 1.0)
            assert np.allclose(np.diag(pm), c / n)


def test_pass_and_gather_entry_points_reject_bad_arguments_before_any_launch():
    """The entry points of a streamed pass check their arguments on the host: null pointers and sizes out of range
    come back as RWK_E_NULL / RWK_E_SIZE without a GPU being touched, and the plan call - host code only - serves
    nothing when the first batch is outside what the one-call step takes."""
    import ctypes
    from gnn_amd import _lib
    from gnn_amd.composite import ModelStep, PassBatch, PassDataset, PassNeed, PassSlot
    lib = _lib.load()
    E_NULL, E_SIZE = -1, -2
    ds, need, model = PassDataset(), PassNeed(), ModelStep()
    batches, slots = (PassBatch * 2)(), (PassSlot * 3)()
    args = [ctypes.byref(ds), batches, 2, slots, 3, ctypes.byref(model), ctypes.byref(need)]
    assert lib.rwk_model_train_pass_plan(*args) == 0 and need.served == 0        # b = 0: nothing served
    assert lib.rwk_model_train_pass_plan(None, *args[1:]) == E_NULL
    assert lib.rwk_model_train_pass_plan(*args[:4], 0, *args[5:]) == E_SIZE      # no slots
    one = ctypes.c_float()
    buf = ctypes.create_string_buffer(256)
    p = ctypes.addressof(buf)
    tail = ctypes.create_string_buffer(512)
    good = [ctypes.byref(ds), p, batches, 2, slots, 3, ctypes.byref(model), ctypes.addressof(tail), ctypes.byref(need),
            p, p, p, p, ctypes.addressof(one), p, None, None, 0]
    for k in (0, 1, 2, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14):
        bad = list(good)
        bad[k] = None
        assert lib.rwk_model_train_pass(*bad) == E_NULL, k
    for k, v in ((3, 0), (5, 0), (5, 9), (17, 3), (17, -1)):
        bad = list(good)
        bad[k] = v
        assert lib.rwk_model_train_pass(*bad) == E_SIZE, (k, v)
    assert lib.rwk_x_chain_hidden_forward_gather(None, p, p, 10, 1, 2, 8, 2, p, p, p, 4, 3, p, p, None, None,
                                                 None) == E_NULL
    assert lib.rwk_eval_forward_gather(None, p, p, p, 1, p, p, p, p, 2, 8, 4, 4, 3, 2, 0, p, p, p, p, 1e-5, p, p, p, p,
                                       8, 2, None, p, None) == E_NULL
    assert lib.rwk_eval_forward_gather(p, p, p, p, 1, p, p, p, p, 0, 8, 4, 4, 3, 2, 0, p, p, p, p, 1e-5, p, p, p, p,
                                       8, 2, None, p, None) == E_SIZE
