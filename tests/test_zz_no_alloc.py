"""``rwk_reduce_rows_static`` (the entry point that allocates nothing) against ``rwk_reduce_rows`` with a caller-owned
workspace: same bits, on two calls in a row, the second with a smaller input (the library's fixed device scratch
keeps the first call's contents), the size limit, and a call recorded in a HIP graph.  Collected last, so it runs
after every other test of the suite in the same process; it also passes when this file is run on its own."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _both(lib, _lib, x, dev):
    p, v = x.shape
    stream = _lib.current_stream(dev)
    ws_n = lib.rwk_reduce_rows_workspace(p, v)
    ws = torch.full((max(ws_n, 1),), float("nan"), device=dev)
    want = torch.full((v,), float("nan"), device=dev)
    got = torch.full((v,), float("nan"), device=dev)
    _lib.check(lib.rwk_reduce_rows(x.data_ptr(), p, v, want.data_ptr(), ws.data_ptr(), stream), "rwk_reduce_rows")
    _lib.check(lib.rwk_reduce_rows_static(x.data_ptr(), p, v, got.data_ptr(), stream), "rwk_reduce_rows_static")
    torch.cuda.synchronize()
    return want, got, ws_n


def test_static_scratch_reduction_equals_workspace_reduction(dev):
    from gnn_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device="cpu").manual_seed(3)
    # (rows, columns): the first is the larger; both need the level-1 block (more than 32 rows, more than 64 columns)
    for p, v in ((1203, 1088), (97, 130)):
        x = torch.randn(p, v, generator=g).to(dev)
        want, got, ws_n = _both(lib, _lib, x, dev)
        assert ws_n == 32 * v
        assert torch.equal(want, got)
        ref = x.double().sum(0)
        assert float((got.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    # shapes whose kernels use no scratch (at most 32 rows; at most 64 columns: one launch) take the same path as
    # the workspace entry point
    for p, v in ((20, 300), (5000, 40)):
        x = torch.randn(p, v, generator=g).to(dev)
        want, got, _ = _both(lib, _lib, x, dev)
        assert torch.equal(want, got)


def test_static_scratch_limit_is_reported(dev):
    from gnn_amd import _lib
    lib = _lib.load()
    v = 16384 + 1
    x = torch.zeros(40, v, device=dev)
    out = torch.zeros(v, device=dev)
    rc = lib.rwk_reduce_rows_static(x.data_ptr(), 40, v, out.data_ptr(), _lib.current_stream(dev))
    assert rc == -5
    assert b"scratch" in lib.rwk_error_string(rc)
    # at the limit it runs
    x = torch.ones(40, 16384, device=dev)
    out = torch.zeros(16384, device=dev)
    _lib.check(lib.rwk_reduce_rows_static(x.data_ptr(), 40, 16384, out.data_ptr(), _lib.current_stream(dev)), "static")
    torch.cuda.synchronize()
    assert bool((out == 40.0).all())


def test_static_scratch_call_inside_a_captured_graph(dev):
    """After one call outside (the per-device address lookup), the entry point is recorded into a HIP graph and
    replayed on new contents of the same input buffer: same bits as the workspace entry point."""
    from gnn_amd import _lib
    lib = _lib.load()
    p, v = 300, 700
    x = torch.randn(p, v, device=dev)
    got = torch.zeros(v, device=dev)
    _lib.check(lib.rwk_reduce_rows_static(x.data_ptr(), p, v, got.data_ptr(), _lib.current_stream(dev)), "warm-up")
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        _lib.check(lib.rwk_reduce_rows_static(x.data_ptr(), p, v, got.data_ptr(), _lib.current_stream(dev)),
                   "rwk_reduce_rows_static under capture")
    for seed in (1, 2):
        x.copy_(torch.randn(p, v, generator=torch.Generator().manual_seed(seed)))
        g.replay()
        torch.cuda.synchronize()
        want, _, _ = _both(lib, _lib, x, dev)
        assert torch.equal(want, got)
