"""Two real processes stepping the HIP kernels: each rank trains on its ``shard_ranges`` share of one batch, gradients
all-reduced over gloo (both ranks share the one GPU of the box).  The ranks are started by ``tests/conftest.py`` when
the collection ends - before this process makes its first GPU call - import what they need and wait; this test gives
them the go-ahead, waits for them and reads their verdicts (``tests/dp2_worker.py`` documents what each rank checks).
The file's name puts it near the end of the collection: processes that share one GPU have hung in the runtime on this
pool now and then (a rank that makes no progress dumps its stack into the job's log after four minutes and exits),
and under ``pytest -x`` a failure here should not hide the single-process tests."""
import json
import os

import pytest

pytestmark = pytest.mark.gpu


def test_two_ranks_on_the_hip_kernels_end_with_identical_parameters(request):
    job = getattr(request.config, "_dp2_job", None)
    if job is None:
        pytest.skip("no GPU (the two-rank job was not started)")
    proc, outdir = job
    import torch
    torch.cuda.synchronize()                    # this process leaves the device to the two ranks
    open(os.path.join(outdir, "go"), "w").close()
    try:
        rc = proc.wait(timeout=300)
    except Exception:
        import signal
        os.killpg(proc.pid, signal.SIGKILL)     # the launcher's own process group (conftest started it as one)
        raise
    log = open(os.path.join(outdir, "log.txt")).read()
    ranks = []
    for r in range(2):
        path = os.path.join(outdir, "rank%d.json" % r)
        assert os.path.exists(path), log[-3000:]
        ranks.append(json.load(open(path)))
    if any(r.get("skipped") for r in ranks):
        pytest.skip(ranks[0].get("skipped") or ranks[1].get("skipped"))
    for r in ranks:
        assert "error" not in r, r["error"]
        assert max(r["layer_grad_rel_err"].values()) < 1e-5, r
        assert r["params_bit_identical"] and r["finite"] and r["steps"] == 8, r
        assert r["ok"], r
    assert rc == 0, log[-3000:]
    assert ranks[0]["range"][1] == ranks[1]["range"][0] and ranks[0]["range"][0] == 0 and ranks[1]["range"][1] == 128
