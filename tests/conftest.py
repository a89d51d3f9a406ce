import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_collection_finish(session):
    """Start the two-rank data-parallel job of tests/test_zy_gpu_dp2.py now: child processes are created before this
    process makes any GPU call; they import their modules and wait for the test's go-ahead, so they never share the
    device with other tests.  Nothing happens when that test is deselected
    (the CPU run) or no device is visible."""
    if session.config.option.collectonly:
        return
    if not any(os.path.basename(str(i.fspath)) == "test_zy_gpu_dp2.py" for i in session.items):
        return
    import socket
    import subprocess
    import tempfile
    import torch
    if torch.cuda.device_count() < 1:          # counting devices does not initialise the GPU
        return
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    outdir = tempfile.mkdtemp(prefix="rwk_dp2_")
    log = open(os.path.join(outdir, "log.txt"), "w")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    proc = subprocess.Popen(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
         "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dp2_worker.py"), outdir],
        stdout=log, stderr=subprocess.STDOUT, env=env, cwd=ROOT, start_new_session=True)
    session.config._dp2_job = (proc, outdir)


def pytest_sessionfinish(session, exitstatus):
    job = getattr(session.config, "_dp2_job", None)
    if job is not None and job[0].poll() is None:
        import signal
        os.killpg(job[0].pid, signal.SIGKILL)   # the process group this session started (launcher and its two ranks)
