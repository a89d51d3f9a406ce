"""Same-box A/B of the whole training step of two builds of the library:
python scratch/step_lib_ab.py <other library> [bench.py flags...].  Each build in a process of its own, alternating,
three rounds; prints ms per step."""
import json
import subprocess
import sys

other = sys.argv[1]
flags = sys.argv[2:]
code = ("import sys; sys.argv = ['bench.py', '--no-cpu-baseline', '--no-extras', '--no-kernel-timing'] + %r\n"
        "from gnn_amd import _lib\n"
        "lib = sys.stdin.readline().strip()\n"
        "if lib: _lib.LIB_PATH = lib\n"
        "import bench; bench.main()\n") % (flags,)
for rep in range(3):
    for name, lib in (("this build", ""), ("other build", other)):
        out = subprocess.run([sys.executable, "-c", code], input=lib + "\n", capture_output=True, text=True, cwd=".")
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(name, "failed", out.stderr[-500:])
            continue
        print(name, json.loads(line[-1])["ms_per_step"], "ms", flush=True)
