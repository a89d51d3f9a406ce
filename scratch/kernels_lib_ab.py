"""Same-box A/B of the three timed kernels (bench.py --kernels-only) of two builds of the library:
python scratch/kernels_lib_ab.py <other library> [config] [scale].  Each build in a process of its own, twice."""
import json
import subprocess
import sys

other = sys.argv[1]
cfg = sys.argv[2] if len(sys.argv) > 2 else "proteins1024"
scale = sys.argv[3] if len(sys.argv) > 3 else "64"
code = ("import sys, json; sys.argv = ['bench.py', '--kernels-only', '--config', '%s', '--roofline-scale', '%s']\n"
        "from gnn_amd import _lib\n"
        "lib = sys.stdin.readline().strip()\n"
        "if lib: _lib.LIB_PATH = lib\n"
        "import bench; bench.main()\n") % (cfg, scale)
for rep in range(2):
    for name, lib in (("this build", ""), ("other build", other)):
        out = subprocess.run([sys.executable, "-c", code], input=lib + "\n", capture_output=True, text=True, cwd=".")
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(name, "failed", out.stderr[-500:])
            continue
        k = json.loads(line[-1])["kernels"]
        print(name, cfg, "x" + scale, {n: v["us"] for n, v in k.items() if not n.startswith("_")}, flush=True)
