#!/bin/bash
# usage on the GPU box: bash scratch/prof_cfg.sh <bench config>  -> per-kernel average durations of an eager training run
cfg=${1:-stream_h64}
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf gpurun_out/pc
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pc -- python3 bench.py --config $cfg --mode eager --steps 30 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/pc.log 2>&1 || { tail -5 gpurun_out/pc.log; exit 1; }
python3 - <<PY
import csv, glob, os
f = max(glob.glob("gpurun_out/pc/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
for r in csv.DictReader(open(f)):
    c = int(r["Calls"])
    if c >= 30:
        name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        print("%-70s %4d %7.1f" % (name[:70], c, float(r["AverageNs"]) / 1e3))
PY
