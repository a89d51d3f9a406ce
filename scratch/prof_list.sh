#!/bin/bash
# usage on the GPU box: bash scratch/prof_list.sh  -> per-kernel average durations of an eager training run
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf gpurun_out/ph
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ph -- python3 bench.py --mode eager --steps 30 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/ph.log 2>&1 || { tail -5 gpurun_out/ph.log; exit 1; }
python3 - <<PY
import csv, glob, os
f = max(glob.glob("gpurun_out/ph/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
tot = 0
for r in csv.DictReader(open(f)):
    c = int(r["Calls"])
    if c >= 30:
        per = float(r["AverageNs"]) / 1e3 * (c // 35 if c % 35 == 0 else 1)
        name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        print("%-58s %4d %6.1f" % (name[:58], c, float(r["AverageNs"]) / 1e3))
PY
