#!/bin/bash
# usage on the GPU box: bash scratch/prof_stream.sh [extra bench flags] -> per-kernel durations of a streamed stream_h64 run
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf gpurun_out/ps
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ps -- python3 bench.py --config stream_h64 --stream --steps 62 --warmup 8 --no-cpu-baseline --no-kernel-timing "$@" > gpurun_out/ps.log 2>&1 || { tail -5 gpurun_out/ps.log; exit 1; }
tail -1 gpurun_out/ps.log
python3 - <<PY
import csv, glob, os
f = max(glob.glob("gpurun_out/ps/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
tot = 0
for r in csv.DictReader(open(f)):
    c = int(r["Calls"])
    if c >= 60:
        name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        per = float(r["AverageNs"]) / 1e3 * c / 70
        tot += per
        print("%-70s %4d %7.1f  per step %7.1f" % (name[:70], c, float(r["AverageNs"]) / 1e3, per))
print("sum per step", round(tot, 1))
PY
