#!/bin/bash
# usage on the GPU box: bash scratch/prof_stream_p.sh  -> per-kernel average durations of a streamed proteins1024 run
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf gpurun_out/ps
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ps -- python3 bench.py --config proteins1024 --stream --steps 62 --warmup 31 --no-cpu-baseline --no-extras --no-kernel-timing > gpurun_out/ps.log 2>&1 || { tail -5 gpurun_out/ps.log; exit 1; }
python3 - <<PY
import csv, glob, os
f = max(glob.glob("gpurun_out/ps/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
for r in csv.DictReader(open(f)):
    c = int(r["Calls"])
    if c >= 30:
        name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        print("%-70s %4d %7.1f" % (name[:70], c, float(r["AverageNs"]) / 1e3))
PY
head -c 400 gpurun_out/ps.log
