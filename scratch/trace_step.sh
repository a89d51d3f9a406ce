#!/bin/bash
# Usage on the GPU box: bash scratch/trace_step.sh <tag> [bench args]; kernel trace of the training step
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --mode eager --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing "$@" > $out/prof.log 2>&1 || { tail -5 $out/prof.log; exit 1; }
f=$(find $out/prof -name '*kernel_stats.csv' | head -1)
cp $f $out/kernel_stats.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$out/kernel_stats.csv')))
tot=0
for r in rows:
    calls=int(r['Calls']); avg=float(r['AverageNs'])/1e3
    if calls>=30:
        print('%-90s calls %4d avg %7.2f us' % (r['Name'][:90], calls, avg)); 
        tot+=avg*calls/35.0
print('sum per step (35 steps): %.1f us' % tot)
PY
