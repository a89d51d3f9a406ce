#!/bin/bash
# usage: bash scratch/prof_step.sh <tag>   -> per-kernel time of the graph-replayed training step
tag=$1
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
steps=113
tot=0
for r in rows:
    if int(r['Calls'])<steps: continue
    n=r['Name']; short=n.split('anonymous namespace)::')[-1].split('(')[0][:46] if 'anonymous namespace)::' in n else n.split('(')[0][-46:]
    per=float(r['TotalDurationNs'])/steps/1e3; tot+=per
    print(f"{short:48s} x{int(r['Calls'])/steps:4.1f} avg {float(r['AverageNs'])/1e3:6.2f}  step {per:6.2f}")
print('sum per step %.1f us' % tot)
PY
