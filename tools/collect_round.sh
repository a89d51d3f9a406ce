#!/bin/bash
# usage (here, after a gpurun call merged gpurun_out/<tag> written by scratch/collect_r02.sh): bash tools/collect_round.sh <tag>
# Copies the judged summaries of a measurement round from gpurun_out/<tag> (scratch) into profiles/<tag> (tracked)
# and rebuilds profiles/traffic.json, which bench.py reads for roofline.traffic.
set -e
tag=$1; src=gpurun_out/$tag; dst=profiles/$tag; mkdir -p $dst
cp $src/*.json $src/kernel_stats.csv $dst/
[ -f $src/gpu_tests.log ] && cp $src/gpu_tests.log $dst/
[ -f $src/launch_floor.txt ] && cp $src/launch_floor.txt $dst/
for d in $src/*_x*/; do
  d=${d%/}; b=$(basename $d)
  [ -f $d/kernels.json ] || continue
  mkdir -p $dst/$b
  key=$(echo $b | sed 's/_x\([0-9]*\)$/@\1/')
  python3 tools/pmc_traffic.py $d "$key" > $dst/$b/traffic.json
  cp $d/kernels.json $dst/$b/
  cp "$(find $d/trace -name '*kernel_stats.csv' | head -1)" $dst/$b/kernel_stats.csv
done
python3 scratch/step_gaps.py $(ls $src/trace/*/*kernel_trace.csv | head -1) > $dst/step_kernels.txt || true
python3 - "$dst" <<'PY'
import json, glob, sys
out = {}
for f in sorted(glob.glob(sys.argv[1] + '/*/traffic.json')):
    out.update(json.load(open(f)))
json.dump(out, open('profiles/traffic.json', 'w'), indent=1)
for k, v in out.items():
    for kk, r in v.items():
        if kk.startswith('_'):
            continue
        us = r.get('event_us', 0) or 1e-9
        print(f"{k:18s} {kk:15s} us {r.get('event_us', 0):9.1f} trace {r.get('trace_avg_us', 0):9.1f} "
              f"alg {r.get('algorithmic_bytes', 0) / 1e6:8.1f} MB hbm {r['hbm_bytes'] / 1e6:8.1f} MB "
              f"alg GB/s {r.get('algorithmic_bytes', 0) / us / 1e3:6.0f}")
PY
