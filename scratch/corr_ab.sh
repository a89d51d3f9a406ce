#!/bin/bash
set -u
out=gpurun_out/$1; mkdir -p $out
true
true
for r in 1 2 3 4; do
  for v in 1 0; do
    RWK_NO_CORR_SLOT=$v timeout -k 10 200 python3 -u bench.py --no-cpu-baseline --steps 400 > $out/corr_${v}_${r}.json 2> $out/corr.err || { tail -5 $out/corr.err; exit 1; }
    python3 -c "
import json
d=json.load(open('$out/corr_${v}_${r}.json'))
print('no_corr_slot=$v round $r', 'ms_per_step', d['ms_per_step'], 'device', d['device_ms_per_step'], 'loss', d['final_loss'])"
  done
done
