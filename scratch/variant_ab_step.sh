#!/bin/bash
# bash scratch/variant_ab_step.sh <tag> <variants...>: same-box A/B of library variants on the replayed training step
# (default bench line without the CPU baseline), four alternating rounds
set -u
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
for r in 1 2 3 4; do
  for v in "$@"; do
    RWK_VARIANT_LIB=scratch/librwk_$v.so timeout -k 10 200 python3 -u scratch/variant_bench.py --no-cpu-baseline --steps 400 > $out/${v}_step_$r.json 2> $out/${v}.err || { tail -5 $out/${v}.err; exit 1; }
    python3 -c "
import json
d=json.load(open('$out/${v}_step_$r.json'))
print('%-6s round $r' % '$v', 'ms_per_step', d['ms_per_step'], 'device', d['device_ms_per_step'])"
  done
done
