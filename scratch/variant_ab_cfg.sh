#!/bin/bash
# bash scratch/variant_ab_cfg.sh <tag> <config> <variants...>: same-box A/B of library variants on a configuration's replayed step
set -u
tag=$1; cfg=$2; shift 2
out=gpurun_out/$tag; mkdir -p $out
for r in 1 2 3; do
  for v in "$@"; do
    RWK_VARIANT_LIB=scratch/librwk_$v.so timeout -k 10 200 python3 -u scratch/variant_bench.py --config $cfg --no-cpu-baseline --no-kernel-timing --steps 300 > $out/${v}_${cfg}_$r.json 2> $out/${v}.err || { tail -5 $out/${v}.err; exit 1; }
    python3 -c "
import json
d=json.load(open('$out/${v}_${cfg}_$r.json'))
print('%-6s %-10s round $r' % ('$v', '$cfg'), 'ms_per_step', d['ms_per_step'], 'loss', d['final_loss'])"
  done
done
