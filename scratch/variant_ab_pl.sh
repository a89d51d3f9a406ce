#!/bin/bash
# bash scratch/variant_ab_pl.sh <tag> <variants...>: same-box A/B of library variants on the four-step configuration
set -u
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
for r in 1 2 3; do
  for cs in "powerlaw 32" "powerlaw 1" "proteins1024 64"; do
    set -- $cs "$@"; cfg=$1; sc=$2; shift 2
    for v in "$@"; do
      RWK_VARIANT_LIB=scratch/librwk_$v.so timeout -k 10 200 python3 -u scratch/variant_bench.py --kernels-only --config $cfg --roofline-scale $sc --only fused_forward,fused_backward > $out/${v}_${cfg}_x${sc}_$r.json 2> $out/${v}.err || { tail -5 $out/${v}.err; exit 1; }
      python3 -c "
import json
k=json.load(open('$out/${v}_${cfg}_x${sc}_$r.json'))['kernels']
print('%-6s %-12s x%-3s round $r' % ('$v', '$cfg', '$sc'), 'fwd %.1f bwd %.1f' % (k['fused_forward']['us'],k['fused_backward']['us']))"
    done
  done
done
