#!/bin/bash
# bash scratch/variant_ab.sh <tag> <variants...>: fused kernel times of library variants on ONE box (boxes differ by ~3 %):
# the metric's batch and the batch tiled 64 times, three alternating rounds
set -u
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
for r in 1 2 3; do
  for sc in 1 64; do
    for v in "$@"; do
      RWK_VARIANT_LIB=scratch/librwk_$v.so timeout -k 10 200 python3 -u scratch/variant_bench.py --kernels-only --config proteins1024 --roofline-scale $sc --only fused_forward,fused_backward > $out/${v}_x${sc}_$r.json 2> $out/${v}_x${sc}_$r.err || { tail -5 $out/${v}_x${sc}_$r.err; exit 1; }
      python3 -c "
import json
k=json.load(open('$out/${v}_x${sc}_$r.json'))['kernels']
print('%-8s x%-3s round $r' % ('$v', '$sc'), 'fwd %.1f bwd %.1f' % (k['fused_forward']['us'],k['fused_backward']['us']))"
    done
  done
done
