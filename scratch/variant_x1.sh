#!/bin/bash
# bash scratch/variant_x1.sh <tag> <variants...>: headline-size kernel times, three alternating rounds
set -u
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
for r in 1 2 3; do
  for v in "$@"; do
    RWK_VARIANT_LIB=scratch/librwk_$v.so timeout -k 10 200 python3 -u scratch/variant_bench.py --kernels-only --config proteins1024 --roofline-scale 1 > $out/${v}_x1_$r.json 2> $out/${v}_x1_$r.err || { tail -5 $out/${v}_x1_$r.err; exit 1; }
    python3 -c "
import json
k=json.load(open('$out/${v}_x1_$r.json'))['kernels']
print('$v round $r', 'spmm %.1f fwd %.1f bwd %.1f' % (k['spmm_csr']['us'],k['fused_forward']['us'],k['fused_backward']['us']))"
  done
done
