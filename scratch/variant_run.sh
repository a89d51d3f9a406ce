#!/bin/bash
# bash scratch/variant_run.sh <tag> <variant names...>: fused kernel times of each variant library at three sizes
set -u
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
for v in "$@"; do
  for cs in "proteins1024 1" "proteins1024 64" "powerlaw 32"; do
    set -- $cs
    RWK_VARIANT_LIB=scratch/librwk_$v.so timeout -k 10 200 python3 -u scratch/variant_bench.py --kernels-only --config $1 --roofline-scale $2 > $out/${v}_$1_x$2.json 2> $out/${v}_$1_x$2.err || { tail -5 $out/${v}_$1_x$2.err; exit 1; }
    python3 -c "
import json,sys
d=json.load(open('$out/${v}_$1_x$2.json'))
k=d['kernels']
print('$v $1 x$2', 'spmm %.1f fwd %.1f bwd %.1f' % (k['spmm_csr']['us'],k['fused_forward']['us'],k['fused_backward']['us']))"
  done
done
