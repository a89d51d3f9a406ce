#!/bin/bash
# bash scratch/variant_ab_mfma.sh <tag> <variants...>: same-box A/B of library variants on the matrix-core kernels of stream_h64
set -u
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
for r in 1 2 3; do
  for v in "$@"; do
    RWK_VARIANT_LIB=scratch/librwk_$v.so timeout -k 10 200 python3 -u scratch/variant_bench.py --kernels-only --config stream_h64 > $out/${v}_$r.json 2> $out/${v}.err || { tail -5 $out/${v}.err; exit 1; }
    python3 -c "
import json; d=json.load(open('$out/${v}_$r.json')); print('%-8s round $r' % '$v', {k:v['us'] for k,v in d['kernels'].items() if 'fused' in k})"
  done
done
