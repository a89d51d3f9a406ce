#!/bin/bash
# bash scratch/variant_big.sh <tag> <variants...>: kernel times at the two tiled sizes, two alternating rounds
set -u
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
for r in 1 2; do
  for cs in "proteins1024 64" "powerlaw 32"; do
    set -- $cs "$@"; cfg=$1; sc=$2; shift 2
    for v in "$@"; do
      RWK_VARIANT_LIB=scratch/librwk_$v.so timeout -k 10 200 python3 -u scratch/variant_bench.py --kernels-only --config $cfg --roofline-scale $sc > $out/${v}_${cfg}_x${sc}_$r.json 2> $out/${v}_${cfg}_x${sc}_$r.err || { tail -5 $out/${v}_${cfg}_x${sc}_$r.err; exit 1; }
      python3 -c "
import json
k=json.load(open('$out/${v}_${cfg}_x${sc}_$r.json'))['kernels']
print('$v $cfg x$sc round $r', 'spmm %.1f fwd %.1f bwd %.1f' % (k['spmm_csr']['us'],k['fused_forward']['us'],k['fused_backward']['us']))"
    done
  done
done
