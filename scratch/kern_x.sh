#!/bin/bash
# kernels-only timings of the fused kernels at the metric's batch and tiled 64 times
set -u
out=gpurun_out/$1; mkdir -p $out
for cfg in "proteins1024 1" "proteins1024 64" "powerlaw 32"; do set -- $cfg
timeout -k 10 200 python3 -u bench.py --kernels-only --config $1 --roofline-scale $2 > $out/k_$1_$2.json 2> $out/k.err || { tail $out/k.err; exit 1; }
python3 -c "
import json; d=json.load(open('$out/k_$1_$2.json')); print('$1 x$2', d['shape']['stream_len'], {k:v['us'] for k,v in d['kernels'].items()})"
done
