#!/bin/bash
set -u
out=gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 300 python3 -u -m pytest tests/test_gpu_aux.py tests/test_gpu_sweep.py tests/test_gpu_fullsize.py -x -q -m gpu --timeout 120 --timeout-method thread > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -2 $out/tests.log
for cfg in "powerlaw 32" "proteins1024 64" "proteins1024 1"; do set -- $cfg
timeout -k 10 200 python3 -u bench.py --kernels-only --config $1 --roofline-scale $2 > $out/k_$1_$2.json 2> $out/k.err || { tail $out/k.err; exit 1; }
python3 -c "
import json; d=json.load(open('$out/k_$1_$2.json')); print('$1 x$2', {k:v['us'] for k,v in d['kernels'].items()})"
done
timeout -k 10 200 python3 -u scratch/eager_prof.py > $out/eager_prof.txt 2>&1; head -40 $out/eager_prof.txt
