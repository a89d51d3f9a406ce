#!/bin/bash
set -u
out=gpurun_out/$1; mkdir -p $out
timeout -k 10 300 python3 -u -m pytest tests/test_gpu_aux.py -x -q -m gpu --timeout 120 --timeout-method thread > $out/tests_aux.log 2>&1 || { tail -40 $out/tests_aux.log; exit 1; }
tail -2 $out/tests_aux.log
for sh in 1 2; do
timeout -k 10 200 python3 -u bench.py --no-cpu-baseline --no-kernel-timing --dp-shape $sh > $out/shape$sh.json 2> $out/shape.err || { tail -20 $out/shape.err; exit 1; }
python3 -c "
import json; d=json.load(open('$out/shape$sh.json')); print('dp-shape $sh: ms_per_step', d['ms_per_step'], d.get('phases'))"
done
