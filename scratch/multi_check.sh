#!/bin/bash
set -u
out=gpurun_out/$1; mkdir -p $out
timeout -k 10 300 python3 -u -m pytest tests/test_gpu_aux.py -x -q -m gpu --timeout 120 --timeout-method thread > $out/tests_aux.log 2>&1 || { tail -30 $out/tests_aux.log; exit 1; }
tail -2 $out/tests_aux.log
bash scratch/short_region.sh $1
timeout -k 10 200 python3 -u bench.py --no-cpu-baseline --no-kernel-timing --steps-per-graph 1 > $out/long1.json 2> $out/short.err || { tail -20 $out/short.err; exit 1; }
python3 -c "
import json; d=json.load(open('$out/long1.json')); print('200/20 one step per replay: ms_per_step', d['ms_per_step'], 'device', d['device_ms_per_step'])"
timeout -k 10 200 python3 -u bench.py --no-cpu-baseline --no-kernel-timing --steps-per-graph 10 > $out/long10.json 2> $out/short.err || { tail -20 $out/short.err; exit 1; }
python3 -c "
import json; d=json.load(open('$out/long10.json')); print('200/20 ten steps per replay: ms_per_step', d['ms_per_step'], 'device', d['device_ms_per_step'])"
