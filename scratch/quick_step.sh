#!/bin/bash
# Usage on the GPU box: bash scratch/quick_step.sh <tag>
# head/aux/parity subset, the default bench line at the driver's flags and at the default ones, kernel trace of the
# replayed step with per-kernel durations and gaps
set -u
tag=$1
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 300 python3 -u -m pytest tests/test_gpu_head.py tests/test_gpu_aux.py tests/test_gpu_parity.py -x -q -m gpu --timeout 120 --timeout-method thread > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -2 $out/tests.log
timeout -k 10 300 python3 -u bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_flags.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
python3 -c "
import json; d=json.load(open('$out/bench_driver_flags.json')); print('driver flags: ms_per_step', d['ms_per_step'], 'value', d['value'])"
timeout -k 10 300 python3 -u bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
python3 -c "
import json; d=json.load(open('$out/bench.json')); print('default: ms_per_step', d['ms_per_step'], 'value', d['value'], d.get('drop_in_eager'))"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu-baseline --steps 100 > $out/trace.log 2>&1 || { tail -20 $out/trace.log; exit 1; }
python3 scratch/step_gaps.py $(ls $out/trace/*/*kernel_trace.csv | head -1)
