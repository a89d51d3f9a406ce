#!/bin/bash
# head kernels: parity tests, cycle stamps of the training-step kernel, the default bench line
set -u
out=gpurun_out/quick_head; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 300 python3 -u -m pytest tests/test_gpu_head.py tests/test_gpu_aux.py -x -q -m gpu --timeout 120 --timeout-method thread > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -2 $out/tests.log
timeout -k 10 200 python3 -u scratch/stamp_head.py > $out/stamp.log 2>&1 || { tail -20 $out/stamp.log; exit 1; }; tail -1 $out/stamp.log
timeout -k 10 200 python3 -u scratch/stamp_head.py > $out/stamp.log 2>&1 || { tail -20 $out/stamp.log; exit 1; }; tail -1 $out/stamp.log
timeout -k 10 300 python3 -u bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
python3 -c "
import json; d=json.load(open('$out/bench.json')); print('ms_per_step', d['ms_per_step'], 'value', d['value'])"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu-baseline --steps 100 > $out/trace.log 2>&1 || { tail -20 $out/trace.log; exit 1; }
f=$(ls $out/trace/*/*kernel_stats.csv | head -1); cp $f $out/kernel_stats.csv
python3 - <<PY
import csv
for r in list(csv.DictReader(open('$out/kernel_stats.csv')))[:26]:
    n=r['Name']
    if 'at::native' in n or 'softmax_warp' in n: continue
    print(f"{float(r['AverageNs'])/1e3:8.2f} us  calls {r['Calls']:>5}  {n[:80]}")
PY
