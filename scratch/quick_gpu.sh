#!/bin/bash
# Usage on the GPU box: bash scratch/quick_gpu.sh <tag> [pytest -k expression]
# Kernel timings of the headline batch, the training step replayed from a graph, and a parity subset.
set -u
tag=$1; kexpr=${2:-"dropout or layer_forward or train_mode"}
out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 120 python3 -u bench.py --kernels-only --config proteins1024 > $out/kernels.json 2> $out/kernels.err || { tail -5 $out/kernels.err; exit 1; }
python3 - <<PY
import json
d=json.load(open('$out/kernels.json'))
print({k:v['us'] for k,v in d['kernels'].items()})
PY
timeout -k 10 200 python3 -u bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-kernel-timing > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
python3 -c "import json; d=json.load(open('$out/bench.json')); print('step ms', d['ms_per_step'], 'graphs/s', d['value'])"
timeout -k 10 500 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -k "$kexpr" > $out/tests.log 2>&1; rc=$?
tail -2 $out/tests.log
exit $rc
