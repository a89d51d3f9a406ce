#!/bin/bash
# refresh of the round's bench lines and the kernel trace of the default command (after the last kernel changes)
set -u
tag=$1; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $out/gpu_tests.log 2>&1 || { tail -20 $out/gpu_tests.log; exit 1; }
tail -2 $out/gpu_tests.log
timeout -k 10 300 python3 -u bench.py > $out/bench_graph.json 2> $out/bench_graph.err || { tail -20 $out/bench_graph.err; exit 1; }
timeout -k 10 300 python3 -u bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_flags.json 2>> $out/bench_graph.err || exit 1
timeout -k 10 300 python3 -u bench.py --mode eager --steps 100 --warmup 10 --no-cpu-baseline > $out/bench_eager.json 2>> $out/bench_graph.err || exit 1
timeout -k 10 300 python3 -u bench.py --config stream_h64 > $out/stream_h64_bench_graph.json 2>> $out/bench_graph.err || exit 1
timeout -k 10 300 python3 -u bench.py --config powerlaw --no-cpu-baseline > $out/powerlaw_bench_graph.json 2>> $out/bench_graph.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py > $out/trace.log 2>&1 || { tail -20 $out/trace.log; exit 1; }
f=$(ls $out/trace/*/*kernel_stats.csv | head -1); cp $f $out/kernel_stats.csv
python3 scratch/step_gaps.py $(ls $out/trace/*/*kernel_trace.csv | head -1)
for f in bench_graph bench_driver_flags bench_eager stream_h64_bench_graph powerlaw_bench_graph; do python3 -c "
import json; d=json.load(open('$out/$f.json')); print('$f', d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'))"; done
