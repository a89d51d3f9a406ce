#!/bin/bash
# Usage on the GPU box (from the repo root): bash run_gpu_round.sh <tag>
# Runs the GPU parity tests, the benchmark (eager and graph launch) and a rocprofv3 kernel trace.
set -u
tag=${1:-r01}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $out/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 $out/tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -u bench.py --mode eager --steps 100 --warmup 10 --no-cpu-baseline > $out/bench_eager.json 2> $out/bench_eager.err
rc=$?; echo "bench eager rc=$rc"; cat $out/bench_eager.json; [ $rc -ne 0 ] && { tail -20 $out/bench_eager.err; exit $rc; }
timeout -k 10 300 python -u bench.py --mode graph --steps 200 --warmup 20 > $out/bench_graph.json 2> $out/bench_graph.err
rc=$?; echo "bench graph rc=$rc"; cat $out/bench_graph.json; [ $rc -ne 0 ] && { tail -20 $out/bench_graph.err; exit $rc; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --mode eager --steps 30 --warmup 5 --no-cpu-baseline > $out/prof.log 2>&1
rc=$?; echo "rocprof rc=$rc"
find $out/prof -name '*kernel_stats.csv' | head -1 | xargs -r head -25
exit $rc
