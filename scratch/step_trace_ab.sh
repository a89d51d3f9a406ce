#!/bin/bash
# Per-kernel durations inside the replayed step of this build and of another library, same box:
# bash scratch/step_trace_ab.sh <other library>   (prints scratch/step_gaps.py's table for each)
set -u
other=$1
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for which in this other this other; do
  rm -rf gpurun_out/sta
  if [ $which = other ]; then cp gnn_amd/librwk.so /tmp/rwk_this.so && cp $other gnn_amd/librwk.so; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/sta -- python3 bench.py --no-extras --no-cpu-baseline --no-kernel-timing > gpurun_out/sta.log 2>&1 || { tail -5 gpurun_out/sta.log; exit 1; }
  if [ $which = other ]; then cp /tmp/rwk_this.so gnn_amd/librwk.so; fi
  echo "== $which build"
  python3 scratch/step_gaps.py $(ls -t gpurun_out/sta/*/*kernel_trace.csv | head -1) | cut -c1-8,90-130
done
