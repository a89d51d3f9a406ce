#!/bin/bash
# Usage on the GPU box (from the repo root): bash tools/pmc_passes.sh <tag> <name> "<counters of one pass>" <filter> [bench args]
# One rocprofv3 --pmc pass (a run of its own, no tracing beside it) of `bench.py --kernels-only`, summarised per
# kernel into gpurun_out/<tag>/<name>.json by tools/pmc_summary.py.
set -u
tag=$1; name=$2; counters=$3; filter=$4; shift 4
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf $out/${name}_raw
timeout -s KILL 180 rocprofv3 --pmc $counters --output-format csv -d $out/${name}_raw -- python3 bench.py --kernels-only "$@" > $out/${name}.log 2>&1 || { tail -5 $out/${name}.log; exit 1; }
python3 tools/pmc_summary.py $out/${name}_raw $filter > $out/${name}.json || exit 1
python3 -c "
import json
d=json.load(open('$out/${name}.json'))
for k,v in d.items(): print(k[:70], v['dispatches'], v['counters'])"
