#!/bin/bash
# Usage on the GPU box (from the repo root): bash tools/roofline_passes.sh <tag> <config> <scale> [bench args]
# For one configuration and tiling factor: event-timed kernel launches, a rocprofv3 kernel trace of the same
# command, and two counter passes (FETCH_SIZE, WRITE_SIZE; each in a run of its own) summarised into
# gpurun_out/<tag>/traffic_<config>@<scale>.json.
set -u
tag=$1; cfg=$2; scale=$3; shift 3
out=gpurun_out/$tag/${cfg}_x${scale}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
cmd="bench.py --kernels-only --config $cfg --roofline-scale $scale $*"
timeout -k 10 240 python3 -u $cmd > $out/kernels.json 2> $out/kernels.err
rc=$?; echo "kernels rc=$rc"; cat $out/kernels.json; [ $rc -ne 0 ] && { tail -20 $out/kernels.err; exit $rc; }
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $cmd > $out/trace.log 2>&1
rc=$?; echo "trace rc=$rc"; [ $rc -ne 0 ] && { tail -20 $out/trace.log; exit $rc; }
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $cmd --calibrate > $out/pmc_fetch.log 2>&1
rc=$?; echo "pmc fetch rc=$rc"; [ $rc -ne 0 ] && { tail -20 $out/pmc_fetch.log; exit $rc; }
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $cmd --calibrate > $out/pmc_write.log 2>&1
rc=$?; echo "pmc write rc=$rc"; [ $rc -ne 0 ] && { tail -20 $out/pmc_write.log; exit $rc; }
python3 tools/pmc_traffic.py $out "$cfg@$scale" > $out/traffic.json
rc=$?; cat $out/traffic.json
exit $rc
