#!/bin/bash
# The judged bench files of a tag from the final tree: bash scratch/final_bench.sh <tag>
# smoke(), the default bench line, the same command under rocprofv3 --kernel-trace --stats, the driver's flags.
set -u
tag=$1
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.txt 2>&1 || { tail -20 $out/smoke.txt; exit 1; }
tail -2 $out/smoke.txt
timeout -k 10 400 python3 -u bench.py > $out/bench_default.json 2> $out/bench_default.err || { tail -20 $out/bench_default.err; exit 1; }
cut -c1-400 $out/bench_default.json
rm -rf $out/trace
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-extras > $out/trace.log 2>&1 || { tail -20 $out/trace.log; exit 1; }
f=$(ls $out/trace/*/*kernel_stats.csv | head -1); cp $f $out/kernel_stats.csv; head -5 $out/kernel_stats.csv | cut -c1-160
timeout -k 10 300 python3 -u bench.py --steps 20 --warmup 5 > $out/bench_driver_flags.json 2> $out/bench_driver_flags.err || { tail -20 $out/bench_driver_flags.err; exit 1; }
python3 -c "
import json
for f in ('bench_default', 'bench_driver_flags'):
    d = json.load(open('$out/%s.json' % f)); print(f, d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['us'])"
