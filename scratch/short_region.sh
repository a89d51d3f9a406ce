#!/bin/bash
# the bench line at the driver's flags (20 timed steps after 5 warm-up steps), three times, host clock against events
set -u
out=gpurun_out/$1; mkdir -p $out
for i in 1 2 3; do
timeout -k 10 200 python3 -u bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $out/short$i.json 2> $out/short.err || { tail -20 $out/short.err; exit 1; }
python3 -c "
import json; d=json.load(open('$out/short$i.json')); print('20/5: ms_per_step', d['ms_per_step'], 'device', d['device_ms_per_step'])"
done
timeout -k 10 200 python3 -u bench.py --no-cpu-baseline --no-kernel-timing > $out/long.json 2> $out/short.err || { tail -20 $out/short.err; exit 1; }
python3 -c "
import json; d=json.load(open('$out/long.json')); print('200/20: ms_per_step', d['ms_per_step'], 'device', d['device_ms_per_step'])"
