#!/bin/bash
# One-GPU rehearsal of the data-parallel benchmark path (two ranks share the device, all-reduce over gloo) and the
# cost of the data-parallel step shape itself on one device.  Usage: bash scratch/dp_rehearsal.sh <tag>
tag=$1
out=gpurun_out/$tag; mkdir -p $out
for cfg in proteins1024 powerlaw stream_h64; do
  RWK_DIST_BACKEND=gloo timeout -k 10 200 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 30 --warmup 5 --config $cfg --no-kernel-timing > $out/gloo2_$cfg.json 2> $out/gloo2_$cfg.err || { tail -5 $out/gloo2_$cfg.err; exit 1; }
  python3 -c "import json,sys; d=json.loads(open('$out/gloo2_$cfg.json').read().strip().splitlines()[-1]); print('$cfg gloo2', d['ms_per_step'], d.get('phases'))"
done
timeout -k 10 200 python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-kernel-timing > $out/shape1.json 2> $out/shape1.err || exit 1
timeout -k 10 200 python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-kernel-timing --dp-shape 2 > $out/shape2.json 2> $out/shape2.err || { tail -5 $out/shape2.err; exit 1; }
python3 -c "
import json
a=json.load(open('$out/shape1.json')); b=json.load(open('$out/shape2.json'))
print('single graph ms', a['ms_per_step'], '| two graphs + collective slot ms', b['ms_per_step'], b.get('phases'))"
