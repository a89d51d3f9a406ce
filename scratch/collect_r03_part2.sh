#!/bin/bash
# Second part of the round-3 collection: bash scratch/collect_r03_part2.sh <tag>
# the two-rank strong-scaling rehearsal over gloo (once; a run that stalls dumps its stacks into gpurun_out/trace_rank<r>.txt after 100 s
# and the script ends there), the per-graph chain against the SpMM launches, power-law kernels, the one-rank RCCL rehearsal.
set -u
tag=$1
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
RWK_BENCH_TRACE=100 RWK_DIST_BACKEND=gloo timeout -k 10 160 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --scaling strong --steps 30 --warmup 5 --no-kernel-timing > $out/gloo2_strong.json 2> $out/gloo2_strong.err || { tail -80 $out/gloo2_strong.err; exit 1; }
python3 -c "
import json; d=json.loads(open('$out/gloo2_strong.json').read().strip().splitlines()[-1]); print('gloo2 strong', d['ms_per_step'], d['value'], d['scaling'])"
PYTHONPATH=. timeout -k 10 200 python3 scratch/chain_time.py > $out/chain_time.txt 2>&1 || { tail -5 $out/chain_time.txt; exit 1; }; grep -v amdgpu $out/chain_time.txt
PYTHONPATH=. timeout -k 10 200 python3 scratch/chain_ab.py powerlaw256 > $out/chain_ab_powerlaw256.txt 2>&1 || { tail -5 $out/chain_ab_powerlaw256.txt; exit 1; }; grep "ms per" $out/chain_ab_powerlaw256.txt
bash scratch/prof_cfg.sh powerlaw > $out/powerlaw_kernels.txt 2>&1 || { tail -5 $out/powerlaw_kernels.txt; exit 1; }; head -12 $out/powerlaw_kernels.txt
bash scratch/rccl_rehearsal.sh $tag > $out/rccl_rehearsal.txt 2>&1 || { tail -5 $out/rccl_rehearsal.txt; exit 1; }
for f in rccl1_two_graphs rccl1_captured rccl1_stream; do python3 -c "
import json; d=json.loads(open('$out/$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['value'])"; done
echo done2
