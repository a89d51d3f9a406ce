#!/bin/bash
# Round-3 profile collection on the GPU box: bash scratch/collect_r03.sh <tag>
# default bench line (with its extras), the same command under rocprofv3 --kernel-trace --stats, the driver's flags,
# stream_h64 resident and streamed, roofline + counter passes for three configurations, evaluation forward, the
# two-rank rehearsals over gloo.
set -u
tag=$1
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 400 python3 -u bench.py > $out/bench_default.json 2> $out/bench_default.err || { tail -20 $out/bench_default.err; exit 1; }
cut -c1-700 $out/bench_default.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-extras > $out/trace.log 2>&1 || { tail -20 $out/trace.log; exit 1; }
f=$(ls $out/trace/*/*kernel_stats.csv | head -1); cp $f $out/kernel_stats.csv; head -12 $out/kernel_stats.csv | cut -c1-160
timeout -k 10 300 python3 -u bench.py --steps 20 --warmup 5 > $out/bench_driver_flags.json 2> $out/bench_driver_flags.err || { tail -20 $out/bench_driver_flags.err; exit 1; }
timeout -k 10 300 python3 -u bench.py --config stream_h64 --no-cpu-baseline > $out/stream_h64_bench_graph.json 2> $out/stream_h64.err || { tail -20 $out/stream_h64.err; exit 1; }
timeout -k 10 300 python3 -u bench.py --config stream_h64 --stream --steps 124 --warmup 31 --no-cpu-baseline --no-kernel-timing > $out/stream_h64_streamed.json 2> $out/stream_h64_streamed.err || { tail -20 $out/stream_h64_streamed.err; exit 1; }
timeout -k 10 300 python3 -u bench.py --distinct-batches 16 --steps 160 --warmup 16 --no-cpu-baseline --no-kernel-timing > $out/distinct16.json 2> $out/distinct16.err || { tail -20 $out/distinct16.err; exit 1; }
timeout -k 10 300 python3 -u bench.py --config powerlaw --no-cpu-baseline > $out/powerlaw_bench_graph.json 2> $out/powerlaw.err || { tail -20 $out/powerlaw.err; exit 1; }
for f in stream_h64_bench_graph stream_h64_streamed distinct16 powerlaw_bench_graph bench_driver_flags; do python3 -c "
import json; d=json.load(open('$out/$f.json')); print('$f', d['ms_per_step'], d['value'])"; done
timeout -k 10 100 python3 -u scratch/eval_time.py > $out/eval_forward.txt 2>&1; grep -v amdgpu $out/eval_forward.txt
timeout -k 10 100 python3 -u scratch/xchain_time.py > $out/x_chain.txt 2>&1; grep -v amdgpu $out/x_chain.txt
bash scratch/prof_list.sh > $out/step_kernels_eager.txt 2>&1; head -14 $out/step_kernels_eager.txt
bash scratch/prof_cfg.sh stream_h64 > $out/stream_h64_kernels.txt 2>&1; head -14 $out/stream_h64_kernels.txt
bash scratch/prof_stream.sh > $out/stream_h64_streamed_kernels.txt 2>&1; tail -16 $out/stream_h64_streamed_kernels.txt
bash tools/roofline_passes.sh $tag proteins1024 64 > $out/rp_p64.log 2>&1 || { tail -20 $out/rp_p64.log; exit 1; }
bash tools/roofline_passes.sh $tag proteins1024 1 > $out/rp_p1.log 2>&1 || { tail -20 $out/rp_p1.log; exit 1; }
bash tools/roofline_passes.sh $tag powerlaw 32 > $out/rp_pl32.log 2>&1 || { tail -20 $out/rp_pl32.log; exit 1; }
bash tools/roofline_passes.sh $tag stream_h64 1 > $out/rp_s1.log 2>&1 || { tail -20 $out/rp_s1.log; exit 1; }
for c in proteins1024_x64 proteins1024_x1 powerlaw_x32 stream_h64_x1; do echo $c; python3 -c "
import json; d=json.load(open('$out/$c/traffic.json'))
for k,v in d.items():
    for kk,vv in v.items():
        if not kk.startswith('_'): print(' ', kk, vv.get('event_us'), vv.get('hbm_bytes'), vv.get('algorithmic_bytes'))"; done
for sc in weak strong; do RWK_DIST_BACKEND=gloo timeout -k 10 200 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --scaling $sc --steps 30 --warmup 5 --no-kernel-timing > $out/gloo2_$sc.json 2> $out/gloo2_$sc.err || { tail -5 $out/gloo2_$sc.err; exit 1; }; python3 -c "
import json; d=json.loads(open('$out/gloo2_$sc.json').read().strip().splitlines()[-1]); print('gloo2 $sc', d['ms_per_step'], d['value'], d['scaling'])"; done
echo done
# what the later part of the round added: the per-graph chain against the SpMM launches, the one-rank RCCL rehearsal
PYTHONPATH=. timeout -k 10 200 python3 scratch/chain_time.py > $out/chain_time.txt 2>&1; grep -v amdgpu $out/chain_time.txt
PYTHONPATH=. timeout -k 10 200 python3 scratch/chain_ab.py powerlaw256 > $out/chain_ab_powerlaw256.txt 2>&1; grep "ms per" $out/chain_ab_powerlaw256.txt
bash scratch/prof_cfg.sh powerlaw > $out/powerlaw_kernels.txt 2>&1; head -12 $out/powerlaw_kernels.txt
bash scratch/rccl_rehearsal.sh $tag > $out/rccl_rehearsal.txt 2>&1 || { tail -5 $out/rccl_rehearsal.txt; exit 1; }
for f in rccl1_two_graphs rccl1_captured rccl1_stream; do python3 -c "
import json; d=json.loads(open('$out/$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['value'])"; done
echo done2
