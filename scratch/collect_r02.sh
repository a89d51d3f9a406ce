#!/bin/bash
# Round profile collection on the GPU box: default bench line, kernel trace of the same command, roofline passes for
# three configurations, counter passes for the fused kernels and the SpMM.  bash scratch/collect_r02.sh <tag>
set -u
tag=$1
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 300 python3 -u bench.py > $out/bench_graph.json 2> $out/bench_graph.err || { tail -20 $out/bench_graph.err; exit 1; }
cat $out/bench_graph.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py > $out/trace.log 2>&1 || { tail -20 $out/trace.log; exit 1; }
f=$(ls $out/trace/*/*kernel_stats.csv | head -1); cp $f $out/kernel_stats.csv; head -25 $out/kernel_stats.csv
timeout -k 10 300 python3 -u bench.py --config stream_h64 > $out/stream_h64_bench_graph.json 2> $out/stream_h64.err || { tail -20 $out/stream_h64.err; exit 1; }
cat $out/stream_h64_bench_graph.json
bash tools/roofline_passes.sh $tag proteins1024 64 || exit 1
bash tools/roofline_passes.sh $tag proteins1024 1 || exit 1
bash tools/roofline_passes.sh $tag powerlaw 32 || exit 1
bash tools/roofline_passes.sh $tag stream_h64 1 || exit 1
echo "== SQ passes"
bash tools/pmc_passes.sh $tag sq_mix_after "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" rwk_reg --config proteins1024 --roofline-scale 1 || exit 1
bash tools/pmc_passes.sh $tag sq_wait_after "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES SQ_WAVE_CYCLES" rwk_reg --config proteins1024 --roofline-scale 1 || exit 1
echo "== SpMM stall passes"
bash tools/pmc_passes.sh $tag spmm_tcp_x64 "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" spmm --config proteins1024 --roofline-scale 64 || exit 1
bash tools/pmc_passes.sh $tag spmm_tcc_x64 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY" spmm --config proteins1024 --roofline-scale 64 || exit 1
bash tools/pmc_passes.sh $tag spmm_tcp_pl32 "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" spmm --config powerlaw --roofline-scale 32 || exit 1
echo done
