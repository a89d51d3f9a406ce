#!/bin/bash
# One-rank rehearsal of the multi-rank benchmark path on the real collective library (RCCL): process-group
# initialisation with a device id, the averaging all-reduce of the flat bucket between the step's graph and the
# Adam launch, the same with the collective captured inside the graph, barriers, the final reductions, shutdown.
# Usage: bash scratch/rccl_rehearsal.sh <tag>
tag=$1
out=gpurun_out/$tag; mkdir -p $out
export RWK_REHEARSE_RCCL=1
timeout -k 10 200 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-kernel-timing --dp-shape 2 --no-capture-allreduce > $out/rccl1_two_graphs.json 2> $out/rccl1_two_graphs.err || { tail -20 $out/rccl1_two_graphs.err; exit 1; }
tail -c 600 $out/rccl1_two_graphs.json; echo
timeout -k 10 200 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-kernel-timing --dp-shape 2 > $out/rccl1_captured.json 2> $out/rccl1_captured.err || { tail -20 $out/rccl1_captured.err; exit 1; }
tail -c 600 $out/rccl1_captured.json; echo
timeout -k 10 200 python3 bench.py --steps 62 --warmup 8 --no-cpu-baseline --no-kernel-timing --dp-shape 2 --config stream_h64 --stream > $out/rccl1_stream.json 2> $out/rccl1_stream.err || { tail -20 $out/rccl1_stream.err; exit 1; }
tail -c 600 $out/rccl1_stream.json; echo
# the fall-back after a capture that fails once the collective has been recorded: the communicator must stay usable
RWK_ABORT_CAPTURE_FOR_TEST=1 timeout -k 10 200 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-kernel-timing --dp-shape 2 > $out/rccl1_aborted_capture.json 2> $out/rccl1_aborted_capture.err || { tail -20 $out/rccl1_aborted_capture.err; exit 1; }
tail -c 700 $out/rccl1_aborted_capture.json; echo
