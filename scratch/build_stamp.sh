#!/bin/bash
# Builds scratch/librwk_stamp.so: the library with the cycle-stamp diagnostics of rwk_fused_reg.hip, rwk_fused_mfma.hip and rwk_head.hip
# compiled in (-DRWK_STAMP).
set -e
cd "$(dirname "$0")/../gnn_amd/csrc"
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -munsafe-fp-atomics -DRWK_STAMP"
/opt/rocm/bin/hipcc $F -c rwk_fused_reg.hip -o /tmp/rwk_fused_reg_stamp.o &
/opt/rocm/bin/hipcc $F -c rwk_head.hip -o /tmp/rwk_head_stamp.o &
/opt/rocm/bin/hipcc $F -c rwk_fused_mfma.hip -o /tmp/rwk_fused_mfma_stamp.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC rwk_structure.o rwk_spmm.o rwk_hidden.o rwk_fused.o /tmp/rwk_fused_reg_stamp.o /tmp/rwk_fused_mfma_stamp.o rwk_project.o rwk_merged.o rwk_step.o /tmp/rwk_head_stamp.o rwk_optim.o rwk_khop.o -o ../../scratch/librwk_stamp.so
