#!/bin/bash
# Builds scratch/librwk_stamp.so: the library with the cycle-stamp diagnostics of rwk_fused_reg.hip compiled in.
set -e
cd "$(dirname "$0")/../gnn_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -munsafe-fp-atomics -DRWK_STAMP -c rwk_fused_reg.hip -o /tmp/rwk_fused_reg_stamp.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC rwk_structure.o rwk_spmm.o rwk_hidden.o rwk_fused.o /tmp/rwk_fused_reg_stamp.o rwk_fused_mfma.o rwk_project.o rwk_merged.o rwk_head.o rwk_optim.o rwk_khop.o -o ../../scratch/librwk_stamp.so
