#!/bin/bash
# bash scratch/build_variant_mfma.sh <name> [-D...]: scratch/librwk_<name>.so with rwk_fused_mfma.hip compiled under the given defines
set -e
name=$1; shift
cd "$(dirname "$0")/../gnn_amd/csrc"
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -munsafe-fp-atomics"
/opt/rocm/bin/hipcc $F "$@" -c rwk_fused_mfma.hip -o /tmp/rwk_fused_mfma_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC rwk_structure.o rwk_spmm.o rwk_hidden.o rwk_fused.o rwk_fused_reg.o /tmp/rwk_fused_mfma_$name.o rwk_project.o rwk_merged.o rwk_step.o rwk_head.o rwk_optim.o rwk_khop.o -o ../../scratch/librwk_$name.so
