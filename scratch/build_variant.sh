#!/bin/bash
# bash scratch/build_variant.sh <name> <file.hip> <-Dflags...>: scratch/librwk_<name>.so with one source rebuilt under extra flags
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../gnn_amd/csrc"
make -s >/dev/null
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -munsafe-fp-atomics $*"
/opt/rocm/bin/hipcc $F -c $src.hip -o /tmp/${src}_$name.o
objs=""
for o in rwk_structure rwk_spmm rwk_hidden rwk_fused rwk_fused_reg rwk_fused_mfma rwk_project rwk_merged rwk_step rwk_head rwk_optim rwk_khop; do
  if [ $o = ${SRC_REPLACES:-$src} ]; then objs="$objs /tmp/${src}_$name.o"; else objs="$objs $o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o ../../scratch/librwk_$name.so
echo built scratch/librwk_$name.so
