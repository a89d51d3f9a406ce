#!/bin/bash
# usage: bash scratch/pmc.sh <tag> <args to kprof.py>
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $out/p1 -- python3 scratch/kprof.py "$@" > $out/p1.log 2>&1 || exit 1
timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $out/p2 -- python3 scratch/kprof.py "$@" > $out/p2.log 2>&1 || exit 1
python3 - <<PY
import csv,glob,collections
for d in ('p1','p2'):
    f=glob.glob('$out/'+d+'/*/*counter_collection.csv')
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k=r['Kernel_Name'][:60]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); 
    for k,v in agg.items():
        if 'kernel' in k and ('ward' in k or 'spmm' in k):
            print(k, {c: round(x) for c,x in v.items()})
PY
