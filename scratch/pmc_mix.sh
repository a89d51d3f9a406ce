#!/bin/bash
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $out/p1 -- python3 bench.py --kernels-only "$@" > $out/p1.log 2>&1 || { tail -5 $out/p1.log; exit 1; }
python3 - <<PY
import csv,glob,collections
f=glob.glob('$out/p1/**/*counter_collection.csv', recursive=True)
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set)
for r in csv.DictReader(open(f[0])):
    k=r['Kernel_Name']
    if 'ward_kernel' not in k: continue
    k=('fwd' if 'forward_kernel' in k else 'bwd')
    agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k].add(r['Dispatch_Id'])
for k,v in agg.items():
    n=len(cnt[k]); d={c: x/n for c,x in v.items()}
    w=d['SQ_WAVES']
    print(k, 'waves', round(w), 'per wave:', {c: round(x/w) for c,x in d.items() if c!='SQ_WAVES'})
PY
