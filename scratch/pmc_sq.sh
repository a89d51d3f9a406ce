#!/bin/bash
# usage: bash scratch/pmc_sq.sh <tag> <bench args...>
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $out/p1 -- python3 bench.py --kernels-only "$@" > $out/p1.log 2>&1 || { tail -5 $out/p1.log; exit 1; }
python3 - <<PY
import csv,glob,collections
f=glob.glob('$out/p1/**/*counter_collection.csv', recursive=True)
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set)
for r in csv.DictReader(open(f[0])):
    k=r['Kernel_Name']
    if 'ward_kernel' not in k and 'spmm' not in k: continue
    k=('fwd ' if 'forward_kernel' in k else 'bwd ' if 'backward_kernel' in k else '')+k.split('(')[0][-40:]
    agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k].add(r['Dispatch_Id'])
for k,v in agg.items():
    n=len(cnt[k]); d={c: x/n for c,x in v.items()}
    wc=d.get('SQ_WAVE_CYCLES',1)
    print(k, n, {c: round(x) for c,x in d.items()})
    print('   wait_any %.2f wait_inst %.2f active %.2f lds_stall %.2f | mfma_busy/(gui_active/8*256*4... ) : mfma_busy %.0f gui_active/8 %.0f -> per-SIMD util %.3f' % (d.get('SQ_WAIT_ANY',0)/wc, d.get('SQ_WAIT_INST_ANY',0)/wc, d.get('SQ_ACTIVE_INST_ANY',0)/wc, d.get('SQ_WAIT_INST_LDS',0)/wc, d.get('SQ_VALU_MFMA_BUSY_CYCLES',0), d.get('GRBM_GUI_ACTIVE',0)/8, d.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/max(d.get('GRBM_GUI_ACTIVE',1)/8*1024,1)))
PY
