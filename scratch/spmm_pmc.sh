#!/bin/bash
out=gpurun_out/spmm_pmc; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
i=0
for ctrs in "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 90 rocprofv3 --pmc $ctrs --output-format csv -d $out/p$i -- python3 scratch/spmm_only.py $1 > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/p$i.log; exit 1; }
done
python3 - <<PY
import csv,glob,collections
for d in ('p1','p2','p3','p4'):
    f=glob.glob('$out/'+d+'/**/*counter_collection.csv', recursive=True)
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set)
    for r in csv.DictReader(open(f[0])):
        k=r['Kernel_Name'][:50]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k].add(r['Dispatch_Id'])
    for k,v in agg.items():
        if 'spmm' in k:
            print(k, len(cnt[k]), {c: round(x/len(cnt[k])) for c,x in v.items()})
PY
