"""Per-kernel duration and the gap before each kernel inside the replayed step, from a rocprofv3 kernel trace.
usage: python3 scratch/step_gaps.py <kernel_trace.csv> [kernels per step]"""
import collections
import csv
import statistics as st
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
idx = [i for i, n in enumerate(names) if 'hidden_backward_reduce_adam' in n]
per = int(sys.argv[2]) if len(sys.argv) > 2 else collections.Counter(b - a for a, b in zip(idx[:-1], idx[1:])).most_common(1)[0][0]
dur = collections.defaultdict(list)
gap = collections.defaultdict(list)
steps = []
for a, b in zip(idx[:-1], idx[1:]):
    if b - a != per:
        continue
    seq = rows[a + 1:b + 1]
    t0 = int(rows[a]['End_Timestamp'])
    steps.append(int(seq[-1]['End_Timestamp']) - t0)
    prev = t0
    for j, r in enumerate(seq):
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        key = (j, r['Kernel_Name'].split('(')[0][-46:], r['Grid_Size_X'], r['Workgroup_Size_X'], r['VGPR_Count'], r['LDS_Block_Size'])
        dur[key].append(e - s)
        gap[key].append(s - prev)
        prev = e
print('kernels per step', per, 'steps', len(steps), 'median step us %.2f' % (st.median(steps) / 1e3))
td = tg = 0
for k in sorted(dur):
    d = st.median(dur[k]); g = st.median(gap[k]); td += d; tg += g
    print('%d %-46s grid %8s wg %4s vgpr %3s lds %6s dur %6.2f us gap %5.2f us' % (k[0], k[1], k[2], k[3], k[4], k[5], d / 1e3, g / 1e3))
print('sum dur %.1f gaps %.1f' % (td / 1e3, tg / 1e3))
