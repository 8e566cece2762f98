import csv, collections, sys, glob
import os
f = max(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
mark = sys.argv[2] if len(sys.argv) > 2 else 'cgg_instance_final'
per = int(sys.argv[3]) if len(sys.argv) > 3 else 2
im = [i for i, n in enumerate(names) if mark in n]
s0 = im[-per - 1] + 1
s1 = im[-1] + 1
step = rows[s0:s1]
t0 = int(step[0]['Start_Timestamp']); t1 = int(step[-1]['End_Timestamp'])
agg = collections.defaultdict(lambda: [0, 0])
for r in step:
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    k = r['Kernel_Name'].replace('void at::native::', '').replace('(anonymous namespace)::', '')[:95]
    agg[k][0] += d; agg[k][1] += 1
tot = sum(v[0] for v in agg.values())
print('step wall ms %.2f  kernels %d  sum kernel ms %.2f' % ((t1 - t0) / 1e6, len(step), tot / 1e6))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[4]) if len(sys.argv) > 4 else 28]:
    print('%8.1f us x%-4d %s' % (v[0] / 1e3, v[1], k))
