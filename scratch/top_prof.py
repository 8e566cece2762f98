"""Top kernels by total time in the last fraction of a trace. usage: top_prof.py <dir> [frac=0.3] [n=40]"""
import csv, collections, sys, glob
import os
f = max(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
rows = rows[int(len(rows) * (1 - frac)):]
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    k = r['Kernel_Name'].replace('void at::native::', '').replace('(anonymous namespace)::', '').replace('at::native::', '')[:170]
    agg[k][0] += d; agg[k][1] += 1
tot = sum(v[0] for v in agg.values())
wall = int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])
print('kernels %d  sum kernel ms %.1f  wall ms %.1f' % (len(rows), tot / 1e6, wall / 1e6))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print('%9.2f ms %5.1f%% x%-5d %s' % (v[0] / 1e6, 100.0 * v[0] / tot, v[1], k))
