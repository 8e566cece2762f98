"""Kernel totals of one training step in a trace (between the last MSDeformAttn-backward launches of two steps)."""
import csv, sys, glob, os, collections
f = max(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'cgg_msda_bwd_tiled' in r['Kernel_Name']]
# group consecutive adam launches
groups = []
for i in idx:
    if groups and i - groups[-1][-1] < 3000: groups[-1].append(i)
    else: groups.append([i])
s0, s1 = groups[-2][-1] + 1, groups[-1][-1] + 1
step = rows[s0:s1]
agg = collections.defaultdict(lambda: [0, 0])
for r in step:
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    k = r['Kernel_Name'].replace('void ', '')[:110]
    agg[k][0] += d; agg[k][1] += 1
tot = sum(v[0] for v in agg.values())
wall = int(step[-1]['End_Timestamp']) - int(step[0]['Start_Timestamp'])
print('step: %d kernels, sum %.1f ms, wall %.1f ms' % (len(step), tot / 1e6, wall / 1e6))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
    print('%8.2f ms %5.1f%% x%-5d %s' % (v[0] / 1e6, 100.0 * v[0] / tot, v[1], k))
if len(sys.argv) > 2 and sys.argv[2] == 'big':
    # individual launches >= 0.3 ms, grouped by (kernel, rounded duration)
    big = collections.defaultdict(lambda: [0, 0])
    for r in step:
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
        if d >= 0.1:
            big[(r['Kernel_Name'].replace('void ', '')[:90], round(d, 1))][0] += d
            big[(r['Kernel_Name'].replace('void ', '')[:90], round(d, 1))][1] += 1
    print('\nlaunches >= 0.1 ms grouped by (kernel, duration rounded to 0.1 ms):')
    for (k, d), v in sorted(big.items(), key=lambda kv: -kv[1][0])[:60]:
        print('%8.2f ms  x%-4d ~%.1f ms each  %s' % (v[0], v[1], d, k))
