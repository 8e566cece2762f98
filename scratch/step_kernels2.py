"""Per-step kernel table from a rocprofv3 kernel trace of an EAGER bench run (any precision mode):
`bench.py --graph 0 --pipeline 0 --steps K --warmup W ... ` -- the trace's launches are cut into steps at every launch of a
delimiter kernel (argv[2]; default: the first kernel, in time order, that is launched exactly once per step in the tail of
the trace) and the LAST n steps are averaged.  usage: step_kernels2.py <trace dir> [delimiter substring] [n] [delimiter launches per step | auto:<steps run>]"""
import collections
import csv
import glob
import os
import sys

f = max(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
delim = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] else None
n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
if delim is None:
    # candidates: names seen in the last quarter of the trace; pick the rarest one with >= n + 1 launches there
    tail = rows[len(rows) * 3 // 4:]
    cnt = collections.Counter(r['Kernel_Name'] for r in tail)
    ok = [k for k, c in cnt.items() if c >= n + 1]
    m = min(cnt[k] for k in ok)
    delim = next(r['Kernel_Name'] for r in tail if r['Kernel_Name'] in ok and cnt[r['Kernel_Name']] == m)
idx = [i for i, r in enumerate(rows) if delim in r['Kernel_Name']]
per = 1
if len(sys.argv) > 4:
    if sys.argv[4].startswith('auto:'):       # auto:<steps the traced process ran>: delimiter launches per step = their count / steps
        per = max(1, round(len(idx) / int(sys.argv[4][5:])))
    else:
        per = int(sys.argv[4])
seg = rows[idx[-n * per - 1]:idx[-1]]
agg = collections.defaultdict(lambda: [0.0, 0])
for r in seg:
    k = r['Kernel_Name'].replace('void ', '')[:110]
    agg[k][0] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    agg[k][1] += 1
tot = sum(v[0] for v in agg.values())
span = (int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e3
print('delimiter: %s' % delim[:100])
print('step (eager, serial): %.0f launches, %.1f us of kernel time, %.1f us wall per step (mean of %d steps)'
      % (len(seg) / n, tot / n, span / n, n))
own = sum(v[0] for k, v in agg.items() if 'cgg_' in k.split('(')[0])
print('hand-written (cgg_*) kernels: %.1f us per step = %.1f %% of the kernel time' % (own / n, 100 * own / tot))
print('%9s %6s %6s  %s' % ('us/step', 'calls', '%', 'kernel'))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:60]:
    print('%9.1f %6.1f %6.1f  %s' % (v[0] / n, v[1] / n, 100 * v[0] / tot, k))
