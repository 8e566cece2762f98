"""Per-step kernel table of the inference step from a rocprofv3 kernel trace of
`bench.py --graph 0 --pipeline 0 --steps 20 --warmup 5 --no-cpu-baseline --no-parity-mode --host-results 0` (30 eager steps):
the LAST 10 steps (delimited by the stem convolution launches) are averaged."""
import csv, glob, os, sys, collections
f = max(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'cgg_stem_conv7x7' in r['Kernel_Name']]
n = 10
seg = rows[idx[-n - 1]:idx[-1]]
agg = collections.defaultdict(lambda: [0.0, 0])
for r in seg:
    k = r['Kernel_Name'].replace('void ', '')[:96]
    agg[k][0] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    agg[k][1] += 1
tot = sum(v[0] for v in agg.values())
print('inference step (eager, serial streams): %.0f launches, %.1f us of kernel time per step (mean of %d steps)' % (len(seg) / n, tot / n, n))
print('%9s %6s %6s  %s' % ('us/step', 'calls', '%', 'kernel'))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
    print('%9.1f %6.1f %6.1f  %s' % (v[0] / n, v[1] / n, 100 * v[0] / tot, k))
