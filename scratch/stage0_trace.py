"""Analyse a rocprofv3 kernel trace of scratch/stage_times.py: the LAST back-to-back replay block of stage 0 (the backbone graph):
sum of kernel durations, span, per-kernel gaps. usage: stage0_trace.py <trace dir>"""
import csv, glob, os, sys, collections
f = max(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# stage 0 replays = the first timed block: find runs of the stem kernel
stem = [i for i, r in enumerate(rows) if 'stem_conv7x7' in r['Kernel_Name']]
# consecutive stem launches with no class_topk in between = stage-0-alone block
blocks = []
for a, b in zip(stem, stem[1:]):
    seg = rows[a:b]
    if not any('class_topk' in r['Kernel_Name'] or 'msda' in r['Kernel_Name'] for r in seg):
        blocks.append(seg)
print(len(blocks), 'backbone-only replays found')
segs = blocks[-20:]
tot = collections.defaultdict(lambda: [0.0, 0.0, 0])
span = dur = 0.0
for seg in segs:
    span += (int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e3
    for k, r in enumerate(seg):
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        gap = (int(seg[k + 1]['Start_Timestamp']) - int(r['End_Timestamp'])) / 1e3 if k + 1 < len(seg) else 0.0
        dur += d
        key = r['Kernel_Name'][:60] + ' grid ' + r.get('Grid_Size', r.get('Grid_Size_X', '?'))
        tot[key][0] += d; tot[key][1] += gap; tot[key][2] += 1
n = len(segs)
print('per replay: %d kernels, sum of durations %.1f us, span %.1f us' % (len(segs[0]), dur / n, span / n))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][0])[:40]:
    print('%8.1f us  gap after %6.1f  x%.1f  %s' % (v[0] / n, v[1] / n, v[2] / n, k))
