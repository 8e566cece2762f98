"""Per-phase kernel time of the last bench step in a rocprofv3 kernel trace.
usage: phase_prof.py <dir> [n_top]"""
import csv, collections, sys, glob
import os
f = max(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
im = [i for i, n in enumerate(names) if 'cgg_instance_final_kernel' in n]
step = rows[im[-3] + 1: im[-1] + 1]
def short(n):
    return n.replace('void at::native::', '').replace('(anonymous namespace)::', '').replace('at::native::', '')[:150]
phase = 'backbone'
agg = collections.OrderedDict()
for r in step:
    n = r['Kernel_Name']
    if phase == 'backbone' and ('cgg_gn_partial' in n or 'cgg_gn_nhwc_stats' in n): phase = 'pixel_decoder'
    elif phase == 'pixel_decoder' and ('cgg_pack_kernel' in n or 'cgg_pack_nhwc' in n): phase = 'query_decoder'
    elif phase == 'query_decoder' and ('upsample' in n or 'softmax' in n.lower() and 'cgg' not in n and False): phase = 'postproc'
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    a = agg.setdefault(phase, collections.defaultdict(lambda: [0, 0]))
    a[short(n)][0] += d; a[short(n)][1] += 1
top = int(sys.argv[2]) if len(sys.argv) > 2 else 14
for ph, a in agg.items():
    tot = sum(v[0] for v in a.values()); cnt = sum(v[1] for v in a.values())
    print('== %s: %.2f ms, %d kernels' % (ph, tot / 1e6, cnt))
    for k, v in sorted(a.items(), key=lambda kv: -kv[1][0])[:top]:
        print('  %8.1f us x%-4d %s' % (v[0] / 1e3, v[1], k))
