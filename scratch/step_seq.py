"""Ordered kernel list of the last eager step in a trace. usage: step_seq.py <dir> [marker] > file"""
import csv, sys, glob, os
f = max(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
mark = sys.argv[2] if len(sys.argv) > 2 else 'cgg_instance_final'
im = [i for i, r in enumerate(rows) if mark in r['Kernel_Name']]
per = int(sys.argv[3]) if len(sys.argv) > 3 else 2
step = rows[im[-per - 1] + 1: im[-1] + 1]
t0 = int(step[0]['Start_Timestamp'])
prev_end = t0
for i, r in enumerate(step):
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    n = r['Kernel_Name'].replace('void at::native::', '').replace('(anonymous namespace)::', '')[:110]
    print('%4d t=%8.1f dur=%7.1f gap=%6.1f grid=%s wg=%s %s' % (i, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3,
          r.get('Grid_Size_X', '?'), r.get('Workgroup_Size_X', '?'), n))
    prev_end = e
