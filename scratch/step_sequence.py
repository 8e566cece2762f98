"""Ordered kernel sequence of ONE eager inference step (rocprofv3 kernel trace of bench.py --graph 0 --pipeline 0)."""
import csv, glob, os, sys
f = max(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'cgg_stem_conv7x7' in r['Kernel_Name']]
seg = rows[idx[-2]:idx[-1]]
t0 = int(seg[0]['Start_Timestamp'])
cum = 0.0
for r in seg:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    cum += d
    print('%8.1f %7.1f %8.1f  %s  grid=%s' % ((int(r['Start_Timestamp']) - t0) / 1e3, d, cum, r['Kernel_Name'].replace('void ', '')[:90], r['Grid_Size_X']))
