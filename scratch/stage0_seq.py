"""Per-launch durations of the backbone graph in replay order, averaged over the last 20 backbone-only replays.
usage: stage0_seq.py <trace dir A> <trace dir B>"""
import csv, glob, os, sys
def load(d):
    f = max(glob.glob(d + '/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    stem = [i for i, r in enumerate(rows) if 'stem_conv7x7' in r['Kernel_Name']]
    blocks = []
    for a, b in zip(stem, stem[1:]):
        seg = rows[a:b]
        if not any('class_topk' in r['Kernel_Name'] or 'msda' in r['Kernel_Name'] for r in seg):
            blocks.append(seg)
    segs = blocks[-20:]
    n = len(segs[0])
    out = []
    for k in range(n):
        d = sum((int(s[k]['End_Timestamp']) - int(s[k]['Start_Timestamp'])) / 1e3 for s in segs) / len(segs)
        nm = segs[0][k]['Kernel_Name']
        nm = nm[nm.find('<'):nm.find('>') + 1] if '<' in nm else nm[:30]
        out.append((d, nm, segs[0][k].get('Grid_Size', '?'), segs[0][k].get('LDS_Block_Size', '?')))
    return out
A, B = load(sys.argv[1]), load(sys.argv[2])
ta = tb = 0
for i, (a, b) in enumerate(zip(A, B)):
    ta += a[0]; tb += b[0]
    print('%2d  %7.1f %-28s g%-8s lds %-7s | %7.1f %-14s g%-8s  %+6.1f' % (i, a[0], a[1], a[2], a[3], b[0], b[1], b[2], a[0] - b[0]))
print('total', ta, tb)
