"""Overlap analysis of the pipelined steady state. usage: pipe_overlap.py <trace dir> [window_ms=30]"""
import csv, sys, glob, os, collections
f = max(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
win = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
# steady state: find the last instance-final kernel, go back `win` ms ... but skip the eager event-timed tail:
# use the densest region: the pipelined timed loop is where two queues are active.
t_end_all = int(rows[-1]['End_Timestamp'])
qs = collections.Counter(r['Queue_Id'] for r in rows)
print('queues', qs.most_common(6))
# choose window = [T - win, T] where T = end of last kernel on the second most active *pipeline* queue
def analyse(t0, t1, label):
    sel = [r for r in rows if int(r['Start_Timestamp']) >= t0 and int(r['End_Timestamp']) <= t1]
    if not sel:
        return
    per_q = collections.defaultdict(list)
    for r in sel:
        per_q[r['Queue_Id']].append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
    ev = sorted((s, e) for v in per_q.values() for (s, e) in v)
    busy, cur_s, cur_e = 0, ev[0][0], ev[0][1]
    for s, e in ev[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    span = t1 - t0
    print('%s: window %.2f ms, union busy %.2f ms (%.1f%%), kernels %d' % (label, span / 1e6, busy / 1e6, 100.0 * busy / span, len(sel)))
    for q, v in per_q.items():
        tot = sum(e - s for s, e in v)
        print('   queue %s: %d kernels, sum %.2f ms (%.1f%% of window)' % (q, len(v), tot / 1e6, 100.0 * tot / span))
marks = [i for i, r in enumerate(rows) if 'cgg_instance_final' in r['Kernel_Name']]
# pipelined region: marks whose queue differs from the eager tail's queue
tailq = rows[marks[-1]]['Queue_Id']
pm = [i for i in marks if rows[i]['Queue_Id'] != tailq]
if pm:
    t1 = int(rows[pm[-1]]['End_Timestamp'])
    analyse(t1 - int(win * 1e6), t1, 'pipelined')
    # per-kernel slowdown inside the window vs the eager tail, by name
    t0 = t1 - int(win * 1e6)
    ins = collections.defaultdict(list); out = collections.defaultdict(list)
    for r in rows:
        d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        if t0 <= int(r['Start_Timestamp']) <= t1:
            ins[r['Kernel_Name'][:80]].append(d)
        elif int(r['Start_Timestamp']) > t1 and r['Queue_Id'] == tailq:
            out[r['Kernel_Name'][:80]].append(d)
    tot_in = tot_out = 0
    lines = []
    for k, v in ins.items():
        if k in out:
            a, b = sum(v) / len(v), sum(out[k]) / len(out[k])
            lines.append((sum(v) / 1e3, a / 1e3, b / 1e3, len(v), k))
    lines.sort(reverse=True)
    print('kernel: total_us_in_window  avg_pipelined_us  avg_eager_us  n')
    for l in lines[:25]:
        print('%10.1f %8.1f %8.1f %5d  %s' % l)
t1 = int(rows[marks[-1]]['End_Timestamp'])
analyse(t1 - int(win * 1e6), t1, 'eager tail')
