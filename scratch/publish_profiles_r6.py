"""gpurun_out/r6/prof/* (scratch/collect_profiles_r6.sh on the GPU box) -> profiles/r6_* : bench line, kernel stats (top 50), per-step kernel
tables (inference configs[1] / [4], training configs[2] / [3] both precisions), per-kernel PMC aggregates and the summaries bench.py reads
for its `traffic` / `rocprof` fields (`r6_fp32_kernels.json`, `r6_cfg4_kernels.json`, labelled "committed profile"), the MSDeformAttn
backward counters, the agreement record of the full-size tests."""
import collections
import csv
import json
import os
import shutil

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
src, dst = os.path.join(R, 'gpurun_out', 'r6', 'prof'), os.path.join(R, 'profiles')
FAM = ('cgg_gemm_x3s_kernel', 'cgg_encoder_tail_x3_kernel', 'cgg_mask_logits_kernel', 'cgg_msda_fwd_stream2_f32_kernel')


def cp(a, b):
    if os.path.exists(os.path.join(src, a)):
        shutil.copyfile(os.path.join(src, a), os.path.join(dst, b))
        return True
    return False


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    if not os.path.exists(path):
        return agg
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        for f in FAM:
            if f in r['Kernel_Name']:
                agg[f].append(float(r['Counter_Value']))
    return agg


def aggregate(paths, out):
    agg = collections.defaultdict(list)
    for n, path in paths:
        for r in csv.DictReader(open(path)):
            if any(f in r['Kernel_Name'] for f in FAM):
                agg[(n, r['Kernel_Name'][:70], r.get('Grid_Size', ''), r['Counter_Name'])].append(float(r['Counter_Value']))
    with open(out, 'w', newline='') as fh:
        w = csv.writer(fh)
        w.writerow(['pass', 'kernel', 'grid_size', 'counter', 'dispatches', 'mean', 'min', 'max'])
        for k in sorted(agg):
            v = agg[k]
            w.writerow(list(k) + [len(v), '%.6g' % (sum(v) / len(v)), '%.6g' % min(v), '%.6g' % max(v)])


def workload(tag, name):
    """tag = cfg1 | cfg4 -> profiles/r6_<name>_*"""
    cp(f'{tag}_kernel_stats.csv', f'r6_{name}_bench_kernel_stats.csv')
    cp(f'{tag}_hot_kernel_launches.txt', f'r6_{name}_hot_kernel_launches.txt')
    cp(f'{tag}_step_kernels.txt', f'r6_{name}_step_kernels.txt')
    hot = {}
    p = os.path.join(src, f'{tag}_hot_kernel_launches.json')
    if os.path.exists(p):
        hot = json.load(open(p))
    fetch = per_kernel(os.path.join(src, f'{tag}_pmc_FETCH_SIZE.csv'), 'FETCH_SIZE')
    write = per_kernel(os.path.join(src, f'{tag}_pmc_WRITE_SIZE.csv'), 'WRITE_SIZE')
    cmd = (f'rocprofv3 --pmc <counter> -- python3 bench.py --workload {tag} --graph 0 --pipeline 0 --steps 4 --warmup 3 --repeats 1 (lean flags; '
           'separate passes for FETCH_SIZE and WRITE_SIZE; KB units; FETCH_SIZE x2: gfx950 tallies 128-B requests of wide coalesced reads at '
           '64 B, MI355X_MICROARCH.md "HBM")')
    steps_traced = 3 + 4 + 5 + 4      # warm-up + timed + latency + event re-run steps of the eager PMC command
    summary = {}
    for f in FAM:
        if f not in fetch or f not in write:
            continue
        fv, wv = fetch[f], write[f]
        if f == 'cgg_mask_logits_kernel':
            wv = [x for x in wv if x * 1024.0 > 20e6]         # the full-resolution launch: the one that WRITES the logits
            fv = sorted(fv, reverse=True)[:len(wv)]
            if not wv:
                continue
        fb = sum(fv) / len(fv) * 1024.0 * 2.0
        wb = sum(wv) / len(wv) * 1024.0
        rec = dict(traffic_bytes=fb + wb, fetch_bytes_x2_corrected=fb, write_bytes=wb, launches_in_trace=len(fv), source='committed profile: ' + cmd,
                   rocprof=hot.get(f if f != 'cgg_mask_logits_kernel' else 'cgg_mask_logits_kernel_full_resolution'))
        if f == 'cgg_gemm_x3s_kernel':
            per_step = len(fv) / steps_traced
            rec['launches_per_step'] = per_step
            rec['traffic_bytes_per_step'] = (fb + wb) * per_step
        summary[f] = rec
    sq = collections.defaultdict(lambda: collections.defaultdict(list))
    p = os.path.join(src, f'{tag}_pmc_sq.csv')
    if os.path.exists(p):
        for r in csv.DictReader(open(p)):
            for f in FAM:
                if f in r['Kernel_Name']:
                    sq[f][r['Counter_Name']].append(float(r['Counter_Value']))
        for f, d in sq.items():
            if f in summary:
                summary[f]['sq_means_per_launch'] = {k: sum(v) / len(v) for k, v in d.items()}
    paths = [(n, os.path.join(src, f'{tag}_pmc_{c}.csv')) for c, n in (('sq', 'sq'), ('FETCH_SIZE', 'fetch'), ('WRITE_SIZE', 'write'), ('grbm', 'grbm'))]
    aggregate([(n, q) for n, q in paths if os.path.exists(q)], os.path.join(dst, f'r6_{name}_pmc_counters_by_kernel.csv'))
    json.dump(summary, open(os.path.join(dst, f'r6_{name}_kernels.json'), 'w'), indent=1)
    p = os.path.join(dst, f'r6_{name}_bench_kernel_stats.csv')
    if os.path.exists(p):
        rows = list(csv.DictReader(open(p)))
        rows.sort(key=lambda r: -int(r['TotalDurationNs']))
        with open(os.path.join(dst, f'r6_{name}_bench_kernel_stats_top.txt'), 'w') as f:
            f.write(f'rocprofv3 --kernel-trace --stats -- python3 bench.py --workload {tag} --steps 20 --warmup 5 --no-cpu-baseline --no-bf16-mode '
                    '--host-results 0 --train-step 0 --extra-workloads 0 --no-einsum-sweep --repeats 3   (parity mode fp32 = f16 x 3 MFMA, hipGraph, staged pipeline, 1x MI355X)\n')
            f.write('whole process: eager warm-up + graph captures + 3 x 20 timed pipelined steps + 20 eager event-timed steps\n')
            f.write('%-100s %8s %12s %10s %7s\n' % ('kernel', 'calls', 'total_us', 'avg_us', 'pct'))
            for r in rows[:50]:
                f.write('%-100s %8d %12.1f %10.2f %7s\n' % (r['Name'][:100], int(r['Calls']), int(r['TotalDurationNs']) / 1e3,
                                                          float(r['AverageNs']) / 1e3, r['Percentage']))
        os.remove(p)            # the top-50 text is what is committed
    return summary


os.makedirs(dst, exist_ok=True)
cp('bench_line.json', 'r6_bench_line.json')
cp('bench_full.json', 'r6_bench_full.json')
s1 = workload('cfg1', 'fp32')
s4 = workload('cfg4', 'cfg4')
for w, n in (('cfg2', ''), ('cfg3', 'cfg3_')):
    for p in ('fp32', 'bf16'):
        cp(f'{w}_train_step_kernels_{p}.txt', f'r6_{n}train_step_kernels_{p}.txt')
m = os.path.join(src, 'msda_bwd')
if os.path.isdir(m):
    out = ['MSDeformAttn backward, round 6 (csrc/msda_bwd.hip two-pass sorted scatter + gather kernel), configs[2] shapes (B = 16, 21 504 queries, 8 heads x 32 channels, '
           '3 levels x 4 points), scratch/msda_bwd_only.py; counters: separate rocprofv3 --pmc passes, mean per launch', '']
    if os.path.exists(os.path.join(m, 'timing.txt')):
        out += open(os.path.join(m, 'timing.txt')).read().splitlines() + ['']
    ks = os.path.join(m, 'kernel_stats.csv')
    if os.path.exists(ks):
        out.append('kernel trace (offset std 2.0 px):')
        for r in list(csv.DictReader(open(ks)))[:6]:
            out.append('  %-70s calls %4d avg %10.1f us' % (r['Name'][:70], int(r['Calls']), float(r['AverageNs']) / 1e3))
        out.append('')
    for f in sorted(os.listdir(m)):
        if f.startswith('pmc_'):
            agg = collections.defaultdict(lambda: collections.defaultdict(list))
            for r in csv.DictReader(open(os.path.join(m, f))):
                agg[r['Kernel_Name'].split('(')[0][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
            for k, d in agg.items():
                out.append('%s  %s: ' % (f, k) + ', '.join('%s %.4g' % (c, sum(v) / len(v)) for c, v in sorted(d.items())) + '  (%d launches)' % max(len(v) for v in d.values()))
    open(os.path.join(dst, 'r6_msda_bwd_pmc.txt'), 'w').write('\n'.join(out) + '\n')
# (the agreement records are published by tools/collect_agreement.sh, which checks that all of them are there)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != 'source'} for k, v in s1.items()}, indent=1)[:2500])
