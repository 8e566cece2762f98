"""gpurun_out/r4/prof/* (scratch/collect_profiles_r4.sh on the GPU box) -> profiles/r4_* + profiles/r4_fp32_kernels.json, the
summary bench.py reads for its `traffic` / `rocprof` fields (labelled "committed profile")."""
import collections, csv, json, os, shutil
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
src, dst = os.path.join(R, 'gpurun_out', 'r4', 'prof'), os.path.join(R, 'profiles')
pairs = {'kernel_stats.csv': 'r4_fp32_bench_kernel_stats.csv', 'hot_kernel_launches.txt': 'r4_fp32_hot_kernel_launches.txt',
         'hot_kernel_launches.json': 'r4_fp32_hot_kernel_launches.json', 'step_kernels.txt': 'r4_fp32_step_kernels.txt',
         'bench_line.json': 'r4_bench_line.json', 'bench_line_nopipeline.json': 'r4_bench_line_nopipeline.json',
         'bench_line_eager.json': 'r4_bench_line_eager.json', 'x3s_gemm_bench.txt': 'r4_x3s_gemm_bench.txt',
         'tail_x3_bench.txt': 'r4_tail_x3_bench.txt', 'x3s_cold_bench.txt': 'r4_x3s_cold_bench.txt', 'bf16_msda_pmc_FETCH_SIZE.csv': 'r4_bf16_msda_hm_pmc_fetch.csv',
         'bf16_msda_pmc_WRITE_SIZE.csv': 'r4_bf16_msda_hm_pmc_write.csv', 'bf16_msda_pmc_sq.csv': 'r4_bf16_msda_hm_pmc_sq_l2.csv'}
for a, b in pairs.items():
    if os.path.exists(os.path.join(src, a)):
        shutil.copyfile(os.path.join(src, a), os.path.join(dst, b))
FAM = ('cgg_gemm_x3s_kernel', 'cgg_encoder_tail_x3_kernel', 'cgg_mask_logits_kernel', 'cgg_msda_fwd_stream2_f32_kernel')


def per_kernel(path, counter):
    """mean counter value per launch and launches for each family, over ALL dispatches of the traced process"""
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        for f in FAM:
            if f in r['Kernel_Name']:
                agg[f].append((float(r['Counter_Value']), int(r.get('Grid_Size') or 0)))
    return agg


summary = {}
steps_traced = 3 + 4 + 5 + 4      # warm-up + timed + latency + event re-run steps of the eager PMC command
fetch = per_kernel(os.path.join(src, 'pmc_FETCH_SIZE.csv'), 'FETCH_SIZE')
write = per_kernel(os.path.join(src, 'pmc_WRITE_SIZE.csv'), 'WRITE_SIZE')
cmd = ('rocprofv3 --pmc <counter> -- python3 bench.py --graph 0 --pipeline 0 --steps 4 --warmup 3 --repeats 1 --no-cpu-baseline '
       '--no-bf16-mode --host-results 0 --train-step 0 (separate passes for FETCH_SIZE and WRITE_SIZE; KB units; FETCH_SIZE x2: '
       'gfx950 tallies 128-B requests of wide coalesced reads at 64 B, MI355X_MICROARCH.md "HBM")')
hot = json.load(open(os.path.join(src, 'hot_kernel_launches.json'))) if os.path.exists(os.path.join(src, 'hot_kernel_launches.json')) else {}
for f in FAM:
    if f not in fetch or f not in write:
        continue
    fv, wv = fetch[f], write[f]
    if f == 'cgg_mask_logits_kernel':
        # the full-resolution launch only: the one that WRITES the logits (> 30 MB); it is also the one that fetches most
        wv = [x for x in wv if x[0] * 1024.0 > 30e6]
        fv = sorted(fv, reverse=True)[:len(wv)]
    fb = sum(v for v, _ in fv) / len(fv) * 1024.0 * 2.0
    wb = sum(v for v, _ in wv) / len(wv) * 1024.0
    rec = dict(traffic_bytes=fb + wb, fetch_bytes_x2_corrected=fb, write_bytes=wb, launches_in_trace=len(fv), source='committed profile: ' + cmd,
               rocprof=hot.get(f if f != 'cgg_mask_logits_kernel' else 'cgg_mask_logits_kernel_full_resolution'))
    if f == 'cgg_gemm_x3s_kernel':
        per_step = len(fv) / steps_traced
        rec['launches_per_step'] = per_step
        rec['traffic_bytes_per_step'] = (fb + wb) * per_step
    summary[f] = rec
# MFMA busy from the SQ pass
sq = collections.defaultdict(lambda: collections.defaultdict(list))
p = os.path.join(src, 'pmc_sq.csv')
if os.path.exists(p):
    for r in csv.DictReader(open(p)):
        for f in FAM:
            if f in r['Kernel_Name']:
                sq[f][r['Counter_Name']].append(float(r['Counter_Value']))
    for f, d in sq.items():
        if f in summary:
            summary[f]['sq_means_per_launch'] = {k: sum(v) / len(v) for k, v in d.items()}


def aggregate(paths, out):
    """ONE committed table for all counter passes: per (kernel instantiation, grid, counter) the launch count, mean, min and max of the
    per-dispatch values of the four kernel families (the raw per-dispatch passes are 1.3-10 MB each and stay in gpurun_out/)"""
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


paths = [(n, os.path.join(src, f'pmc_{c}.csv')) for c, n in (('sq', 'sq'), ('FETCH_SIZE', 'fetch'), ('WRITE_SIZE', 'write'), ('l2', 'l2'), ('grbm', 'grbm'))]
aggregate([(n, p) for n, p in paths if os.path.exists(p)], os.path.join(dst, 'r4_fp32_pmc_counters_by_kernel.csv'))
json.dump(summary, open(os.path.join(dst, 'r4_fp32_kernels.json'), 'w'), indent=1)
rows = list(csv.DictReader(open(os.path.join(dst, 'r4_fp32_bench_kernel_stats.csv'))))
rows.sort(key=lambda r: -int(r['TotalDurationNs']))
with open(os.path.join(dst, 'r4_fp32_bench_kernel_stats_top.txt'), 'w') as f:
    f.write('rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-bf16-mode --host-results 0 --train-step 0 --no-einsum-sweep --repeats 3   (parity mode fp32 = f16 x 3 MFMA, hipGraph, 3-stage pipeline, 1x MI355X)\n')
    f.write('whole process: eager warm-up (incl. the MIOpen solver search of the stem) + graph captures + 3 x 20 timed pipelined steps + 20 eager event-timed steps\n')
    f.write('%-100s %8s %12s %10s %7s\n' % ('kernel', 'calls', 'total_us', 'avg_us', 'pct'))
    for r in rows[:50]:
        f.write('%-100s %8d %12.1f %10.2f %7s\n' % (r['Name'][:100], int(r['Calls']), int(r['TotalDurationNs']) / 1e3,
                                                  float(r['AverageNs']) / 1e3, r['Percentage']))
print(json.dumps(summary, indent=1)[:3000])
