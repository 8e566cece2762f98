"""Per-launch durations of the roofline kernels from a rocprofv3 kernel trace, split by grid size."""
import csv, glob, collections, json, sys, os
f = max(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if 'cgg_mask_logits_kernel' in n:
        agg[('cgg_mask_logits_kernel', r['Grid_Size_X'], r['Grid_Size_Y'])].append(d)
    elif 'cgg_msda_fwd_stream' in n:
        agg[('cgg_msda_fwd_stream_kernel', r['Grid_Size_X'], r['Grid_Size_Y'])].append(d)
    elif 'cgg_encoder_ffn_ln_kernel<false, true>' in n:
        agg[('cgg_encoder_ffn_ln_kernel<false, true>', r['Grid_Size_X'], r['Grid_Size_Y'])].append(d)
    elif 'cgg_encoder_ffn_ln_kernel<true, true>' in n:
        agg[('cgg_encoder_ffn_ln_kernel<true, true>', r['Grid_Size_X'], r['Grid_Size_Y'])].append(d)
    elif 'cgg_encoder_proj_kernel' in n:
        agg[('cgg_encoder_proj_kernel', r['Grid_Size_X'], r['Grid_Size_Y'])].append(d)
print('Per-launch durations from the rocprofv3 kernel trace of `python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-mode --host-results 0`')
print('(same run as r2_bench_kernel_stats.csv; the stats file averages the 9 bit-mask-only launches per step together')
print(' with the ONE full-resolution launch that bench.py prices in `roofline`, so they are split here by grid size)\n')
print('%-40s %10s %8s %8s %10s %10s %10s' % ('kernel', 'grid_x', 'grid_y', 'calls', 'avg_us', 'min_us', 'max_us'))
for k, v in sorted(agg.items(), key=lambda kv: (kv[0][0], int(kv[0][1]))):
    print('%-40s %10s %8s %8d %10.2f %10.2f %10.2f' % (k[0], k[1], k[2], len(v), sum(v) / len(v), min(v), max(v)))
print('\ncgg_mask_logits_kernel grid_x = 65536 threads (128 workgroups x 512) x 2 images = the full-resolution launch:')
print('algorithmic bytes 119 742 464 per launch (bf16 packed feature 67.1 MB + mask_embed 0.2 MB + f32 logits 52.4 MB).')

if len(sys.argv) > 2:
    # machine-readable per-launch means of the launches bench.py prices in `roofline` / `kernels` (its cross-check)
    out = {}
    for k, v in agg.items():
        full = k[0] == 'cgg_mask_logits_kernel' and int(k[1]) == 65536 and int(k[2]) == 2
        if full or k[0].startswith('cgg_msda_fwd') or k[0].startswith('cgg_encoder_'):
            name = 'cgg_msda_fwd_stream_kernel' if k[0].startswith('cgg_msda_fwd') else k[0]
            out[name] = dict(launch_ms_mean=sum(v) / len(v) / 1e3, launch_ms_min=min(v) / 1e3, launch_ms_max=max(v) / 1e3,
                             launches=len(v), grid=[int(k[1]), int(k[2])],
                             command='rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-mode --host-results 0')
    json.dump(out, open(sys.argv[2], 'w'), indent=1)
