"""Per-launch durations of parity mode's hot kernels from the rocprofv3 kernel trace of the bench command (argv[1] = trace dir;
argv[2] = json out): the cross-check of bench.py's event-timed `roofline` / `kernels` objects."""
import csv, glob, collections, json, sys, os
f = max(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    for key in ('cgg_gemm_x3s_kernel', 'cgg_gemm_x3_kernel', 'cgg_encoder_tail_x3_kernel', 'cgg_mask_logits_kernel', 'cgg_msda_fwd_stream2_f32_kernel', 'cgg_decoder_mid_kernel',
                'cgg_decoder_tail_kernel', 'cgg_decoder_ffn_kernel', 'cgg_xattn_partial_f32'):
        if key in n:
            if key == 'cgg_gemm_x3_kernel' and 'cgg_gemm_x3s_kernel' in n:
                continue
            inst = n[n.index(key):].split('(')[0]
            agg[(inst, r['Grid_Size_X'], r['Grid_Size_Y'])].append(d)
cmd = 'rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-bf16-mode --host-results 0 --train-step 0 --no-einsum-sweep --repeats 3'
print('Per-launch durations from the kernel trace of `%s`\n' % cmd)
print('%-46s %10s %8s %8s %10s %10s %10s' % ('kernel', 'grid_x', 'grid_y', 'calls', 'avg_us', 'min_us', 'max_us'))
for k, v in sorted(agg.items(), key=lambda kv: (kv[0][0], int(kv[0][1]))):
    print('%-46s %10s %8s %8d %10.2f %10.2f %10.2f' % (k[0][:46], k[1], k[2], len(v), sum(v) / len(v), min(v), max(v)))
fam = collections.defaultdict(list)
for k, v in agg.items():
    fam[k[0].split('<')[0]] += v
print('\nfamilies:')
for k, v in fam.items():
    print('%-46s calls %6d total %10.1f us avg %8.2f us' % (k, len(v), sum(v), sum(v) / len(v)))
if len(sys.argv) > 2:
    out = {}
    for k, v in fam.items():
        out[k] = dict(launch_ms_mean=sum(v) / len(v) / 1e3, launches=len(v), total_ms=sum(v) / 1e3, command=cmd)
    full = [v for k, v in agg.items() if k[0].startswith('cgg_mask_logits_kernel') and int(k[1]) == 65536]
    if full:
        v = full[0]
        out['cgg_mask_logits_kernel_full_resolution'] = dict(launch_ms_mean=sum(v) / len(v) / 1e3, launches=len(v), command=cmd)
    json.dump(out, open(sys.argv[2], 'w'), indent=1)
