"""Copy gpurun_out/r2/* (written by scratch/collect_profiles.sh on the GPU box) into profiles/r2_* and write the
readable top-kernel table."""
import csv, os, shutil, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
src, dst = os.path.join(R, 'gpurun_out', 'r2'), os.path.join(R, 'profiles')
pairs = {'kernel_stats.csv': 'r2_bench_kernel_stats.csv', 'hot_kernel_launches.txt': 'r2_hot_kernel_launches.txt',
         'pmc_FETCH_SIZE.csv': 'r2_pmc_fetch_counter_collection.csv', 'pmc_WRITE_SIZE.csv': 'r2_pmc_write_counter_collection.csv',
         'pmc_l2.csv': 'r2_pmc_l2_counter_collection.csv', 'pmc_sq.csv': 'r2_pmc_sq_counter_collection.csv', 'bench_line.json': 'r2_bench_line.json',
         'bench_line_nopipeline.json': 'r2_bench_line_nopipeline.json', 'bench_line_eager.json': 'r2_bench_line_eager.json',
         'bench_train_line.json': 'r2_bench_train_line.json', 'hot_kernel_launches.json': 'r2_hot_kernel_launches.json',
         'kernel_bench.json': 'r2_kernel_bench.json', 'xattn_bwd_bench.txt': 'r2_xattn_bwd_bench.txt',
         'ml_bwd_bench.txt': 'r2_ml_bwd_bench.txt', 'msda_ab.txt': 'r2_msda_ab.txt', 'serve_bench.txt': 'r2_serve_bench.txt',
         'train_top.txt': 'r2_train_step_top_kernels.txt', 'encoder_ffn_bench.txt': 'r2_encoder_ffn_bench.txt',
         'encoder_proj_bench.txt': 'r2_encoder_proj_bench.txt', 'step_kernels.txt': 'r2_inference_step_kernels.txt'}
for a, b in pairs.items():
    if os.path.exists(os.path.join(src, a)):
        shutil.copyfile(os.path.join(src, a), os.path.join(dst, b))
rows = list(csv.DictReader(open(os.path.join(dst, 'r2_bench_kernel_stats.csv'))))
rows.sort(key=lambda r: -int(r['TotalDurationNs']))
with open(os.path.join(dst, 'r2_bench_kernel_stats_top.txt'), 'w') as f:
    f.write('rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-mode --host-results 0   (bf16, hipGraph, 3-stage pipeline, 1x MI355X)\n')
    f.write('whole process: eager warm-up (incl. the MIOpen solver search) + graph captures + 20 timed pipelined steps + 20 eager event-timed steps\n')
    f.write('(kernel tracing serialises the pipeline streams: ms_per_step under the profiler is ~1.3 ms above the unprofiled step)\n')
    f.write('%-100s %8s %12s %10s %7s\n' % ('kernel', 'calls', 'total_us', 'avg_us', 'pct'))
    for r in rows[:60]:
        f.write('%-100s %8d %12.1f %10.2f %7s\n' % (r['Name'][:100], int(r['Calls']), int(r['TotalDurationNs']) / 1e3,
                                                  float(r['AverageNs']) / 1e3, r['Percentage']))
print(open(os.path.join(dst, 'r2_bench_kernel_stats_top.txt')).read()[:3000])
