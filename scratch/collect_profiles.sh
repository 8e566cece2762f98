#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel stats of the bench command + separate PMC passes of the hot kernels + bench lines.
set -x
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/r2
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-mode --host-results 0 > $O/bench_under_rocprof.log 2>&1
cp $(find /tmp/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python3 $R/scratch/hot_launches.py /tmp/stats $O/hot_kernel_launches.json > $O/hot_kernel_launches.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -- python3 $R/scratch/kernel_only.py > /dev/null 2>&1
  cp $(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1) $O/pmc_$c.csv
done
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/pmc_l2 -- python3 $R/scratch/kernel_only.py > /dev/null 2>&1
cp $(find /tmp/pmc_l2 -name "*counter_collection.csv" | head -1) $O/pmc_l2.csv
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_sq -- python3 $R/scratch/kernel_only.py > /dev/null 2>&1
cp $(find /tmp/pmc_sq -name "*counter_collection.csv" | head -1) $O/pmc_sq.csv
cd $R
python bench.py > $O/bench_line.json 2> $O/bench_line.err
python bench.py --pipeline 0 --no-cpu-baseline > $O/bench_line_nopipeline.json 2>> $O/bench_line.err
python bench.py --graph 0 --no-cpu-baseline > $O/bench_line_eager.json 2>> $O/bench_line.err
python bench.py --mode train --steps 5 --warmup 3 > $O/bench_train_line.json 2>> $O/bench_line.err
python scratch/kernel_bench.py > $O/kernel_bench.json 2>> $O/bench_line.err
python scratch/xattn_bwd_bench.py > $O/xattn_bwd_bench.txt 2>> $O/bench_line.err
python scratch/ml_bwd_bench.py > $O/ml_bwd_bench.txt 2>> $O/bench_line.err
python scratch/msda_ab.py > $O/msda_ab.txt 2>> $O/bench_line.err
python scratch/ffn_bench.py > $O/encoder_ffn_bench.txt 2>> $O/bench_line.err
python scratch/proj_bench.py > $O/encoder_proj_bench.txt 2>> $O/bench_line.err
python scratch/serve_bench.py 128 2>> $O/bench_line.err | grep images > $O/serve_bench.txt   # one process per result format
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/train_prof -- python3 $R/bench.py --mode train --steps 3 --warmup 2 > /dev/null 2>&1
python3 $R/scratch/train_step_prof.py /tmp/train_prof > $O/train_top.txt
rocprofv3 --kernel-trace --output-format csv -d /tmp/stepprof -- python3 $R/bench.py --graph 0 --pipeline 0 --steps 20 --warmup 5 --no-cpu-baseline --no-parity-mode --host-results 0 > /dev/null 2>&1
python3 $R/scratch/step_kernels.py /tmp/stepprof > $O/step_kernels.txt
cd $R
ls -la $O
tail -c 400 $O/bench_line.json
