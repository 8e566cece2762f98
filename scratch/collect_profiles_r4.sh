#!/bin/bash
# Round-4 profile collection on the GPU box (parity mode = the headline): rocprofv3 kernel stats of the bench command, the eager
# per-step kernel table, separate PMC passes (FETCH_SIZE / WRITE_SIZE / SQ) of the eager step, the bench lines.
# Results -> gpurun_out/r4/prof ; scratch/publish_profiles_r4.py copies the summaries into profiles/.
set -x
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/r4/prof
rm -rf $O && mkdir -p $O
BENCH="$R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-bf16-mode --host-results 0 --train-step 0 --no-einsum-sweep --repeats 3"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stats -- python3 $BENCH > $O/bench_under_rocprof.log 2>&1
cp $(find /tmp/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python3 $R/scratch/r4_launches.py /tmp/stats $O/hot_kernel_launches.json > $O/hot_kernel_launches.txt
EAGER="$R/bench.py --graph 0 --pipeline 0 --steps 4 --warmup 3 --repeats 1 --no-cpu-baseline --no-bf16-mode --host-results 0 --train-step 0 --no-einsum-sweep"
rocprofv3 --kernel-trace --output-format csv -d /tmp/stepprof -- python3 $R/bench.py --graph 0 --pipeline 0 --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-bf16-mode --host-results 0 --train-step 0 --no-einsum-sweep > /dev/null 2>&1
python3 $R/scratch/step_kernels2.py /tmp/stepprof cgg_class_topk 8 > $O/step_kernels.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -- python3 $EAGER > /dev/null 2>&1
  cp $(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1) $O/pmc_$c.csv
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pmc_sq -- python3 $EAGER > /dev/null 2>&1
cp $(find /tmp/pmc_sq -name "*counter_collection.csv" | head -1) $O/pmc_sq.csv
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/pmc_l2 -- python3 $EAGER > /dev/null 2>&1
cp $(find /tmp/pmc_l2 -name "*counter_collection.csv" | head -1) $O/pmc_l2.csv
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_grbm -- python3 $EAGER > /dev/null 2>&1
cp $(find /tmp/pmc_grbm -name "*counter_collection.csv" | head -1) $O/pmc_grbm.csv
cd $R
python bench.py > $O/bench_line.json 2> $O/bench_line.err
python bench.py --pipeline 0 --no-cpu-baseline --no-bf16-mode --host-results 0 --train-step 0 --no-einsum-sweep > $O/bench_line_nopipeline.json 2>> $O/bench_line.err
python bench.py --graph 0 --no-cpu-baseline --no-bf16-mode --host-results 0 --train-step 0 --no-einsum-sweep > $O/bench_line_eager.json 2>> $O/bench_line.err
python scratch/x3s_bench.py auto > $O/x3s_gemm_bench.txt 2>> $O/bench_line.err
python scratch/x3s_cold2.py > $O/x3s_cold_bench.txt 2>> $O/bench_line.err
python scratch/tail_x3_bench.py > $O/tail_x3_bench.txt 2>> $O/bench_line.err
ls -la $O
tail -c 300 $O/bench_line.json
