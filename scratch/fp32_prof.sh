#!/bin/bash
# fp32 (parity) mode: per-step kernel table of the eager step + the bench line in that mode
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/r3
mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d /tmp/fp32prof -- python3 $R/bench.py --precision fp32 --graph 0 --pipeline 0 --steps 10 --warmup 3 --no-cpu-baseline --host-results 0 > $O/fp32_eager_under_rocprof.log 2>&1
python3 $R/scratch/step_kernels2.py /tmp/fp32prof > $O/fp32_step_kernels.txt 2>&1
cd $R
python bench.py --precision fp32 --no-cpu-baseline --host-results 0 > $O/fp32_bench_line.json 2> $O/fp32_bench_line.err
head -70 $O/fp32_step_kernels.txt
tail -c 600 $O/fp32_bench_line.json
