#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel stats of the bench command + separate PMC passes of the hot kernels.
set -x
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/r1
mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/scratch/kernel_only.py > $O/pmc_$c.log 2>&1
done
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_l2 -- python3 $R/scratch/kernel_only.py > $O/pmc_l2.log 2>&1
cd $R
python bench.py > $O/bench_line.json 2> $O/bench_line.err
find $O -name "*.csv" | head -20
tail -c 600 $O/bench_line.json
