#!/bin/bash
# training step (configs[2], bf16): per-step kernel table of the last steps of a short run (the first steps hold MIOpen's find phase)
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/r3
mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d /tmp/trainprof -- python3 $R/bench.py --mode train --steps 4 --warmup 2 --precision bf16 > $O/train_under_rocprof.log 2>&1
python3 $R/scratch/step_kernels2.py /tmp/trainprof NormTwoOps 3 7 > $O/train_step_kernels.txt 2>&1
head -70 $O/train_step_kernels.txt
