#!/bin/bash
# round 4: per-step kernel table of the PARITY-mode (fp32-class) training step, then the bf16 one
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/r4
mkdir -p $O
for p in fp32 bf16; do
rm -rf /tmp/trainprof_$p
rocprofv3 --kernel-trace --output-format csv -d /tmp/trainprof_$p -- python3 $R/bench.py --mode train --steps 4 --warmup 2 --precision $p > $O/train_under_rocprof_$p.log 2>&1
python3 $R/scratch/step_kernels2.py /tmp/trainprof_$p NormTwoOps 3 7 > $O/train_step_kernels_$p.txt 2>&1
done
head -48 $O/train_step_kernels_fp32.txt | cut -c1-160
