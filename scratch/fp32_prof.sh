#!/bin/bash
# fp32 (parity) mode: per-step kernel table of the eager step
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/r3
mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d /tmp/fp32prof -- python3 $R/bench.py --precision fp32 --graph 0 --pipeline 0 --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --host-results 0 --no-bf16-mode --train-step 0 --no-einsum-sweep > $O/fp32_eager_under_rocprof.log 2>&1
python3 $R/scratch/step_kernels2.py /tmp/fp32prof cgg_class_topk 8 > $O/fp32_step_kernels.txt 2>&1
head -60 $O/fp32_step_kernels.txt
