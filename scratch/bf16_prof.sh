#!/bin/bash
# bf16 (throughput) mode: per-step kernel table of the eager step
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/r3
mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d /tmp/bf16prof -- python3 $R/bench.py --precision bf16 --graph 0 --pipeline 0 --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --host-results 0 --train-step 0 --no-einsum-sweep > $O/bf16_eager_under_rocprof.log 2>&1
python3 $R/scratch/step_kernels2.py /tmp/bf16prof cgg_class_topk 8 > $O/bf16_step_kernels.txt 2>&1
head -40 $O/bf16_step_kernels.txt
