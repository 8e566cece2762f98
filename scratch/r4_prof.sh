#!/bin/bash
# round 4: per-step kernel table of the eager parity-mode step, x3a stream on (default) / off
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/r4
mkdir -p $O
for v in 1 0; do
  export CGG_X3A=$v
  rm -rf /tmp/fp32prof$v
  rocprofv3 --kernel-trace --output-format csv -d /tmp/fp32prof$v -- python3 $R/bench.py --precision fp32 --graph 0 --pipeline 0 --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --host-results 0 --no-bf16-mode --train-step 0 --no-einsum-sweep > $O/fp32_eager_under_rocprof_x3a$v.log 2>&1
  python3 $R/scratch/step_kernels2.py /tmp/fp32prof$v cgg_class_topk 8 > $O/fp32_step_kernels_x3a$v.txt 2>&1
done
head -45 $O/fp32_step_kernels_x3a1.txt
head -12 $O/fp32_step_kernels_x3a0.txt
unset CGG_X3A
cd $R
for p in 0 3; do for v in 1 0; do
CGG_X3A=$v python3 bench.py --pipeline $p --steps 20 --warmup 5 --no-cpu-baseline --no-bf16-mode --host-results 0 --train-step 0 --no-einsum-sweep --repeats 5 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('pipeline $p X3A=$v', d['value'], d['ms_per_step'])"
done; done
