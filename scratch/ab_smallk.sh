#!/bin/bash
# A/B of the x3 GEMM tile heuristic (CGG_XG_SMALLK) on the pipelined fp32 step, same box
for rep in 1 2; do
for k in 512 0 256 1024; do
  CGG_XG_SMALLK=$k python bench.py --no-cpu-baseline --no-bf16-mode --host-results 0 --train-step 0 --no-einsum-sweep 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('smallk=$k', round(d['value'],1), 'img/s', round(d['roofline']['ms_per_step'],3), 'ms gemm/step')"
done
done
