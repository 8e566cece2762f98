#!/bin/bash
# A/B of the x3 GEMM tile-shape thresholds on the pipelined fp32 step (same box): CGG_XG_MINTILES x CGG_XG_SMALLK
for rep in 1 2; do
for cfg in "384 512" "192 0" "160 0" "128 0" "112 0" "96 0"; do
  set -- $cfg
  CGG_XG_MINTILES=$1 CGG_XG_SMALLK=$2 python bench.py --no-cpu-baseline --no-bf16-mode --host-results 0 --train-step 0 --no-einsum-sweep 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('mintiles=$1 smallk=$2', round(d['value'],1), 'img/s', round(d['roofline']['ms_per_step'],3), 'ms gemm/step (eager events)', round(d['latency_ms_per_batch'],2) if 'latency_ms_per_batch' in d else '')"
done
done
