#!/bin/bash
# A/B of stream priorities of the three pipeline stages (CGG_PIPE_PRIO), fp32 step, same box
for rep in 1 2; do
for p in "0,0,0" "0,0,-1" "-1,0,0" "0,-1,0" "0,-1,-1" "-1,-1,0"; do
  CGG_PIPE_PRIO=$p python bench.py --no-cpu-baseline --no-bf16-mode --host-results 0 --train-step 0 --no-einsum-sweep 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('prio=$p', round(d['value'],1), 'img/s', round(d.get('latency_ms_per_batch',0),2), 'ms latency')"
done
done
