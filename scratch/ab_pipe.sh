#!/bin/bash
# A/B of the software-pipeline depth on the fp32 step (same box)
for rep in 1 2; do
for p in 3 4 5 2; do
  python bench.py --pipeline $p --no-cpu-baseline --no-bf16-mode --host-results 0 --train-step 0 --no-einsum-sweep 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pipeline=$p', round(d['value'],1), 'img/s', round(d.get('latency_ms_per_batch',0),2), 'ms latency')"
done
done
