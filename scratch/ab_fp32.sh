#!/bin/bash
# A/B of fp32-mode bench lines inside one call: each argument is an env assignment list, e.g. "CGG_FUSED_PROJ=0"
cd /root/repo
for cfg in "$@"; do
  for rep in 1 2; do
    v=$(env $cfg python bench.py --precision fp32 --no-cpu-baseline --host-results 0 --no-bf16-mode --train-step 0 --no-einsum-sweep --repeats 5 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('%.1f img/s %.3f ms/step lat %.2f' % (d['value'],d['ms_per_step'],d['latency_ms_per_batch']))")
    echo "$cfg : $v"
  done
done
