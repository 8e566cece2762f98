#!/bin/bash
# pipelined parity-mode step only, one run per env assignment (A/B of many variants in one call)
cd /root/repo
for e in "$@"; do
env $e python3 bench.py --pipeline 3 --steps 20 --warmup 5 --no-cpu-baseline --no-bf16-mode --host-results 0 --train-step 0 --no-einsum-sweep --repeats 7 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$e]', round(d['value'],1), round(d['ms_per_step'],3))"
done
