#!/bin/bash
# A/B of the pipelined / unpipelined parity-mode step: env assignments given as arguments, e.g. "CGG_X3A=1" "CGG_X3A=0"
cd /root/repo
for rep in 1 2; do for e in "$@"; do for p in 3 0; do
env $e python3 bench.py --pipeline $p --steps 20 --warmup 5 --no-cpu-baseline --no-bf16-mode --host-results 0 --train-step 0 --no-einsum-sweep --repeats 5 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('pipeline $p [$e]', round(d['value'],1), round(d['ms_per_step'],3), 'gemm ms/step', round(d['roofline']['ms_per_step'],3), 'frac', round(d['roofline']['frac'],3))"
done; done; done
