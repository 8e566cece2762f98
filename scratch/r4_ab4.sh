#!/bin/bash
# usage: r4_ab4.sh "<pipeline> <env>" ...
cd /root/repo
for pe in "$@"; do
p=${pe%% *}; e=${pe#* }
env $e python3 bench.py --pipeline $p --steps 20 --warmup 5 --no-cpu-baseline --no-bf16-mode --host-results 0 --train-step 0 --no-einsum-sweep --repeats 7 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[pipeline $p $e]', round(d['value'],1), round(d['ms_per_step'],3))"
done
