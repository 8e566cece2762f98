#!/bin/bash
# usage: r4_ab5.sh "<bench args>" ...
cd /root/repo
for a in "$@"; do
python3 bench.py $a --steps 20 --warmup 5 --no-cpu-baseline --no-bf16-mode --host-results 0 --train-step 0 --no-einsum-sweep --repeats 7 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$a]', round(d['value'],1), round(d['ms_per_step'],3))"
done
