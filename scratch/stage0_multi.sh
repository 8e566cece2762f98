#!/bin/bash
# usage: stage0_multi.sh "<env 0>" "<env 1>" ...   backbone graph, per-launch table of every variant vs the LAST one
cd /tmp && export TMPDIR=/tmp
R=/root/repo
i=0
for e in "$@"; do
  rm -rf /tmp/st0_$i
  env $e rocprofv3 --kernel-trace --output-format csv -d /tmp/st0_$i -- python3 $R/scratch/stage_times.py 3 > /tmp/st0_$i.log 2>&1
  i=$((i+1))
done
n=$((i-1))
for k in $(seq 0 $((n-1))); do
  eval "ek=\${$((k+1))}"
  echo "=== A = [$ek]   B = [${!i}]"
  python3 $R/scratch/stage0_seq.py /tmp/st0_$k /tmp/st0_$n | awk '{print}' | cut -c1-150
done
