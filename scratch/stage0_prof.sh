#!/bin/bash
# usage: stage0_prof.sh "<env A>" "<env B>"   per-launch backbone table, A vs B
cd /tmp && export TMPDIR=/tmp
R=/root/repo
i=0
for e in "$1" "$2"; do
  rm -rf /tmp/st0_$i
  env $e rocprofv3 --kernel-trace --output-format csv -d /tmp/st0_$i -- python3 $R/scratch/stage_times.py 3 > /tmp/st0_$i.log 2>&1
  i=$((i+1))
done
echo "A = [$1]   B = [$2]"
python3 $R/scratch/stage0_seq.py /tmp/st0_0 /tmp/st0_1
