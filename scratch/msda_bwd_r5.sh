#!/bin/bash
# round 5: the sorted-scatter MSDeformAttn backward (csrc/msda_bwd.hip) at configs[2] shapes: timing for tile edges c = 2 / 4 and three
# offset spreads, kernel trace, PMC passes (separate, as the guide prescribes)
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/r5/msda_bwd
mkdir -p $O
: > $O/timing.txt
for c in 4 2; do
  for std in 0.5 2.0 8.0; do
    echo -n "c=$c " >> $O/timing.txt
    CGG_MSDA_BWD_C=$c python3 $R/scratch/msda_bwd_only.py $std 5 2>&1 | grep -v amdgpu >> $O/timing.txt
  done
done
for c in 4 2; do
  CGG_MSDA_BWD_C=$c rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mb_kt$c -- python3 $R/scratch/msda_bwd_only.py 2.0 5 > /dev/null 2>&1
  cp $(find /tmp/mb_kt$c -name "*kernel_stats.csv" | head -1) $O/kernel_stats_c$c.csv
done
cat $O/timing.txt
for c in 4 2; do echo "c=$c"; head -8 $O/kernel_stats_c$c.csv | cut -c1-200; done
