#!/bin/bash
# round 4: where the scatter-only tiled MSDeformAttn backward spends its time. CGG_MSDA_BWD_ABL (1 = no flush, 2 = no scatter loop) was a
# TEMPORARY build switch (removed again after the measurement, profiles/r4_msda_bwd_split_ablation.txt): without it all four passes
# time the shipped kernels.
cd /tmp && export TMPDIR=/tmp
R=/root/repo
for a in 0 1 2 3; do
  export CGG_MSDA_BWD_ABL=$a
  rm -rf /tmp/mb_abl
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mb_abl -- python3 $R/scratch/msda_bwd_only.py ${1:-2.0} 5 > /dev/null 2>&1
  echo "ABL=$a"; python3 -c "
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'msda' in r['Name']: print('  %-50s calls %s avg %.1f us' % (r['Name'][:50], r['Calls'], float(r['AverageNs'])/1e3))
" $(find /tmp/mb_abl -name "*kernel_stats.csv" | head -1)
done
