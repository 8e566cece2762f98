#!/bin/bash
# round 5: the sorted-scatter MSDeformAttn backward (csrc/msda_bwd.hip) at configs[2] shapes: timing of the workgroup / phase-B variants
# and three offset spreads, kernel trace
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/r5/msda_bwd
mkdir -p $O
: > $O/timing.txt
for v in ${VARS:-0 1 2 3}; do
  for std in 0.5 2.0 8.0; do
    echo -n "var=$v " >> $O/timing.txt
    CGG_MSDA_BWD_VAR=$v python3 $R/scratch/msda_bwd_only.py $std 5 2>&1 | grep -v amdgpu | grep host-level >> $O/timing.txt
  done
done
cat $O/timing.txt
