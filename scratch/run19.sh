#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tp3
rocprofv3 --kernel-trace --output-format csv -d /tmp/tp3 -- python3 $R/bench.py --workload cfg3 --steps 4 --warmup 2 --precision fp32 > /dev/null 2>&1
python3 $R/scratch/step_kernels2.py /tmp/tp3 NormTwoOps 3 auto:7 2>&1 | head -40 | cut -c1-170
