#!/bin/bash
# the whole GPU suite on a fresh box, complete output kept
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
python -m pytest tests -q -m gpu -x > gpurun_out/r5/pytest_gpu_full.txt 2>&1
tail -5 gpurun_out/r5/pytest_gpu_full.txt
grep -n "rank . failed\|Error\|error:" gpurun_out/r5/pytest_gpu_full.txt | head -20
