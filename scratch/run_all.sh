#!/bin/bash
# whole GPU suite + profile refresh in one box call
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r5/pytest_gpu.txt
cat gpurun_out/r5/pytest_gpu.txt
bash scratch/collect_profiles_r5.sh line cfg1 cfg4 train msda > gpurun_out/r5/collect.log 2>&1
tail -5 gpurun_out/r5/collect.log
