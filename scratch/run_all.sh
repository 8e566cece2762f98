#!/bin/bash
# profile refresh in one box call (the GPU suite: scratch/run_suite.sh)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
bash scratch/collect_profiles_r5.sh line cfg1 cfg4 train msda > gpurun_out/r5/collect.log 2>&1
tail -5 gpurun_out/r5/collect.log
