#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_x3s_gpu.py tests/test_x3_gpu.py -x -q -m gpu 2>&1 | tail -6
python -m pytest tests/test_fullsize_gpu.py -x -q -m gpu -k "test_configs3_detector_train_step_runs or configs2_forward_train" 2>&1 | tail -4
python scratch/x3_train_gemm_bench.py 344064 10 2>&1 | grep -v "amdgpu\|encode\|x3s"
