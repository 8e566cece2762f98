#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
python -m pytest tests -v -m gpu -x --deselect tests/test_fullsize_gpu.py > gpurun_out/r5/pytest_gpu_full.txt 2>&1
grep -n "PASSED\|FAILED\|ERROR" gpurun_out/r5/pytest_gpu_full.txt | tail -3
grep -n "Fatal\|Segmentation\|Abort\|Current thread" -A12 gpurun_out/r5/pytest_gpu_full.txt | head -40
python scratch/x3_train_gemm_bench.py 2>&1 | grep -v amdgpu | tee gpurun_out/r5/x3_train_gemm_bench.txt
python scratch/linear_shapes_bench.py 2>&1 | grep -v amdgpu | tee gpurun_out/r5/linear_shapes_bench.txt
