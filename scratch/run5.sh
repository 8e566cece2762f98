mkdir -p gpurun_out/r5
python scratch/train_ops_prof_r5.py cfg2 fp32 2>&1 | grep -v amdgpu > gpurun_out/r5/ops_cfg2_fp32.txt
python scratch/train_ops_prof_r5.py cfg2 bf16 2>&1 | grep -v amdgpu > gpurun_out/r5/ops_cfg2_bf16.txt
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "astat" 2>&1 | tail -3
python scratch/einsum_sweep.py 2>&1 | grep -v amdgpu | grep astat | cut -c1-330
