python scratch/msda_dbg.py 2>&1 | grep -v amdgpu
python -m pytest tests/test_x3s_gpu.py -x -q -m gpu -k "ffn_node" 2>&1 | tail -5
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "astat or sorted_scatter" 2>&1 | tail -5
python scratch/einsum_sweep.py 2>&1 | grep -v amdgpu | grep astat | cut -c1-330
bash scratch/msda_bwd_r5.sh 2>&1 | tail -14
