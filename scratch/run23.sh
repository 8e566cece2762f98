cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py -q -x -k "xattn" 2>&1 | tail -5
python -m pytest tests/test_x3s_gpu.py tests/test_fullsize_gpu.py -q -x -k "train or decoder or xattn" 2>&1 | tail -3
for i in 1 2; do python bench.py --workload cfg2 --precision fp32 --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('cfg2 fp32', d['value'], d['ms_per_step'])"; done
CGG_XATTN_X3_TRAIN=0 python bench.py --workload cfg2 --precision fp32 --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('cfg2 fp32 (f32 mfma fwd)', d['value'], d['ms_per_step'])"
