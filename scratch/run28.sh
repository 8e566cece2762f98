cd $GRAFT_REPO_ROOT
python -m pytest tests/test_x3s_gpu.py -q -x -k "per_token_table" 2>&1 | tail -3
python -m pytest tests/test_fullsize_gpu.py tests/test_head_gpu.py tests/test_dist_gpu.py -q -x -k "train or decoder or dist" 2>&1 | tail -3
for i in 1 2; do python bench.py --workload cfg2 --precision fp32 --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('cfg2 fp32', d['value'], d['ms_per_step'])"; done
