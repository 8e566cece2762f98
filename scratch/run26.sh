cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py -q -x -k "match_cost" 2>&1 | tail -3
python -m pytest tests/test_train_gpu.py -q -x 2>&1 | tail -2
for i in 1 2; do python bench.py --workload cfg2 --precision fp32 --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('cfg2 fp32', d['value'], d['ms_per_step'])"; done
python bench.py --workload cfg2 --precision bf16 --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('cfg2 bf16', d['value'], d['ms_per_step'])"
