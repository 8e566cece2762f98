cd $GRAFT_REPO_ROOT
python -m pytest tests/test_train_gpu.py -q -x -s 2>&1 | grep -v Warning | tail -6
for i in 1 2; do python bench.py --workload cfg2 --precision fp32 --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('cfg2 fp32', d['value'], d['ms_per_step'])"; done
python bench.py --workload cfg2 --precision bf16 --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('cfg2 bf16', d['value'], d['ms_per_step'])"
