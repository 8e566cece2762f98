#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_train_gpu.py tests/test_dist_gpu.py -x -q -m gpu 2>&1 | tail -2
python -m pytest tests/test_fullsize_gpu.py -x -q -m gpu -k "configs2_forward_train" 2>&1 | tail -2
for p in fp32 bf16; do python bench.py --workload cfg2 --steps 5 --warmup 3 --precision $p 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 $p', d['value'], d['ms_per_step'], d['loss'])"; done
