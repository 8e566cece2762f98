#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "xattn" 2>&1 | tail -3
python -m pytest tests/test_train_gpu.py -x -q -m gpu -k "bf16" 2>&1 | tail -2
for v in 1 0; do echo -n "CGG_XATTN_BF16_TRAIN=$v "; CGG_XATTN_BF16_TRAIN=$v python bench.py --workload cfg2 --steps 5 --warmup 3 --precision bf16 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 bf16', d['value'], d['ms_per_step'], d['loss'])"; done
