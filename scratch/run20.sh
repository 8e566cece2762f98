#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_x3s_gpu.py tests/test_x3_gpu.py -x -q -m gpu 2>&1 | tail -2
for w in cfg2 cfg3; do python bench.py --workload $w --steps 5 --warmup 3 --precision fp32 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w fp32', d['value'], d['ms_per_step'], d['loss'])"; done
