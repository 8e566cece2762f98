#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_fullsize_gpu.py -x -q -m gpu -k "configs3" 2>&1 | tail -2
for p in fp32 bf16; do python bench.py --workload cfg3 --steps 5 --warmup 3 --precision $p 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg3 $p', d['value'], d['ms_per_step'], d['loss'])"; done
