#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_x3s_gpu.py -x -q -m gpu -k "wgrad or linear or ffn or conv3x3 or fpn or encoder_block" 2>&1 | tail -2
python bench.py --workload cfg2 --steps 5 --warmup 3 --precision fp32 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 fp32', d['value'], d['ms_per_step'])"
