#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_train_gpu.py tests/test_x3s_gpu.py -x -q -m gpu 2>&1 | tail -2
