#!/bin/bash
# train-mode RCUs on materialised ReLU maps: train suites + the step
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 1200 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_dp.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python tools/train_bench.py --steps 20 2>&1 | tail -1
timeout 300 python tools/train_bench.py --steps 20 2>&1 | tail -1
