#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
( timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_block_backward.py -q -m gpu -k "attention or block" 2>&1 | tail -2
  timeout 300 python tools/train_bench.py --steps 4 2>&1 | tail -1
  timeout 600 python -m pytest tests/test_gpu_train.py -q -m gpu -k "fixtures" 2>&1 | tail -2 ) > gpurun_out/r3k_bwd_xcd.log 2>&1
cat gpurun_out/r3k_bwd_xcd.log
