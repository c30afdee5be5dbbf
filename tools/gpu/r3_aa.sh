#!/bin/bash
# fp32 linear epilogue (EPI_LIN32): op / forward / train suites, then the training step
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_forward.py tests/test_gpu_train.py -x -q -m gpu 2>&1 | tail -3
LSEG_GEMM_GENERIC_EPI=1 timeout 300 python tools/train_bench.py --steps 20 2>&1 | tail -1
timeout 300 python tools/train_bench.py --steps 20 2>&1 | tail -1
