#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_block_backward.py -q -x -m gpu -k "backward" 2>&1 | tail -12
  for e in "LSEG_WGRAD_TRANSPOSE=1" "A=1"; do echo "== $e"; env $e timeout 300 python tools/train_bench.py --steps 4 2>&1 | tail -1; done
  timeout 900 python -m pytest tests/test_gpu_train.py -q -x -m gpu 2>&1 | tail -8 ) > gpurun_out/r3i_kmaj.log 2>&1
cat gpurun_out/r3i_kmaj.log
