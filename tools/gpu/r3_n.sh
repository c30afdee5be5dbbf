#!/bin/bash
# LayerNorm-backward column reduce on 128 workgroups + GELU fused into the fc1 / fc2-dX epilogues: tests, then the training step A/B
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_train.py -x -q -m gpu > gpurun_out/r3n_tests.log 2>&1; tail -5 gpurun_out/r3n_tests.log
for e in LSEG_NO_GELU_FUSE=1 LSEG_X=1; do
  echo "== $e"; env $e timeout 300 python tools/train_bench.py --steps 20 2>&1 | tail -1
done
