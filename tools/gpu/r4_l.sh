#!/bin/bash
# round 4, lease L: the GPU tests not re-run since the last full-suite run (commit c2b789c) on the round's last library: two-rank trainer,
# evaluator, metrics, strict-mode forward parity, training against the oracle.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out/r4_l; rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 330 python -m pytest tests/test_gpu_train_dp.py tests/test_gpu_evaluator.py tests/test_metrics.py tests/test_gpu_block_backward.py -q -m gpu 2>&1 | tail -4 ) > $O/tests_a.log 2>&1
( timeout 200 python -m pytest tests/test_gpu_forward.py -q -m gpu -k "strict and (k150 or vitb32 or zs)" 2>&1 | tail -4 ) > $O/tests_b.log 2>&1
tail -n 3 $O/tests_a.log $O/tests_b.log
