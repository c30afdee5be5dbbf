#!/bin/bash
# round 3, GPU call D: split-K at small batches (A/B over the work-item target), training tests after the fp16 head gradient, bench
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( for t in 0 256 512 768 1024; do echo "== LSEG_SPLITK_TARGET=$t"; LSEG_SPLITK_TARGET=$t timeout 300 python tools/step_probe.py --batch 1 2 4 6 --steps 20 --dtype fp16 2>&1 | grep "images/s"; done ) > gpurun_out/r3d_splitk.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_train_dp.py tests/test_gpu_train.py "tests/test_gpu_forward.py::test_split_k_residual_gemms_at_small_batch_equal_the_unsplit_schedule" -q -s -m gpu 2>&1 | tail -40 ) > gpurun_out/r3d_tests.log 2>&1
( timeout 300 python tools/train_bench.py 2>&1 | tail -3 ) > gpurun_out/r3d_trainbench.log 2>&1
cat gpurun_out/r3d_splitk.log; tail -n 25 gpurun_out/r3d_tests.log; cat gpurun_out/r3d_trainbench.log
