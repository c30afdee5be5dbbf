#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
( for s in "LSEG_SPLITK=0 LSEG_ATTN_WAVES=4" "LSEG_SPLITK=1 LSEG_ATTN_WAVES=4" "LSEG_SPLITK=1"; do echo "== $s"; env $s timeout 300 python tools/step_probe.py --batch 1 2 4 8 --steps 20 --dtype fp16 2>&1 | grep "images/s"; done ) > gpurun_out/r3f_sweep.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_ops.py -q -x -m gpu 2>&1 | tail -15 ) > gpurun_out/r3f_tests.log 2>&1
cat gpurun_out/r3f_sweep.log; tail -n 12 gpurun_out/r3f_tests.log
