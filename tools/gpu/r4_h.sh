#!/bin/bash
# round 4, lease H: the K-major weight-gradient GEMM with its transpose reads as inline asm (no vmcnt(0) in the K-step): backward op tests,
# brick tests, the training step against the oracle and the reference-autograd fixtures (incl. configs[3]'s own shape), step time.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out/r4_h; rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_block_backward.py -q -m gpu -x -k "backward" 2>&1 | tail -5 ) > $O/tests_ops.log 2>&1
( timeout 700 python -m pytest tests/test_gpu_train.py -q -m gpu -x 2>&1 | tail -6 ) > $O/tests_train.log 2>&1
( timeout 200 python tools/train_bench.py --steps 20 2>&1 | tail -1 ) > $O/trainbench.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_train -o t --output-format csv -- python $R/tools/train_bench.py --steps 3 > $O/train_rocprof.log 2>&1
python $R/tools/trace_train.py $O/prof_train 30 > $O/train_step_table.txt 2>&1; rm -rf $O/prof_train
cd $R
tail -n 3 $O/tests_ops.log $O/tests_train.log; cat $O/trainbench.log; head -14 $O/train_step_table.txt | cut -c1-150
