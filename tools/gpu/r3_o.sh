#!/bin/bash
# training step after the vectorised attention-backward prep: tests of the attention backward, step time, per-(kernel, grid) table of one step
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_train.py -x -q -m gpu -k "attention or fixture or gradient" > gpurun_out/r3o_tests.log 2>&1; tail -3 gpurun_out/r3o_tests.log
timeout 300 python tools/train_bench.py --steps 20 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_o -o t --output-format csv -- python $R/tools/train_bench.py --steps 3 > /dev/null 2>&1
python $R/tools/trace_train.py $R/gpurun_out/prof_o 70 > $R/gpurun_out/r3o_step_table.txt 2>&1; rm -rf $R/gpurun_out/prof_o
head -75 $R/gpurun_out/r3o_step_table.txt
