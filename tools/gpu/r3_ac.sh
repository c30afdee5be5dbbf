#!/bin/bash
# bias gradients from the LayerNorm backward: train suites, step time, then the per-step kernel table and the Lightning-shaped loop
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 1200 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_dp.py tests/test_gpu_ops.py -x -q -m gpu 2>&1 | tail -3
( timeout 300 python tools/train_bench.py --steps 20 2>&1 | tail -1; timeout 400 python tools/train_loop_bench.py 2>&1 | grep -v "Use norm" | tail -3 ) | tee gpurun_out/r3ac_trainbench.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_ac -o t --output-format csv -- python $R/tools/train_bench.py --steps 3 > /dev/null 2>&1
cp $(find $R/gpurun_out/prof_ac -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r3ac_train_kernel_stats.csv
python $R/tools/trace_train.py $R/gpurun_out/prof_ac 70 > $R/gpurun_out/r3ac_train_step_table.txt 2>&1; rm -rf $R/gpurun_out/prof_ac
head -12 $R/gpurun_out/r3ac_train_step_table.txt
