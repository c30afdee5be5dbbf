#!/bin/bash
# round 3, GPU call A: new training tests, bench with dtype selection, small-batch kernel breakdown
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 900 python -m pytest tests/test_gpu_train_dp.py -x -q -s -m gpu 2>&1 | tail -60 ) > gpurun_out/r3a_dp.log 2>&1
( timeout 600 python -m pytest tests/test_gpu_train.py -x -q -s -m gpu -k "fixtures or lsegnet" 2>&1 | tail -40 ) > gpurun_out/r3a_train.log 2>&1
( timeout 600 python bench.py 2>&1 | tail -5 ) > gpurun_out/r3a_bench.log 2>&1
( timeout 400 python tools/train_loop_bench.py --slow 2>&1 | tail -8 ) > gpurun_out/r3a_loop.log 2>&1
cd /tmp && export TMPDIR=/tmp
for b in 1 4; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r3_b$b -o p --output-format csv -- python $R/tools/step_probe.py --batch $b --steps 10 --dtype fp16 > $R/gpurun_out/r3a_prof_b$b.log 2>&1
  python $R/tools/kstats.py $R/gpurun_out/prof_r3_b$b 13 > $R/gpurun_out/r3a_kstats_b$b.txt 2>&1
done
cd $R; tail -3 gpurun_out/r3a_dp.log gpurun_out/r3a_train.log gpurun_out/r3a_bench.log gpurun_out/r3a_loop.log; head -30 gpurun_out/r3a_kstats_b1.txt
