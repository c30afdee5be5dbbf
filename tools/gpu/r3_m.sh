#!/bin/bash
# 256 x 256 K-major wgrad tiles: op tests, train parity, A/B of the training step
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_train.py -x -q -m gpu > gpurun_out/r3m_tests.log 2>&1; tail -5 gpurun_out/r3m_tests.log
for t in 2 0; do
  echo "== LSEG_WGRAD_TILE=$t"; LSEG_WGRAD_TILE=$t timeout 300 python tools/train_bench.py --steps 20 2>&1 | tail -2
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_m -o t --output-format csv -- python $R/tools/train_bench.py --steps 5 > /dev/null 2>&1
python $R/tools/summarize_profiles.py $R/gpurun_out/prof_m 2>/dev/null | head -40 > $R/gpurun_out/r3m_stats.txt; rm -rf $R/gpurun_out/prof_m
head -30 $R/gpurun_out/r3m_stats.txt
