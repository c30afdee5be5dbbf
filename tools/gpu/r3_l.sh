#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_fill -o t --output-format csv -- python $R/tools/train_bench.py --steps 3 > $R/gpurun_out/r3l.log 2>&1
python $R/tools/trace_fills.py $R/gpurun_out/prof_fill > $R/gpurun_out/r3l_fills.txt 2>&1; rm -rf $R/gpurun_out/prof_fill
cat $R/gpurun_out/r3l_fills.txt
