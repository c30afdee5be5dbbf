#!/bin/bash
# forward at B = 4 and B = 1, kernel by kernel (tools/trace_forward.py), current code
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout 300 python tools/step_probe.py --batch 1 2 4 8 --steps 20 --dtype fp16 2>&1 | grep "images/s"
cd /tmp && export TMPDIR=/tmp
for b in 1 4; do
  timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_r3p_b$b -o p --output-format csv -- python $R/tools/step_probe.py --batch $b --steps 5 --dtype fp16 > /dev/null 2>&1
  python $R/tools/trace_forward.py $R/gpurun_out/prof_r3p_b$b all > $R/gpurun_out/r3p_trace_b$b.txt 2>&1
  rm -rf $R/gpurun_out/prof_r3p_b$b
done
tail -26 $R/gpurun_out/r3p_trace_b4.txt
