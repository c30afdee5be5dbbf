#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
( for s in 0 1; do echo "== LSEG_SPLITK=$s"; LSEG_SPLITK=$s timeout 300 python tools/step_probe.py --batch 1 2 3 4 6 8 --steps 20 --dtype fp16 2>&1 | grep "images/s"; done
  echo "== text cache"; timeout 300 python tools/step_probe.py --batch 1 4 --steps 20 --dtype fp16 --text-cache 2>&1 | grep "images/s" ) > gpurun_out/r3e_splitk.log 2>&1
cd /tmp && export TMPDIR=/tmp
for b in 1 4; do
  timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_r3e_b$b -o p --output-format csv -- python $R/tools/step_probe.py --batch $b --steps 5 --dtype fp16 > $R/gpurun_out/r3e_prof_b$b.log 2>&1
  python $R/tools/trace_forward.py $R/gpurun_out/prof_r3e_b$b all > $R/gpurun_out/r3e_trace_b$b.txt 2>&1
done
cd $R; rm -rf gpurun_out/prof_r3e_b1 gpurun_out/prof_r3e_b4
cat gpurun_out/r3e_splitk.log; tail -28 gpurun_out/r3e_trace_b1.txt
