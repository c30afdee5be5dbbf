#!/bin/bash
# round 4, lease M (last of the budget): does a stream priority for the text tower's side stream change the forward?  step_probe at B = 36 and
# B = 1, default / low / high.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out/r4_m; rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for v in default low high; do
  if [ $v = default ]; then unset LSEG_TEXT_STREAM_PRIORITY; else export LSEG_TEXT_STREAM_PRIORITY=$v; fi
  echo "== $v" >> $O/probe.log
  timeout 60 python tools/step_probe.py --batch 36 1 --steps 30 --dtype fp16 2>&1 | grep -v amdgpu >> $O/probe.log
done
cat $O/probe.log
