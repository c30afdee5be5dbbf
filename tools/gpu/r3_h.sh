#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
( for e in "LSEG_ATTN_TREE=0" "LSEG_ATTN_TREE=1" "LSEG_ATTN_TREE=0" "LSEG_ATTN_TREE=1"; do echo "== $e"; env $e timeout 200 python tools/attention_bench.py 36 2>&1 | grep "TF/s"; done
  for e in "LSEG_ATTN_TREE=0" "LSEG_ATTN_TREE=1"; do echo "== $e"; env $e timeout 300 python tools/step_probe.py --batch 36 --steps 10 --dtype fp16 2>&1 | grep "images/s"; done ) > gpurun_out/r3h_attn2.log 2>&1
cat gpurun_out/r3h_attn2.log
