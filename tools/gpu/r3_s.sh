#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for w in 4 8; do echo "== LSEG_ATTN_WAVES=$w"; LSEG_ATTN_WAVES=$w timeout 200 python tools/attention_bench.py 36 8 2>&1 | tail -4; LSEG_ATTN_WAVES=$w timeout 300 python tools/step_probe.py --batch 36 --steps 10 --dtype fp16 2>&1 | grep "images/s"; done
