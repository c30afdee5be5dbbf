#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for cfg in "2 2" "3 2" "2 4"; do set -- $cfg
  echo "== LSEG_ATTN_STAGES=$1 LSEG_ATTN_WAVES=$2"
  LSEG_ATTN_STAGES=$1 LSEG_ATTN_WAVES=$2 timeout 200 python tools/attention_bench.py 36 2>&1 | tail -2
done
