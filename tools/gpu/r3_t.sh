#!/bin/bash
# attention forward: K / V^T stages (2 | 3) x waves per workgroup (4 | 8), standalone op and the B = 36 forward
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "attention" 2>&1 | tail -2
LSEG_ATTN_STAGES=3 timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "attention" 2>&1 | tail -2
for cfg in "2 4" "3 4" "3 8" "2 8"; do set -- $cfg
  echo "== LSEG_ATTN_STAGES=$1 LSEG_ATTN_WAVES=$2"
  LSEG_ATTN_STAGES=$1 LSEG_ATTN_WAVES=$2 timeout 200 python tools/attention_bench.py 36 2>&1 | tail -2
  LSEG_ATTN_STAGES=$1 LSEG_ATTN_WAVES=$2 timeout 300 python tools/step_probe.py --batch 36 --steps 10 --dtype fp16 2>&1 | grep "images/s"
done
