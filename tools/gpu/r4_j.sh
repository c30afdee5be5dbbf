#!/bin/bash
# round 4, lease J: is the pre-scaled attention body what moved the bf16 mask flips at configs[1] (1.62 % in round 3 -> 2.64 %)?  The
# reference-fixture parity tests with LSEG_ATTN_PRE=0 (scale-in-softmax body everywhere) against the default.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out/r4_j; rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for pre in default 0; do
  rm -f gpurun_out/parity_table.txt
  if [ $pre = default ]; then unset LSEG_ATTN_PRE; else export LSEG_ATTN_PRE=$pre; fi
  ( timeout 300 python -m pytest tests/test_gpu_forward.py -q -m gpu -k "baseline_configs and not strict" 2>&1 | tail -3 ) > $O/tests_pre_$pre.log 2>&1
  cp gpurun_out/parity_table.txt $O/parity_pre_$pre.txt
done
unset LSEG_ATTN_PRE
for f in $O/parity_pre_*.txt; do echo == $f; cat $f; done; tail -n 2 $O/tests_pre_*.log
