#!/bin/bash
# round 4, lease K: refresh of the committed evidence at the round's last library (GELU with the folded scale, pre-scaled attention body for
# fp16 only, K-major fix): the GEMM / forward-parity / GELU-fusion tests, the profiles (kernel trace + PMC passes -> traffic.json), the bench
# line (in-run PMC traffic), the training step.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out/r4_k; rm -rf $O; mkdir -p $O; rm -f gpurun_out/parity_table.txt
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 400 python -m pytest tests/test_gpu_ops.py -q -m gpu -x 2>&1 | tail -4 ) > $O/tests_ops.log 2>&1
( timeout 400 python -m pytest tests/test_gpu_forward.py tests/test_gpu_train.py -q -m gpu -x -k "(baseline_configs and not strict) or (masks_match and not strict) or tiny_forward_matches_oracle or gelu_epilogue or vitl16_480_k150" 2>&1 | tail -4 ) > $O/tests_fwd.log 2>&1
cp gpurun_out/parity_table.txt $O/parity_table.txt 2>/dev/null
bash tools/collect_profiles.sh fp16 $(cat .commit_id 2>/dev/null || echo unknown) > $O/collect.log 2>&1
( timeout 900 python bench.py 2>&1 | grep "^{" ) > $O/bench_line.json
( timeout 200 python tools/train_bench.py --steps 20 2>&1 | tail -1 ) > $O/trainbench.log 2>&1
tail -n 2 $O/tests_ops.log $O/tests_fwd.log; cat $O/trainbench.log; cat $O/parity_table.txt; cut -c1-400 $O/bench_line.json
