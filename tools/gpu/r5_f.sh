#!/bin/bash
# lease F (round 5): split-K slab workspace 8 192 -> 16 384 rows: the batch sweep with both, the split-K parity test
cd /root/repo; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5_f; mkdir -p $O
for rows in 8192 16384 8192 16384; do
  echo "== LSEG_SPLIT_ROWS=$rows" >> $O/step_probe.txt
  LSEG_SPLIT_ROWS=$rows timeout 200 python tools/step_probe.py --batch 2 4 6 8 12 --steps 30 --dtype fp16 2>&1 | grep -v amdgpu >> $O/step_probe.txt
done
cat $O/step_probe.txt
timeout 600 python -m pytest tests/test_gpu_forward.py -q -m gpu -k "split_k or batch_of_8 or tiny_forward_matches_golden" --timeout 500 > $O/tests.log 2>&1; tail -3 $O/tests.log
