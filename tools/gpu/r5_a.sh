#!/bin/bash
# lease A (round 5): is the r04 driver failure a regression, a race or a tolerance?  spread of the random-net gradient comparison --
# HEAD (deterministic / atomics), and the bisect libraries c2b789c (last green suite), afef5d5 (asm transpose reads), bfa908d (before the
# GELU refit) on the same inputs -- then the training / DP / metric test files that the r04 driver never reached.
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r5_a; mkdir -p $O
P=lang-seg_amd/lseg_hip/probe
timeout 900 python tests/gpu_train_spread.py --runs 3 --out $O/spread_head_det.json > $O/spread_head_det.txt 2>&1
timeout 300 python tests/gpu_train_spread.py --runs 5 --deterministic 0 --smooth 1 --seeds 3 4 6 --out $O/spread_head_atomics.json > $O/spread_head_atomics.txt 2>&1
for c in c2b789c afef5d5 bfa908d; do
  LSEG_HIP_LIB=$PWD/$P/liblseg_hip_$c.so timeout 300 python tests/gpu_train_spread.py --runs 3 --deterministic 0 --smooth 1 --seeds 3 4 6 --out $O/spread_$c.json > $O/spread_$c.txt 2>&1
done
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_dp.py tests/test_metrics.py tests/test_zz_gpu_random_net_gradients.py -q -m gpu --timeout 900 > $O/tests_train.log 2>&1
tail -5 $O/tests_train.log
