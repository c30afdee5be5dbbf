#!/bin/bash
# lease D (round 5): the whole GPU suite (no -x: every failure listed) before the final run
cd /root/repo; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5_d; mkdir -p $O
timeout 1700 python -m pytest tests/ -q -m gpu --timeout 900 -rf > $O/tests_all.log 2>&1; tail -30 $O/tests_all.log
