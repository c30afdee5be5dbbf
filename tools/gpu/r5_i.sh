#!/bin/bash
# lease I (round 5): the training step under rocprofv3 with the PMC passes (tools/collect_train_profiles.sh) at the final library
cd /root/repo; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 560 bash tools/collect_train_profiles.sh > gpurun_out/r5_i.log 2>&1
cp gpurun_out/train_profiles/summary.txt gpurun_out/r5_i_train_pmc_summary.txt 2>/dev/null
head -40 gpurun_out/train_profiles/summary.txt
