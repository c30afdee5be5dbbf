#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
( timeout 300 python tools/batch_invariance_probe.py tiny16 64 64 4 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r3j.log 2>&1
cat gpurun_out/r3j.log
