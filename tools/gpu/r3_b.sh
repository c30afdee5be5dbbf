#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
N=ref_train_full_vitl16_480x480_k150_b2
( timeout 200 python tools/train_fixture_report.py $N 2>&1 | tail -150 ) > gpurun_out/r3b_rep_default.log
cat gpurun_out/r3b_rep_default.log
