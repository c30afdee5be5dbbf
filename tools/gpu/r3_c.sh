#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
( python tools/probes/f16_denorm_probe.py 2>&1 | grep -v amdgpu.ids
  for a in "clip_vitl16_384 480 480 2 150 3" "clip_vitl16_384 480 480 1 150 3" "clip_vitl16_384 96 96 2 5 3"; do
  timeout 400 python tools/train_oracle_probe.py $a 2>&1 | grep -v amdgpu.ids
done
timeout 600 python -m pytest tests/test_gpu_train.py -x -q -s -m gpu 2>&1 | tail -25 ) > gpurun_out/r3c_probe4.log 2>&1
cat gpurun_out/r3c_probe4.log
