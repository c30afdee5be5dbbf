#!/bin/bash
# (needs tools/ab/liblseg_hip_prev.so [and _noslp.so]: libraries of the compared commits built with `make -C lang-seg_amd/csrc` from `git archive`)
# attention kernels: previous build / scalar fp32 forms + saddr loads + scale folded out of the backward loops / the same without SLP-vectorised packed ops
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_train.py -x -q -m gpu -k "attention or fixture or gradient" > gpurun_out/r3x_tests.log 2>&1; tail -3 gpurun_out/r3x_tests.log
for lib in $R/tools/ab/liblseg_hip_prev.so $R/lang-seg_amd/lseg_hip/liblseg_hip.so $R/tools/ab/liblseg_hip_noslp.so; do
  echo "== $(basename $lib)"
  LSEG_HIP_LIB=$lib timeout 200 python tools/attention_bench.py 36 2>&1 | tail -2
  LSEG_HIP_LIB=$lib timeout 300 python tools/train_bench.py --steps 20 2>&1 | tail -1
  LSEG_HIP_LIB=$lib timeout 300 python tools/step_probe.py --batch 36 --steps 10 --dtype fp16 2>&1 | grep "images/s"
done
