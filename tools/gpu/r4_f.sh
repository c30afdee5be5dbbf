#!/bin/bash
# round 4, lease F: (1) op tests + forward parity on the QKV-epilogue fix, the widened attention stores and the separable x4 upsample;
# (2) in-engine per-kernel times (bench.py roofline_kernels, fp16, B = 36) of the library against the A/B builds: oldau = previous
# attention stores + previous x4 upsample, r32 = 32-register residual chunks in EPI_RES32 (no spills), noslp = gemm.hip without the SLP
# vectoriser, prio = s_setprio 1 for waves 4-7; (3) the x4 upsample standalone, new against old.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out/r4_f; rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
LIBD=$R/lang-seg_amd/lseg_hip
( timeout 400 python -m pytest tests/test_gpu_ops.py -q -m gpu -x 2>&1 | tail -8 ) > $O/tests_ops.log 2>&1
( timeout 500 python -m pytest tests/test_gpu_forward.py -q -m gpu -x -k "(baseline_configs and k150 and fp16) or x4_upsample or tiny_forward_matches_oracle or masks_and_metrics or batch_of_8" 2>&1 | tail -10 ) > $O/tests_fwd.log 2>&1
for v in main oldau r32 noslp prio main; do
  case $v in main) L=$LIBD/liblseg_hip.so;; oldau) L=$LIBD/probe_old/liblseg_hip_oldau.so;; *) L=$LIBD/probe/liblseg_hip_var_$v.so;; esac
  ( LSEG_HIP_LIB=$L timeout 200 python bench.py --steps 20 --dtype fp16 --no-parity --no-sweep --no-cpu-baseline --no-pmc-traffic 2>&1 | grep "^{" | sed "s/^{/{\"variant\": \"$v\", /" ) >> $O/bench_ab.jsonl
done
for v in main oldau; do
  case $v in main) L=$LIBD/liblseg_hip.so;; oldau) L=$LIBD/probe_old/liblseg_hip_oldau.so;; esac
  echo "== $v" >> $O/ups.log; LSEG_HIP_LIB=$L timeout 100 python tools/upsample_bench.py 36 4 2>&1 | grep -v amdgpu >> $O/ups.log
done
tail -n 3 $O/tests_ops.log $O/tests_fwd.log; cat $O/ups.log; python - <<PY
import json
for l in open("$O/bench_ab.jsonl"):
    d=json.loads(l)
    print(d["variant"], d["value"], {k: round(v["avg_launch_ms"]*1e3,1) for k,v in d["roofline_kernels"].items()}, d.get("engine_forward_ms_hip_events"))
PY
