#!/bin/bash
# round 4, lease I: the relaxed first barrier behind a full tile's epilogue (vmcnt A_SPW + EPI_OPS): op tests, forward parity at the BASELINE
# configs, batch-invariance / tile-config guards, and the in-engine A/B against the previous library (same box, interleaved).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out/r4_i; rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
LIBD=$R/lang-seg_amd/lseg_hip
( timeout 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_tile_configs.py -q -m gpu -x 2>&1 | tail -5 ) > $O/tests_ops.log 2>&1
( timeout 500 python -m pytest tests/test_gpu_forward.py -q -m gpu -x -k "(baseline_configs and not strict) or masks_match and not strict or batch_of_8 or batch_entries or tiny_forward_matches_oracle" 2>&1 | tail -5 ) > $O/tests_fwd.log 2>&1
for v in new prev new prev; do
  case $v in new) L=$LIBD/liblseg_hip.so;; prev) L=$LIBD/probe_old/liblseg_hip_prev.so;; esac
  ( LSEG_HIP_LIB=$L timeout 200 python bench.py --steps 20 --dtype fp16 --no-parity --no-sweep --no-cpu-baseline --no-pmc-traffic 2>&1 | grep "^{" | sed "s/^{/{\"variant\": \"$v\", /" ) >> $O/bench_ab.jsonl
done
tail -n 3 $O/tests_ops.log $O/tests_fwd.log; python - <<PY
import json
for l in open("$O/bench_ab.jsonl"):
    d=json.loads(l)
    print(d["variant"], d["value"], {k: round(v["avg_launch_ms"]*1e3,1) for k,v in d["roofline_kernels"].items()}, d.get("engine_forward_ms_hip_events"))
PY
