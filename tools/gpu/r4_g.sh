#!/bin/bash
# round 4, lease G: the x4 logits upsample's phase-3 forms and store flavours standalone (bit equality at B = 4 for each), the upsample
# op test, and the bench line's kernel table with 32-register residual chunks as the library default.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out/r4_g; rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python tools/upsample_bench.py 36 4 -- 0 1 10 11 110 20 120 0 10 2>&1 | grep -v amdgpu > $O/ups.log
( timeout 300 python -m pytest tests/test_gpu_forward.py tests/test_gpu_ops.py -q -m gpu -x -k "x4_upsample or upsample or vit_block_gemm or gemm" 2>&1 | tail -5 ) > $O/tests.log 2>&1
( LSEG_UPS4_VARIANT=10 timeout 300 python -m pytest tests/test_gpu_forward.py -q -m gpu -x -k "x4_upsample" 2>&1 | tail -3 ) > $O/tests_rolling.log 2>&1
for v in 0 10; do
  ( LSEG_UPS4_VARIANT=$v timeout 200 python bench.py --steps 20 --dtype fp16 --no-parity --no-sweep --no-cpu-baseline --no-pmc-traffic 2>&1 | grep "^{" | sed "s/^{/{\"variant\": \"ups$v\", /" ) >> $O/bench_ab.jsonl
done
cat $O/ups.log; tail -n 3 $O/tests.log $O/tests_rolling.log; python - <<PY
import json
for l in open("$O/bench_ab.jsonl"):
    d=json.loads(l)
    print(d["variant"], d["value"], {k: round(v["avg_launch_ms"]*1e3,1) for k,v in d["roofline_kernels"].items()}, d.get("engine_forward_ms_hip_events"))
PY
