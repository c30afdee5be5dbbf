#!/bin/bash
# lease G (round 5): correlation kernel with two waves per tile (LSEG_CORR_ONE_WAVE=1 = the one-wave form of lease C)
cd /root/repo; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5_g; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "corr" --timeout 300 > $O/tests_corr_op.log 2>&1; tail -3 $O/tests_corr_op.log
( echo "== two waves per tile"; timeout 120 python tools/corr_bench.py; echo "== LSEG_CORR_ONE_WAVE=1"; LSEG_CORR_ONE_WAVE=1 timeout 120 python tools/corr_bench.py; echo "== two waves per tile (again)"; timeout 120 python tools/corr_bench.py --batch 36 ) 2>&1 | grep -v amdgpu > $O/corr_bench.txt; cat $O/corr_bench.txt
timeout 900 python -m pytest tests/test_gpu_forward.py -q -m gpu -k "dedicated or (baseline_configs and k150 and fp16)" --timeout 600 -s > $O/tests_fwd.log 2>&1; tail -3 $O/tests_fwd.log; grep "fused vs generic" $O/tests_fwd.log
timeout 300 python bench.py --no-cpu-baseline --no-pmc-traffic --no-sweep --no-parity --dtype fp16 --steps 10 > $O/bench_two.json 2> $O/bench_two.err
LSEG_CORR_ONE_WAVE=1 timeout 300 python bench.py --no-cpu-baseline --no-pmc-traffic --no-sweep --no-parity --dtype fp16 --steps 10 > $O/bench_one.json 2> $O/bench_one.err
python - <<'PY'
import json
for f in ("bench_two", "bench_one"):
    d = json.loads([l for l in open(f"/root/repo/gpurun_out/r5_g/{f}.json") if l.startswith("{")][-1])
    print(f, d["value"], d["ms_per_step"], d["roofline_kernels"].get("correlation"))
PY
