#!/bin/bash
# round 4, lease D: correctness of the rewritten epilogues / attention / upsample (op tests + forward parity), timing of the four ViT GEMMs
# and the bench line with them, momentum hand-over test.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out/r4_d; rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -x 2>&1 | tail -8 ) > $O/tests_ops.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_forward.py -q -m gpu -x -k "not strict" 2>&1 | tail -10 ) > $O/tests_fwd.log 2>&1
( timeout 600 python -m pytest tests/test_gpu_train_dp.py tests/test_gpu_tile_configs.py -q -m gpu -x 2>&1 | tail -10 ) > $O/tests_dp_tiles.log 2>&1
for round in 1 2; do
  for v in full abl1 abl2; do
    L=$R/lang-seg_amd/lseg_hip/probe/liblseg_hip_$v.so; [ $v = full ] && L=$R/lang-seg_amd/lseg_hip/liblseg_hip.so
    LSEG_HIP_LIB=$L LSEG_PROBE_VARIANT=$v timeout 200 python tools/epilogue_table.py table >> $O/epi_table.jsonl 2>> $O/epi_err.log
  done
done
timeout 120 python tools/upsample_bench.py 36 4 > $O/ups_bench.log 2>&1
PRESCALED=1 timeout 120 python tools/attention_bench.py 36 >> $O/attn_bench.log 2>&1
timeout 120 python tools/attention_bench.py 36 >> $O/attn_bench.log 2>&1
( timeout 400 python bench.py --steps 20 --no-cpu-baseline 2>&1 | grep "^{" ) > $O/bench.json
tail -n 3 $O/tests_ops.log $O/tests_fwd.log $O/tests_dp_tiles.log; grep -v amdgpu $O/ups_bench.log $O/attn_bench.log; python - <<PY
import json, collections
rows=[json.loads(l) for l in open("$O/epi_table.jsonl")]
agg=collections.defaultdict(list)
for r in rows: agg[(r['kind'],r['variant'])].append(r['us'])
for k in ["qkv","fc1","proj","fc2"]: print(k, {v:agg[(k,v)] for v in ["full","abl1","abl2"]})
d=json.load(open("$O/bench.json"))
print(d["value"], d["dtype"], d["dtype_selection"], d["roofline"]["frac"], {k:v["avg_launch_ms"] for k,v in d["roofline_kernels"].items()}, d["batch_sweep_images_per_sec"], d["train_step"])
PY
