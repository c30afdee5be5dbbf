#!/bin/bash
# round 4, lease E: (1) the op tests + the forward parity tests at the BASELINE configs on the round's rewritten epilogues / attention body,
# (2) K-loop rate and whole-kernel time of the ViT GEMMs under the probe tiles (LSEG_GEMM_TILE 6 / 7 / 9 / 8) in the full, K-loop-only (abl2)
# and no-residual-load (abl3) builds, (3) the pure-write ceiling of the chip, (4) the bench line at HEAD (no CPU leg, PMC traffic in-run).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out/r4_e; rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 500 python -m pytest tests/test_gpu_ops.py -q -m gpu -x 2>&1 | tail -8 ) > $O/tests_ops.log 2>&1
( timeout 600 python -m pytest tests/test_gpu_forward.py -q -m gpu -x -k "(baseline_configs or masks_match or tiny_forward_matches_oracle or range_check or x4_upsample) and not strict" 2>&1 | tail -10 ) > $O/tests_fwd.log 2>&1
for v in full abl2 abl3; do
  L=$R/lang-seg_amd/lseg_hip/probe/liblseg_hip_$v.so; [ $v = full ] && L=$R/lang-seg_amd/lseg_hip/liblseg_hip.so
  LSEG_HIP_LIB=$L LSEG_PROBE_VARIANT=$v timeout 150 python tools/epilogue_table.py tiles >> $O/epi_tiles.jsonl 2>> $O/epi_err.log
done
timeout 100 python tools/fill_bench.py > $O/fill.log 2>&1
( timeout 400 python bench.py --steps 20 --no-cpu-baseline 2>&1 | grep "^{" ) > $O/bench.json
tail -n 3 $O/tests_ops.log $O/tests_fwd.log; grep -v amdgpu $O/fill.log; python - <<PY
import json, collections
rows=[json.loads(l) for l in open("$O/epi_tiles.jsonl")]
for k in ["qkv","fc1","proj","fc2"]:
    for t in (6,7,9,8):
        print(k, t, {r["variant"]: r["us"] for r in rows if r["kind"]==k and r["tile"]==t})
d=json.load(open("$O/bench.json"))
print(d["value"], d["dtype"], d.get("dtype_selection"), d["roofline"]["frac"], {k:v["avg_launch_ms"] for k,v in d["roofline_kernels"].items()}, d.get("batch_sweep_images_per_sec"), d.get("train_step"))
PY
