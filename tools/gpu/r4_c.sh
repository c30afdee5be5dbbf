#!/bin/bash
# round 4, lease C: per-epilogue attribution (fresh probe builds), fixed tests (fp16 range check, SyncBatchNorm in DDP-wrapper mode),
# x4-upsample LDS-layout variants, small-batch sweep after the layernorm_reduce fix, one default bench run (in-run PMC traffic).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out/r4_c; rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for round in 1 2; do
  for v in full abl1 abl2 abl3 abl4; do
    L=$R/lang-seg_amd/lseg_hip/probe/liblseg_hip_$v.so; [ $v = full ] && L=$R/lang-seg_amd/lseg_hip/liblseg_hip.so
    LSEG_HIP_LIB=$L LSEG_PROBE_VARIANT=$v timeout 200 python tools/epilogue_table.py table >> $O/epi_table.jsonl 2>> $O/epi_err.log
  done
done
for v in full abl2; do
  L=$R/lang-seg_amd/lseg_hip/probe/liblseg_hip_$v.so; [ $v = full ] && L=$R/lang-seg_amd/lseg_hip/liblseg_hip.so
  LSEG_HIP_LIB=$L LSEG_PROBE_VARIANT=$v PROBE_DTYPE=bf16 timeout 200 python tools/epilogue_table.py table >> $O/epi_table_bf16.jsonl 2>> $O/epi_err.log
done
L=$R/lang-seg_amd/lseg_hip/probe/liblseg_hip_abl2.so
LSEG_HIP_LIB=$L LSEG_PROBE_VARIANT=abl2 timeout 300 python tools/epilogue_table.py partial >> $O/epi_partial.jsonl 2>> $O/epi_err.log
for v in 0 1 2 3 0 1; do LSEG_UPS4_VARIANT=$v timeout 120 python tools/upsample_bench.py 36 4 >> $O/ups_bench.log 2>&1; done
( timeout 300 python -m pytest tests/test_gpu_forward.py -q -m gpu -x -k "range_check or one_pass_x4 or split_k_residual" -s 2>&1 | grep -v "^$" | tail -12 ) > $O/tests_fwd.log 2>&1
( timeout 300 python -m pytest tests/test_gpu_train_dp.py -q -m gpu -k "sync_batchnorm_is_installed" 2>&1 | tail -12 ) > $O/tests_ddp.log 2>&1
timeout 200 python tools/step_probe.py --batch 1 2 4 8 --steps 20 --dtype fp16 > $O/step_probe.log 2>&1
( timeout 600 python bench.py 2>&1 | grep "^{" ) > $O/bench_default.json
tail -n 4 $O/tests_fwd.log $O/tests_ddp.log; cat $O/ups_bench.log $O/step_probe.log | grep -v amdgpu; python - <<PY
import json
d=json.load(open("$O/bench_default.json"))
print(d["value"], d["dtype"], d["roofline"]["frac"], d["roofline"]["traffic"], d["batch_sweep_images_per_sec"], d["train_step"], d["cpu_baseline"])
PY
