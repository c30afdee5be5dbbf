#!/bin/bash
# round 4, lease A: new attention bodies (parity + A/B timing), the per-epilogue attribution of the four ViT GEMMs, partial-chip and
# staggered-launch experiments, engine parity with the pre-scaled attention, a bench A/B.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out/r4_a; rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 400 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "attention or vit_block" 2>&1 | tail -8 ) > $O/tests_ops.log 2>&1
for v in 0 1; do LSEG_ATTN_VER=$v timeout 120 python tools/attention_bench.py 36 >> $O/attn_bench.log 2>&1; done
for v in 2 3; do PRESCALED=1 LSEG_ATTN_VER=$v timeout 120 python tools/attention_bench.py 36 >> $O/attn_bench.log 2>&1; done
for v in 0 1; do LSEG_ATTN_VER=$v timeout 120 python tools/attention_bench.py 1 4 >> $O/attn_bench_small.log 2>&1; done
PRESCALED=1 LSEG_ATTN_VER=3 timeout 120 python tools/attention_bench.py 1 4 >> $O/attn_bench_small.log 2>&1
for round in 1 2; do
  for v in full abl1 abl2 abl3 abl4; do
    L=$R/lang-seg_amd/lseg_hip/probe/liblseg_hip_$v.so; [ $v = full ] && L=$R/lang-seg_amd/lseg_hip/liblseg_hip.so
    LSEG_HIP_LIB=$L LSEG_PROBE_VARIANT=$v timeout 200 python tools/epilogue_table.py table >> $O/epi_table.jsonl 2>> $O/epi_err.log
  done
done
for v in full abl2; do
  L=$R/lang-seg_amd/lseg_hip/probe/liblseg_hip_$v.so; [ $v = full ] && L=$R/lang-seg_amd/lseg_hip/liblseg_hip.so
  LSEG_HIP_LIB=$L LSEG_PROBE_VARIANT=$v timeout 300 python tools/epilogue_table.py partial >> $O/epi_partial.jsonl 2>> $O/epi_err.log
done
LSEG_PROBE_VARIANT=full timeout 400 python tools/epilogue_table.py stagger >> $O/epi_stagger.jsonl 2>> $O/epi_err.log
( timeout 600 python -m pytest tests/test_gpu_forward.py -q -m gpu -x -k "tiny_forward_matches_oracle or baseline_configs or batch_of_8" 2>&1 | tail -8 ) > $O/tests_fwd.log 2>&1
( timeout 300 python bench.py --steps 10 --no-sweep --no-cpu-baseline --no-pmc-traffic 2>&1 | grep "^{" ) > $O/bench_new.json
( LSEG_ATTN_NO_PRESCALE=1 LSEG_ATTN_VER=0 timeout 300 python bench.py --steps 10 --no-sweep --no-cpu-baseline --no-pmc-traffic 2>&1 | grep "^{" ) > $O/bench_old.json
( LSEG_ATTN_NO_PRESCALE=1 LSEG_ATTN_VER=1 timeout 300 python bench.py --steps 10 --no-sweep --no-cpu-baseline --no-pmc-traffic 2>&1 | grep "^{" ) > $O/bench_ver1.json
tail -3 $O/tests_ops.log $O/tests_fwd.log; cat $O/attn_bench.log; for f in new old ver1; do python -c "
import json,sys
d=json.load(open('$O/bench_$f.json')); print('$f', d['value'], d['dtype'], d['dtype_selection'].get('bf16'), d['dtype_selection'].get('fp16'), {k:v['avg_launch_ms'] for k,v in d['roofline_kernels'].items()})" 2>&1 | tail -1; done
