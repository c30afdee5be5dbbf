#!/bin/bash
# round 3, GPU call G: bench line + profiles at the chosen operand type
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
( timeout 600 python bench.py 2>&1 | grep "^{" ) > gpurun_out/r3g_bench_line.json
DT=$(python -c "import json;print(json.load(open('gpurun_out/r3g_bench_line.json'))['dtype'])" 2>/dev/null || echo fp16)
bash tools/collect_profiles.sh $DT $(cat .commit_id 2>/dev/null || echo unknown) > gpurun_out/r3g_collect.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_train -o t --output-format csv -- python $R/tools/train_bench.py --steps 3 > $R/gpurun_out/r3g_train.log 2>&1
cp $(find $R/gpurun_out/prof_train -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r3g_train_kernel_stats.csv; rm -rf $R/gpurun_out/prof_train
cd $R; python -c "
import json;d=json.load(open('gpurun_out/r3g_bench_line.json'))
for k in ('value','dtype','dtype_selection','roofline','batch_sweep_images_per_sec','config5_k1000','train_step'): print(k, json.dumps(d.get(k))[:700])"
tail -40 gpurun_out/profiles/summary.txt; tail -3 gpurun_out/r3g_train.log
