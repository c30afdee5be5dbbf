#!/bin/bash
# round 5 final: the WHOLE GPU suite at the library commit in lang-seg_amd/lseg_hip/_build_id.txt (log kept as profiles/r05_gpu_suite_HEAD.txt),
# smoke, the profiles (kernel trace + PMC passes -> traffic.json), the bench line (in-run PMC traffic), the training kernel statistics and
# step table -- all at one commit.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out/r5_final; rm -rf $O; mkdir -p $O; rm -f gpurun_out/parity_table.txt gpurun_out/train_parity_table.txt
export HSA_ENABLE_IPC_MODE_LEGACY=0
COMMIT=$(cat lang-seg_amd/lseg_hip/_build_id.txt 2>/dev/null || echo unknown)
( echo "library built at commit $COMMIT; python -m pytest tests/ -q -m gpu -x"; timeout 1700 python -m pytest tests/ -q -m gpu -x 2>&1 | tail -25 ) > $O/tests.log 2>&1
( timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > $O/smoke.log 2>&1
bash tools/collect_profiles.sh fp16 $COMMIT > $O/collect.log 2>&1
cp gpurun_out/profiles/traffic.json $O/traffic.json; cp gpurun_out/profiles/summary.txt $O/rocprofv3_summary.txt; cp gpurun_out/profiles/kernel_stats.csv $O/kernel_stats.csv
( timeout 900 python bench.py 2>&1 | grep "^{" ) > $O/bench_line.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_train -o t --output-format csv -- python $R/tools/train_bench.py --steps 3 > $O/train_rocprof.log 2>&1
cp $(find $O/prof_train -name "*kernel_stats.csv" | head -1) $O/train_kernel_stats.csv
python $R/tools/trace_train.py $O/prof_train 70 > $O/train_step_table.txt 2>&1; rm -rf $O/prof_train
cd $R
( timeout 300 python tools/train_bench.py --steps 20 2>&1 | tail -1; timeout 400 python tools/train_loop_bench.py 2>&1 | grep -v "Use norm" | tail -4 ) > $O/trainbench.log 2>&1
cp gpurun_out/parity_table.txt gpurun_out/train_parity_table.txt $O/ 2>/dev/null
tail -6 $O/tests.log; cat $O/smoke.log $O/trainbench.log; cut -c1-600 $O/bench_line.json
