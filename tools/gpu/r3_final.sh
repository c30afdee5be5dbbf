#!/bin/bash
# round 3: the whole GPU suite, the profiles (kernel trace + PMC passes -> traffic.json), the bench line replaying that traffic, and the
# training kernel statistics, all at one commit
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; rm -f gpurun_out/parity_table.txt gpurun_out/train_parity_table.txt
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 1500 python -m pytest tests/ -q -m gpu 2>&1 | tail -15 ) > gpurun_out/r3_final_tests.log 2>&1
( timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > gpurun_out/r3_final_smoke.log 2>&1
DT=fp16
bash tools/collect_profiles.sh $DT $(cat .commit_id 2>/dev/null || echo unknown) > gpurun_out/r3_final_collect.log 2>&1
cp gpurun_out/profiles/traffic.json profiles/r03_traffic.json        # the bench line below replays the traffic measured at this commit
( timeout 600 python bench.py 2>&1 | grep "^{" ) > gpurun_out/r3_bench_line.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_train -o t --output-format csv -- python $R/tools/train_bench.py --steps 3 > $R/gpurun_out/r3_final_train.log 2>&1
cp $(find $R/gpurun_out/prof_train -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r3_train_kernel_stats.csv
python $R/tools/trace_train.py $R/gpurun_out/prof_train 70 > $R/gpurun_out/r3_train_step_table.txt 2>&1; rm -rf $R/gpurun_out/prof_train
cd $R
( timeout 300 python tools/train_bench.py --steps 20 2>&1 | tail -1; timeout 400 python tools/train_loop_bench.py 2>&1 | grep -v "Use norm" | tail -4 ) > gpurun_out/r3_final_trainbench.log 2>&1
tail -5 gpurun_out/r3_final_tests.log; cat gpurun_out/r3_final_smoke.log gpurun_out/r3_final_trainbench.log; cut -c1-600 gpurun_out/r3_bench_line.json
