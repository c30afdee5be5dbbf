#!/bin/bash
# round 4: the whole GPU suite, smoke, the profiles (kernel trace + PMC passes -> traffic.json), the bench line (in-run PMC traffic), the
# training kernel statistics and step table, all at one commit.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out/r4_final; rm -rf $O; mkdir -p $O; rm -f gpurun_out/parity_table.txt gpurun_out/train_parity_table.txt
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 1500 python -m pytest tests/ -q -m gpu 2>&1 | tail -15 ) > $O/tests.log 2>&1
( timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > $O/smoke.log 2>&1
bash tools/collect_profiles.sh fp16 $(cat .commit_id 2>/dev/null || echo unknown) > $O/collect.log 2>&1
cp gpurun_out/profiles/traffic.json profiles/r04_traffic.json
( timeout 900 python bench.py 2>&1 | grep "^{" ) > $O/bench_line.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_train -o t --output-format csv -- python $R/tools/train_bench.py --steps 3 > $O/train_rocprof.log 2>&1
cp $(find $O/prof_train -name "*kernel_stats.csv" | head -1) $O/train_kernel_stats.csv
python $R/tools/trace_train.py $O/prof_train 70 > $O/train_step_table.txt 2>&1; rm -rf $O/prof_train
cd $R
( timeout 300 python tools/train_bench.py --steps 20 2>&1 | tail -1; timeout 400 python tools/train_loop_bench.py 2>&1 | grep -v "Use norm" | tail -4 ) > $O/trainbench.log 2>&1
timeout 200 python tools/step_probe.py --batch 1 2 4 8 16 --steps 20 --dtype fp16 > $O/step_probe.log 2>&1
cp gpurun_out/parity_table.txt gpurun_out/train_parity_table.txt $O/ 2>/dev/null
tail -5 $O/tests.log; cat $O/smoke.log $O/trainbench.log; grep -v amdgpu $O/step_probe.log; cut -c1-700 $O/bench_line.json
