#!/bin/bash
# ONE lease script for the GPU box (replaces the per-round r4_* / r5_* / r6_* files):
#   gpurun --timeout T -- 'bash tools/gpu/lease.sh <name> <task> [<task> ...]'
# Output goes to gpurun_out/<name>/ (merged back by gpurun); copy what should be judged into profiles/ as rNN_*.
# Tasks:
#   fast        pytest -m gpu_fast (~4 min: every operator test, one fixture per network family, one training fixture)
#   full        the whole GPU suite (pytest -m gpu -x, ~11 min) + smoke
#   smoke       __graft_entry__.smoke()
#   bench       the default bench line (python bench.py) -> bench_line.json
#   profile     rocprofv3 --kernel-trace --stats of the bench command + the separate PMC passes -> rocprofv3_summary.txt, kernel_stats.csv, traffic.json
#   train       tools/train_bench.py (20 steps), tools/train_loop_bench.py, the training kernel statistics and step table under rocprofv3
#   asm         tools/gemm_asm_bench.py (the hand-scheduled residual GEMM against the generic kernels)
#   cmd:<...>   any other command line (quote it), logged to cmd.log
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
NAME=${1:?lease name}; shift
O=$R/gpurun_out/$NAME; rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
COMMIT=$(cat lang-seg_amd/lseg_hip/_build_id.txt 2>/dev/null || echo unknown)
rm -f gpurun_out/parity_table.txt gpurun_out/train_parity_table.txt
for task in "$@"; do
  echo "== $task (library built at $COMMIT)"
  case "$task" in
    fast)  ( timeout 900 python -m pytest tests/ -q -m gpu_fast -x 2>&1 | tail -15 ) > $O/tests_fast.log 2>&1; tail -4 $O/tests_fast.log ;;
    full)  ( echo "library built at commit $COMMIT; python -m pytest tests/ -q -m gpu -x"; timeout 2400 python -m pytest tests/ -q -m gpu -x 2>&1 | tail -25 ) > $O/tests.log 2>&1
           ( timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > $O/smoke.log 2>&1
           cp gpurun_out/parity_table.txt gpurun_out/train_parity_table.txt $O/ 2>/dev/null; tail -6 $O/tests.log; cat $O/smoke.log ;;
    smoke) ( timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > $O/smoke.log 2>&1; cat $O/smoke.log ;;
    bench) ( timeout 1200 python bench.py 2>$O/bench_stderr.log | grep "^{" ) > $O/bench_line.json; cut -c1-700 $O/bench_line.json ;;
    profile) bash tools/collect_profiles.sh fp16 $COMMIT > $O/collect.log 2>&1
           cp gpurun_out/profiles/traffic.json $O/traffic.json; cp gpurun_out/profiles/summary.txt $O/rocprofv3_summary.txt; cp gpurun_out/profiles/kernel_stats.csv $O/kernel_stats.csv
           head -40 $O/rocprofv3_summary.txt ;;
    train) ( timeout 300 python tools/train_bench.py --steps 20 2>&1 | tail -1; timeout 400 python tools/train_loop_bench.py 2>&1 | grep -v "Use norm" | tail -4 ) > $O/trainbench.log 2>&1
           ( cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_train -o t --output-format csv -- python $R/tools/train_bench.py --steps 3 > $O/train_rocprof.log 2>&1
             cp $(find $O/prof_train -name "*kernel_stats.csv" | head -1) $O/train_kernel_stats.csv
             python $R/tools/trace_train.py $O/prof_train 70 > $O/train_step_table.txt 2>&1; rm -rf $O/prof_train )
           cat $O/trainbench.log ;;
    asm)   ( timeout 600 python tools/gemm_asm_bench.py --shapes proj,fc2,proj_b8,fc2_b8 2>&1 | grep -v amdgpu.ids ) > $O/gemm_asm.log 2>&1; tail -12 $O/gemm_asm.log ;;
    cmd:*) ( timeout ${CMD_TIMEOUT:-900} bash -c "${task#cmd:}" 2>&1 | grep -v amdgpu.ids ) > $O/cmd.log 2>&1; tail -${CMD_TAIL:-30} $O/cmd.log ;;
    *) echo "unknown task $task" ;;
  esac
done
