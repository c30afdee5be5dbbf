#!/bin/bash
# round 5, last lease (after the deterministic-by-default change): the whole GPU suite with -x, smoke and the default bench line (sources
# identical to 533c8dd, where tools/gpu/r5_final.sh also collected the rocprofv3 / PMC / training tables); + the bench at 72 images per step
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r5_final3; rm -rf $O; mkdir -p $O; rm -f gpurun_out/parity_table.txt gpurun_out/train_parity_table.txt
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
COMMIT=$(cat lang-seg_amd/lseg_hip/_build_id.txt 2>/dev/null || echo unknown)
( echo "library built at commit $COMMIT; python -m pytest tests/ -q -m gpu -x"; timeout 1700 python -m pytest tests/ -q -m gpu -x 2>&1 | tail -25 ) > $O/tests.log 2>&1
( timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > $O/smoke.log 2>&1
( timeout 900 python bench.py 2>&1 | grep "^{" ) > $O/bench_line.json
cp gpurun_out/parity_table.txt gpurun_out/train_parity_table.txt $O/ 2>/dev/null
tail -4 $O/tests.log; cat $O/smoke.log; cut -c1-300 $O/bench_line.json
