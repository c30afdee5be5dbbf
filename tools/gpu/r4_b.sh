#!/bin/bash
# round 4, lease B: new parity tests (480x480 masks, fp16 range check, B = 8 training fixture + cosines, SyncBatchNorm in DDP-wrapper
# mode), the per-epilogue attribution table (rebuilt probe libraries), kernel timelines of one forward at B = 4 and B = 1.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out/r4_b; rm -rf $O; mkdir -p $O; rm -f gpurun_out/parity_table.txt gpurun_out/train_parity_table.txt
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 900 python -m pytest tests/test_gpu_forward.py -q -m gpu -k "range_check or masks_match_the_reference_at_480" -s 2>&1 | grep -v "^$" | tail -30 ) > $O/tests_fwd_new.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_train.py -q -m gpu -k "fixtures_made_by_reference_autograd" -s 2>&1 | grep -v "^$" | tail -30 ) > $O/tests_train_fix.log 2>&1
( timeout 400 python -m pytest tests/test_gpu_train_dp.py -q -m gpu -k "sync_batchnorm_is_installed" 2>&1 | tail -15 ) > $O/tests_ddp.log 2>&1
for round in 1 2; do
  for v in full abl1 abl2 abl3 abl4; do
    L=$R/lang-seg_amd/lseg_hip/probe/liblseg_hip_$v.so; [ $v = full ] && L=$R/lang-seg_amd/lseg_hip/liblseg_hip.so
    LSEG_HIP_LIB=$L LSEG_PROBE_VARIANT=$v timeout 200 python tools/epilogue_table.py table >> $O/epi_table.jsonl 2>> $O/epi_err.log
  done
done
L=$R/lang-seg_amd/lseg_hip/probe/liblseg_hip_abl2.so
LSEG_HIP_LIB=$L LSEG_PROBE_VARIANT=abl2 timeout 300 python tools/epilogue_table.py partial >> $O/epi_partial.jsonl 2>> $O/epi_err.log
cd /tmp && export TMPDIR=/tmp
for B in 4 1; do
  timeout 300 rocprofv3 --kernel-trace -d $O/trace_b$B -o t --output-format csv -- python $R/tools/step_probe.py --batch $B --steps 6 --dtype fp16 > $O/trace_b$B.log 2>&1
  python $R/tools/trace_forward.py $O/trace_b$B > $O/timeline_b$B.txt 2>&1
  python $R/tools/trace_forward.py $O/trace_b$B all > $O/timeline_b${B}_all.txt 2>&1
  rm -rf $O/trace_b$B
done
cd $R
cp gpurun_out/parity_table.txt $O/ 2>/dev/null; cp gpurun_out/train_parity_table.txt $O/ 2>/dev/null
tail -n 12 $O/tests_fwd_new.log; tail -n 8 $O/tests_train_fix.log; tail -n 5 $O/tests_ddp.log; head -30 $O/timeline_b4.txt
