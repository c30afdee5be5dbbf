#!/bin/bash
# Runs on the GPU box (via gpurun): the training step (tools/train_bench.py, B = 8) under rocprofv3 -- kernel trace + separate PMC passes
# (counters only, kernel-trace domain only).  Output: gpurun_out/train_profiles/summary.txt (+ summary.json)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/train_profiles; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- python $R/tools/train_bench.py --steps 3 > $OUT/under_rocprof.log 2>&1
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 400 rocprofv3 --pmc $pass --kernel-trace -d $OUT/pmc_$tag -o pmc --output-format csv -- python $R/tools/train_bench.py --steps 1 > $OUT/pmc_$tag.log 2>&1
done
python $R/tools/summarize_profiles.py $OUT 14 > $OUT/summary.txt 2>&1
rm -rf $OUT/stats $OUT/pmc_*/
head -50 $OUT/summary.txt
