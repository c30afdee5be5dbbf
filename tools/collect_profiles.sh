#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats of the bench command + separate PMC passes (counters only, kernel-trace domain
# only) for the dominant kernels.  Output: gpurun_out/profiles/ ; the summaries are then copied into profiles/ as <round>_* (profiles/README).
# usage: collect_profiles.sh <dtype> [commit]
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/profiles; rm -rf $OUT; mkdir -p $OUT
DT=${1:-fp16}; COMMIT=${2:-unknown}
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 5 --warmup 2 --dtype $DT --no-parity --no-cpu-baseline --no-sweep --no-pmc-traffic"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- $CMD > $OUT/bench_under_rocprof.log 2>&1
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace -d $OUT/pmc_$tag -o pmc --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --dtype $DT --no-parity --no-cpu-baseline --no-sweep --no-pmc-traffic > $OUT/pmc_$tag.log 2>&1
done
python $R/tools/summarize_profiles.py $OUT > $OUT/summary.txt 2>&1
python $R/tools/make_traffic_json.py $OUT/summary.json 36 $OUT/traffic.json $DT $COMMIT >> $OUT/summary.txt 2>&1
cp $OUT/stats/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null || cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/stats $OUT/pmc_*/   # raw traces are large; the summaries above are what gets committed
head -60 $OUT/summary.txt
