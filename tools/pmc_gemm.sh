#!/bin/bash
# PMC passes over the GEMM micro-benchmark (counters only; no trace domains besides kernel-trace)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; shift; SHAPES="$@"
mkdir -p $OUT
run() { n=$1; shift; ITERS=3 rocprofv3 --pmc "$@" --kernel-trace -d $OUT/$n -o $n --output-format csv -- python $R/tools/gemm_bench.py $SHAPES > $OUT/$n.log 2>&1; }
run A SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES
run B SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_IFETCH SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES
run C TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum
run D TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY
