R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r6_dual; rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
P="timeout 300 python tools/dual_stream_probe.py"
( $P --parts 1
  $P --parts 2
  LSEG_GEMM_MAXGRID=128 $P --parts 2
  LSEG_GEMM_MAXGRID=128 $P --parts 2 --delay-us 700
  LSEG_GEMM_MAXGRID=128 $P --parts 2 --text-cache
  $P --parts 1 --text-cache
  LSEG_GEMM_MAXGRID=88 $P --parts 3
  LSEG_GEMM_MAXGRID=64 $P --parts 4
) > $O/log.txt 2>&1
cat $O/log.txt
