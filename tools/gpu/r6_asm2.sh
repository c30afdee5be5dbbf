R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r6_asm2; rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
for v in ${VARIANTS:-s4f s2f s1f s4s}; do
  echo "== variant $v"
  LSEG_HIP_LIB=$R/lang-seg_amd/lseg_hip/probe/liblseg_hip_$v.so timeout 300 python tools/gemm_asm_bench.py --shapes ${SHAPES:-proj,fc2} 2>&1 | grep -v amdgpu.ids | grep "MISMATCH\|asm\|generic\|Error\|error"
done > $O/log.txt 2>&1
cat $O/log.txt
