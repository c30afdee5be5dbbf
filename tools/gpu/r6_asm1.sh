R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r6_asm1; rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
( timeout 240 python tools/gemm_asm_bench.py --check-only; echo "check rc=$?" ) > $O/check.log 2>&1
cat $O/check.log | grep -v amdgpu.ids
if grep -q "check rc=0" $O/check.log; then
  ( timeout 300 python tools/gemm_asm_bench.py --shapes proj,fc2,proj_b8,fc2_b8 ) > $O/bench.log 2>&1; grep -v amdgpu.ids $O/bench.log | tail -12
fi
