#!/bin/bash
# Variant libraries of the hand-scheduled GEMM for A/B runs (loaded through LSEG_HIP_LIB): tools/build_asm_variants.sh "tag:ENV=1,ENV2=2" ...
# -> lang-seg_amd/lseg_hip/probe/liblseg_hip_<tag>.so (same objects, gemm_asm.o rebuilt from a body generated under the given environment)
set -e
cd "$(dirname "$0")/../lang-seg_amd/csrc"
make -j8 >/dev/null
mkdir -p ../lseg_hip/probe build
OBJS=$(ls build/*.o | grep -v "gemm_asm\|gemm_abl")
for spec in "$@"; do
  tag=${spec%%:*}; envs=${spec#*:}; [ "$envs" = "$spec" ] && envs=""
  env $(echo $envs | tr ',' ' ') python3 gemm_asm_gen.py build/gemm_asm_body.inc 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-inline-asm -c gemm_asm.hip -o build/gemm_asm_v_$tag.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lseg_hip/probe/liblseg_hip_$tag.so build/gemm_asm_v_$tag.o $OBJS
  echo "built probe/liblseg_hip_$tag.so ($envs)"
done
python3 gemm_asm_gen.py build/gemm_asm_body.inc 2>/dev/null      # back to the default body
touch build/gemm_asm_body.inc
