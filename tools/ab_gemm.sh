#!/bin/bash
# Developer tool: a variant of libmellow_hip.so in which ONLY gemm_bf16x3.hip is rebuilt with extra -D flags (the other objects
# come from the release build's cache, mellow_amd/csrc/build/): seconds per variant.
#   tools/ab_gemm.sh nodma "-DMELLOW_X3R_ABL=1"   ->  mellow_amd/lib/ab/libmellow_hip_nodma.so
# Run `python mellow_amd/csrc/build.py` first.  Use with MELLOW_HIP_LIB=... python tools/x3r_probe.py
set -e
cd "$(dirname "$0")/.."
name=$1; flags=$2
mkdir -p mellow_amd/lib/ab
out=mellow_amd/lib/ab/libmellow_hip_$name.so
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wno-unused-function -x hip $flags -mllvm -amdgpu-mfma-vgpr-form=1 \
    -c mellow_amd/csrc/gemm_bf16x3.hip -o $tmp/gemm_bf16x3.hip.o
objs=$(ls mellow_amd/csrc/build/*.o | grep -v gemm_bf16x3.hip.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $out $objs $tmp/gemm_bf16x3.hip.o
rm -rf $tmp
echo $out
