#!/bin/bash
# Developer tool: build a second copy of libmellow_hip.so with extra -D flags for same-box A/B runs.
#   tools/ab_build.sh variantB "-DMELLOW_NO_NT"   ->  mellow_amd/lib/ab/libmellow_hip_variantB.so
# Use it with MELLOW_HIP_LIB=mellow_amd/lib/ab/libmellow_hip_variantB.so python tools/decode_probe.py
set -e
cd "$(dirname "$0")/.."
name=$1; flags=$2
mkdir -p mellow_amd/lib/ab
out=mellow_amd/lib/ab/libmellow_hip_$name.so
tmp=$(mktemp -d)
for f in gemm_f32.hip gemm_fp8.hip gemm_bf16x3.hip decode.hip prefill_attn.hip encoder.hip stft_fft.hip engine.cpp engine_weights.cpp engine_encoder.cpp engine_lm.cpp engine_dev.cpp; do
  extra=""
  case $f in gemm_bf16x3.hip|gemm_fp8.hip|prefill_attn.hip|encoder.hip) extra="-mllvm -amdgpu-mfma-vgpr-form=1";;
            decode.hip) extra="-mllvm -amdgpu-kernarg-preload-count=14";; esac     # the per-file flags of build.py
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wno-unused-function -x hip $flags $extra -c mellow_amd/csrc/$f -o $tmp/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $out $tmp/*.o
rm -rf $tmp
echo $out
