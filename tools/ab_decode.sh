#!/bin/bash
# Developer tool: a variant of libmellow_hip.so in which ONLY decode.hip is rebuilt with extra -D flags (the other objects come
# from the release build's cache, mellow_amd/csrc/build/): seconds instead of minutes per variant.
#   tools/ab_decode.sh nohoist "-DMELLOW_NO_HOIST"   ->  mellow_amd/lib/ab/libmellow_hip_nohoist.so
# Run `python mellow_amd/csrc/build.py` first.  Use with MELLOW_HIP_LIB=... python tools/decode_probe.py, or tools/ab_run.sh.
set -e
cd "$(dirname "$0")/.."
name=$1; flags=$2
mkdir -p mellow_amd/lib/ab
out=mellow_amd/lib/ab/libmellow_hip_$name.so
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wno-unused-function -x hip $flags -mllvm -amdgpu-kernarg-preload-count=14 \
    -c mellow_amd/csrc/decode.hip -o $tmp/decode.hip.o
objs=$(ls mellow_amd/csrc/build/*.o | grep -v decode.hip.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $out $objs $tmp/decode.hip.o
rm -rf $tmp
echo $out
