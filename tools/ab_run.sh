# usage: bash tools/ab_run.sh <variant> [<variant> ...]   (variants built by tools/ab_build.sh; "base" = the release library) -> decode ms per 63 steps at B = 32 and 64
mkdir -p gpurun_out
out=gpurun_out/ab3.txt
: > $out
for v in "$@"; do
  lib=mellow_amd/lib/ab/libmellow_hip_$v.so
  [ "$v" = base ] && lib=mellow_amd/lib/libmellow_hip.so
  echo "== $v" >> $out
  for B in 32 64; do MELLOW_HIP_LIB=$lib timeout 300 python tools/decode_probe.py $B 64 2>&1 | grep decode_ms >> $out; done
done
cat $out
