# usage: bash tools/ab_fp8_b128.sh <variant> [...]   (variants from tools/ab_build.sh / ab_decode.sh; "base" = the release library)
# -> decode ms per 63 steps in the fp8 mode at B = 128 (BASELINE configs[4]'s per-GPU batch) and in the default mode at B = 64
mkdir -p gpurun_out
out=gpurun_out/ab_fp8_b128.txt
: > $out
for v in "$@"; do
  lib=mellow_amd/lib/ab/libmellow_hip_$v.so
  [ "$v" = base ] && lib=mellow_amd/lib/libmellow_hip.so
  echo "== $v" >> $out
  MELLOW_PRECISION=fp8 MELLOW_HIP_LIB=$lib timeout 300 python tools/decode_probe.py 128 64 2>&1 | grep decode_ms | sed 's/^/fp8 B=128 /' >> $out
  MELLOW_HIP_LIB=$lib timeout 300 python tools/decode_probe.py 64 64 2>&1 | grep decode_ms | sed 's/^/x3  B=64  /' >> $out
done
cat $out
