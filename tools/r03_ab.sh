# usage: bash tools/r03_ab.sh <variant> [<variant> ...]  -> decode ms of each library variant (B = 32 and 64)
mkdir -p gpurun_out
out=gpurun_out/ab.txt
: > $out
for v in "$@"; do
  lib=mellow_amd/lib/libmellow_hip_$v.so
  [ "$v" = base ] && lib=mellow_amd/lib/libmellow_hip.so
  echo "== $v" >> $out
  MELLOW_HIP_LIB=$lib python tools/decode_probe.py 32 64 2>&1 | grep decode_ms >> $out
  MELLOW_HIP_LIB=$lib python tools/decode_probe.py 64 64 2>&1 | grep decode_ms >> $out
done
cat $out
