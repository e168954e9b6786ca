set -x
mkdir -p gpurun_out
out=gpurun_out/probe1.txt
: > $out
P=mellow_amd/lib/libmellow_hip_probe.so
run() { echo "== $*" >> $out; env "$@" python tools/decode_probe.py 32 64 >> $out 2>&1; }
run MELLOW_HIP_LIB=$P
run MELLOW_HIP_LIB=$P MELLOW_DEV_SKIP=4
run MELLOW_HIP_LIB=$P MELLOW_DEV_SKIP=16
run MELLOW_HIP_LIB=$P MELLOW_DEV_SKIP=20
run MELLOW_HIP_LIB=$P MELLOW_DEV_SKIP=1
run MELLOW_HIP_LIB=$P MELLOW_DEV_SKIP=8
run MELLOW_HIP_LIB=$P MELLOW_DEV_SKIP=2
run MELLOW_HIP_LIB=$P MELLOW_DEV_DEAD_BLOCKS=1
run MELLOW_HIP_LIB=mellow_amd/lib/libmellow_hip_ts1w16.so
echo "== B64" >> $out
MELLOW_HIP_LIB=$P python tools/decode_probe.py 64 64 >> $out 2>&1
MELLOW_HIP_LIB=mellow_amd/lib/libmellow_hip_ts1w16.so python tools/decode_probe.py 64 64 >> $out 2>&1
cat $out
