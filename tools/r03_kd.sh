set -x
mkdir -p gpurun_out
: > gpurun_out/kd.txt
echo "== fused" >> gpurun_out/kd.txt
MELLOW_HIP_LIB=mellow_amd/lib/libmellow_hip_kd.so python tools/kdebug.py >> gpurun_out/kd.txt 2>&1
echo "== unfused" >> gpurun_out/kd.txt
MELLOW_DECODE_FUSE=0 MELLOW_HIP_LIB=mellow_amd/lib/libmellow_hip_kd.so python tools/kdebug.py >> gpurun_out/kd.txt 2>&1
cat gpurun_out/kd.txt
