bash tools/r03_ab.sh "$@" > /dev/null 2>&1
echo "== kdebug (default build of this tree with stamps)" >> gpurun_out/ab.txt
MELLOW_HIP_LIB=mellow_amd/lib/libmellow_hip_kd.so python tools/kdebug.py 2>&1 | grep -A1 "dec_attn\|dec_qkv2" >> gpurun_out/ab.txt
cat gpurun_out/ab.txt
