mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r05_gputest_c.log 2>&1; tail -6 gpurun_out/r05_gputest_c.log
echo "== f32x3"; for B in 32 64; do timeout 300 python tools/decode_probe.py $B 64 2>&1 | grep decode_ms; done
echo "== fp8 kv16 paired"; for B in 32 128; do MELLOW_PRECISION=fp8 timeout 300 python tools/decode_probe.py $B 64 2>&1 | grep decode_ms; done
timeout 600 python tools/fp8_agreement.py structured 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_fp8_agreement_kv16.txt; tail -4 gpurun_out/r05_fp8_agreement_kv16.txt
