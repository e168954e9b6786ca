set -x
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x -k "pipelined or eos or migrate or early_exit or raw or stop or nan" > gpurun_out/gputest_sub.log 2>&1; tail -5 gpurun_out/gputest_sub.log
python tools/fp8_agreement.py 2>&1 | grep -v amdgpu.ids > gpurun_out/fp8_agree.txt
python tools/fp8_agreement.py structured 2>&1 | grep -v amdgpu.ids >> gpurun_out/fp8_agree.txt
cat gpurun_out/fp8_agree.txt
python tools/decode_probe.py 32 64 2>&1 | grep decode_ms
python tools/decode_probe.py 64 64 2>&1 | grep decode_ms
