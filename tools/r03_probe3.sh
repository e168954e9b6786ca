set -x
mkdir -p gpurun_out
out=gpurun_out/probe3.txt
: > $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_raw_abi.py -x -q -k "late or batch32 or batch64 or prefill_and_decode or ragged or eos or migrate or early_exit or graph or raw or pipelined" >> $out 2>&1
echo "== gu2+qkv2" >> $out; python tools/decode_probe.py 32 64 2>&1 | grep decode_ms >> $out
echo "== qkv2 only" >> $out; MELLOW_DECODE_FUSE=1 python tools/decode_probe.py 32 64 2>&1 | grep decode_ms >> $out
echo "== unfused" >> $out; MELLOW_DECODE_FUSE=0 python tools/decode_probe.py 32 64 2>&1 | grep decode_ms >> $out
tail -12 $out
bash tools/r03_trace.sh gu2 > /dev/null 2>&1; cat gpurun_out/trace_gu2.txt | head -8
