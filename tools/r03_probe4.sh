set -x
mkdir -p gpurun_out
out=gpurun_out/probe4.txt
: > $out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "late or batch32 or batch64 or prefill_and_decode or ragged or eos or migrate or early_exit or graph or pipelined or 2048 or independent_prefill or nan or reproduc" >> $out 2>&1
python tools/decode_probe.py 32 64 2>&1 | grep decode_ms >> $out
python tools/decode_probe.py 64 64 2>&1 | grep decode_ms >> $out
tail -6 $out
