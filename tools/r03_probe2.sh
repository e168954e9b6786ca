set -x
mkdir -p gpurun_out
out=gpurun_out/probe2.txt
: > $out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "late or batch32 or batch64 or prefill_and_decode or ragged or eos or migrate or early_exit or graph" >> $out 2>&1
echo "== fused" >> $out; python tools/decode_probe.py 32 64 >> $out 2>&1
echo "== unfused" >> $out; MELLOW_DECODE_FUSE=0 python tools/decode_probe.py 32 64 >> $out 2>&1
echo "== fused B64" >> $out; python tools/decode_probe.py 64 64 >> $out 2>&1
tail -30 $out
