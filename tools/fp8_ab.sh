# fp8 mode A/B on one box: activation quantisation on/off, fused down+qkv launch on/off (gpurun -- bash tools/fp8_ab.sh)
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "fp8 or fused" 2>&1 | tail -5
for cfg in "1 1" "1 0" "0 1"; do
  set -- $cfg
  MELLOW_FP8_DECODE_ACT=$1 MELLOW_DECODE_FUSE=$2 timeout 300 python bench.py --steps 5 --warmup 2 --precision fp8 --no-cpu-baseline --no-b64 --inflight 0 --no-alt-modes 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('act $1 fuse $2 B32', d['value'], d['phase_ms'])"
  MELLOW_FP8_DECODE_ACT=$1 MELLOW_DECODE_FUSE=$2 timeout 300 python bench.py --steps 3 --warmup 1 --preset configs4 --no-cpu-baseline --no-b64 --inflight 0 --no-alt-modes 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('act $1 fuse $2 B128', d['value'], d['phase_ms'])"
done
