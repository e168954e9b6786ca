mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fp8" > gpurun_out/t1.log 2>&1; tail -4 gpurun_out/t1.log
timeout 300 python bench.py --steps 3 --warmup 1 --preset configs4 --no-cpu-baseline --no-alt-modes --no-b64 --no-configs3 --inflight 0 > gpurun_out/r05_bench_configs4.json 2>/dev/null
timeout 300 python bench.py --steps 5 --warmup 2 --precision fp8 --no-cpu-baseline --no-alt-modes --no-b64 --no-configs3 --inflight 0 > gpurun_out/r05_bench_fp8.json 2>/dev/null
python - <<'PY'
import json
for f in ("gpurun_out/r05_bench_configs4.json","gpurun_out/r05_bench_fp8.json"):
    d=json.load(open(f)); print(f, d["value"], d["phase_ms"], d.get("roofline",{}) and d["roofline"].get("frac"))
PY
