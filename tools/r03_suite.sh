set -x
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x "$@" > gpurun_out/gputest.log 2>&1; tail -15 gpurun_out/gputest.log
python tools/decode_probe.py 32 64 2>&1 | grep -v amdgpu.ids | tee gpurun_out/probe_after_suite.txt
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/probe_after_suite.txt
import sys; sys.path.insert(0, ".")
from mellow_amd import synth
from mellow_amd.engine import Engine
for prec in ("f32x3", "f32"):
    eng = Engine(device=0, precision=prec); eng.load_state_dict(synth.make_state_dict(0))
    a1, a2, ids = synth.make_batch(32)
    a1d, a2d, idsd = eng._f32(a1), eng._f32(a2), eng._i32(ids)
    for _ in range(3):
        eng.generate(a1d, a2d, idsd, max_len=64, stop_id=0, ignore_stop=True)
    print(prec, eng.last_phase_ms())
    eng.close()
PY
