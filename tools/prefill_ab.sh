for v in "$@"; do
lib=mellow_amd/lib/libmellow_hip_$v.so
[ "$v" = base ] && lib=mellow_amd/lib/libmellow_hip.so
echo "== $v"
MELLOW_HIP_LIB=$lib python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys; sys.path.insert(0, ".")
from mellow_amd import synth
from mellow_amd.engine import Engine
eng = Engine(device=0, precision="f32x3"); eng.load_state_dict(synth.make_state_dict(0))
a1, a2, ids = synth.make_batch(32)
a1d, a2d, idsd = eng._f32(a1), eng._f32(a2), eng._i32(ids)
ph = []
for _ in range(6):
    t, *_ = eng.generate(a1d, a2d, idsd, max_len=2, stop_id=0, ignore_stop=True)
    ph.append(eng.last_phase_ms())
print("f32x3", {k: round(min(p[k] for p in ph), 3) for k in ph[0]}, t[:2, :2].tolist())
eng.prof_enable(True); eng.prof_reset(); eng.generate(a1d, a2d, idsd, max_len=2, stop_id=0, ignore_stop=True)
print({k: (v["launches"], round(v["ms"], 3)) for k, v in eng.prof_report().items()})
PY
done
