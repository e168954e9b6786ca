import sys, time, os; sys.path.insert(0, ".")
import numpy as np, torch
from mellow_amd import synth
from mellow_amd.engine import Engine
sd = synth.make_state_dict(0)
e0 = Engine(device=0, max_positions=1024); e0.load_state_dict(sd)
e3 = Engine(device=0, max_positions=1024, precision="f32x3"); e3.load_state_dict(sd)
B, L = 32, 64
a1, a2, ids = synth.make_batch(B)
p0, p3 = e0.prefix(a1, a2, ids), e3.prefix(a1, a2, ids)
print("prefix max|d|", float((p0 - p3).abs().max()), "of", float(p0.abs().max()))
l0, l3 = e0.lm_prefill(p0, reserve=2).cpu(), e3.lm_prefill(p0, reserve=2).cpu()
print("prefill logits max|d|", float((l0 - l3).abs().max()), "std", float(l0.std()))
t0, *_ = e0.generate(a1, a2, ids, max_len=L, stop_id=0, ignore_stop=True)
t3, *_ = e3.generate(a1, a2, ids, max_len=L, stop_id=0, ignore_stop=True)
print("tokens identical:", bool(np.array_equal(t0, t3)), "position agreement", float((t0 == t3).mean()))
g = np.load("tests/golden/gen.npz"); print("golden match:", bool(np.array_equal(t3[:2, :12], g["tokens"])))
for name, e in (("f32", e0), ("f32x3", e3)):
    a1d, a2d, idsd = e._f32(a1), e._f32(a2), e._i32(ids)
    e.generate(a1d, a2d, idsd, max_len=L, stop_id=0, ignore_stop=True)
    t = time.perf_counter()
    for _ in range(3): e.generate(a1d, a2d, idsd, max_len=L, stop_id=0, ignore_stop=True)
    dt = (time.perf_counter() - t) / 3
    print(name, f"{dt*1e3:.1f} ms -> {B/dt:.1f} responses/s", e.last_phase_ms())
