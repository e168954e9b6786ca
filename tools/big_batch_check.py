import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from _opts import engine_options  # noqa: E402  (--opt KEY=VALUE -> engine options)
OPTS = engine_options()
from mellow_amd import synth
from mellow_amd.engine import Engine
sd = synth.make_state_dict(0)
for prec in ("f32x3", "fp8"):
    e = Engine(device=0, precision=prec, options=OPTS)
    e.load_state_dict(sd)
    a1, a2, ids = synth.make_batch(32)
    t32, *_ = e.generate(a1, a2, ids, max_len=64, stop_id=0, ignore_stop=True)
    for B in (256, 512, 1024):
        a1b, a2b, idsb = synth.make_batch(B)
        torch.cuda.synchronize(); t0 = time.time()
        tb, lens, n, ftm = e.generate(a1b, a2b, idsb, max_len=64, stop_id=0, ignore_stop=True)
        torch.cuda.synchronize(); dt = time.time() - t0
        t0 = time.time()
        tb2, *_ = e.generate(a1b, a2b, idsb, max_len=64, stop_id=0, ignore_stop=True)
        torch.cuda.synchronize(); dt2 = time.time() - t0
        print(prec, B, "same as B=32 rows:", np.array_equal(tb[:32], t32), "deterministic:", np.array_equal(tb, tb2),
              f"{B/dt2:.1f} responses/s (second call), mem {torch.cuda.mem_get_info()[0]/1e9:.1f} GB free", flush=True)
    e.close()
