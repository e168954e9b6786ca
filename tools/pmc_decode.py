#!/usr/bin/env python3
"""Small workload for the rocprofv3 --pmc passes of the decode step (counter collection costs ~0.1 s per dispatch, so the
4 x 63-step probe is out of reach): ONE generate() call at B = 32 with max_len = 6 -> 5 decode steps (contexts 389..393) + the
prefill's own last-position step, eager launches (no graph replay)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from _opts import engine_options  # noqa: E402  (--opt KEY=VALUE -> engine options)
OPTS = engine_options()
from mellow_amd import synth
from mellow_amd.engine import Engine
eng = Engine(device=0, precision=os.environ.get("MELLOW_PRECISION", "f32x3"), options=OPTS)
eng.load_state_dict(synth.make_state_dict(0))
eng.set_graph(False)
a1, a2, ids = synth.make_batch(32)
a1d, a2d, idsd = eng._f32(a1), eng._f32(a2), eng._i32(ids)
eng.generate(a1d, a2d, idsd, max_len=6, stop_id=0, ignore_stop=True)
print("done")
