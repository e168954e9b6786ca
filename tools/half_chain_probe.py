#!/usr/bin/env python3
"""Developer probe: would two independent half-batch chains (16 + 16 examples on two streams, ONE weight copy) finish the
encoder + prefill of a 32-example batch sooner than one 32-example chain?  Uses two forked contexts driven by two threads
(max_len = 1: front-end, encoder, prefix, prefill, first token) -- the upper bound of what an in-engine two-stream prefill
could gain, without building it."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from _opts import engine_options  # noqa: E402  (--opt KEY=VALUE -> engine options)
OPTS = engine_options()
import torch
from mellow_amd import synth
from mellow_amd.engine import Engine

prec = os.environ.get("MELLOW_PRECISION", "f32x3")
eng = Engine(device=0, precision=prec, options=OPTS)
eng.load_state_dict(synth.make_state_dict(0))
ctx = [eng.fork(), eng.fork()]
a1, a2, ids = synth.make_batch(32)
full = (eng._f32(a1), eng._f32(a2), eng._i32(ids))
halves = [(eng._f32(a1[:16]), eng._f32(a2[:16]), eng._i32(ids[:16])), (eng._f32(a1[16:]), eng._f32(a2[16:]), eng._i32(ids[16:]))]


def one(n=20):
    for _ in range(3):
        eng.generate(*full, max_len=1, stop_id=0, ignore_stop=True)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n):
        eng.generate(*full, max_len=1, stop_id=0, ignore_stop=True)
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3


def two(n=20):
    def work(i, reps):
        for _ in range(reps):
            ctx[i].generate(*halves[i], max_len=1, stop_id=0, ignore_stop=True)
    for reps in (3, n):
        torch.cuda.synchronize(); t0 = time.time()
        th = [threading.Thread(target=work, args=(i, reps)) for i in range(2)]
        [t.start() for t in th]; [t.join() for t in th]
        torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3


def seq_halves(n=20):
    for _ in range(3):
        ctx[0].generate(*halves[0], max_len=1, stop_id=0, ignore_stop=True)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n):
        ctx[0].generate(*halves[0], max_len=1, stop_id=0, ignore_stop=True)
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3


print(f"{prec}: one 32-example chain {one():.2f} ms;  one 16-example chain {seq_halves():.2f} ms;  two concurrent 16-example chains {two():.2f} ms per pair")
