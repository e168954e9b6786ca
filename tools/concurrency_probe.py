#!/usr/bin/env python3
"""Developer probe (GPU box): aggregate throughput of K engine contexts driven from K host threads on ONE GPU.
Answers: do the latency-bound decode chains of one context overlap with the dense phases of another?"""
from __future__ import annotations

import argparse
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mellow_amd import synth  # noqa: E402
from mellow_amd.engine import Engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--contexts", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--max-len", type=int, default=64)
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--stagger-ms", type=float, default=0.0)
    args = ap.parse_args()
    sd = synth.make_state_dict(0)
    engs = []
    for i in range(args.contexts):
        e = Engine(device=0, max_positions=1024)
        e.load_state_dict(sd)
        engs.append(e)
    B, L = args.batch, args.max_len
    ins = []
    for i, e in enumerate(engs):
        a1, a2, ids = synth.make_batch(B, first=i * B)
        ins.append((e._f32(a1), e._f32(a2), e._i32(ids)))

    def run(i, n):
        e = engs[i]
        a1, a2, ids = ins[i]
        if args.stagger_ms and i:
            time.sleep(i * args.stagger_ms * 1e-3)
        for _ in range(n):
            e.generate(a1, a2, ids, max_len=L, stop_id=0, ignore_stop=True)

    for i in range(args.contexts):
        run(i, 2)
    # one context alone
    t0 = time.perf_counter()
    run(0, args.iters)
    t1 = time.perf_counter() - t0
    print(f"1 context : {B * args.iters / t1:8.1f} responses/s   ({t1 / args.iters * 1e3:.1f} ms per batch)")
    th = [threading.Thread(target=run, args=(i, args.iters)) for i in range(args.contexts)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    tk = time.perf_counter() - t0
    print(f"{args.contexts} contexts: {args.contexts * B * args.iters / tk:8.1f} responses/s   "
          f"({tk / args.iters * 1e3:.1f} ms per round of {args.contexts} batches)")


if __name__ == "__main__":
    main()
