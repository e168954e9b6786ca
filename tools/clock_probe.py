#!/usr/bin/env python3
"""Developer probe (GPU box): shader clock / power while the fp32 MFMA GEMMs run back to back (LM prefill loop),
sampled with rocm-smi from a side thread."""
import os
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mellow_amd import synth  # noqa: E402
from mellow_amd.engine import Engine  # noqa: E402

eng = Engine(device=0, max_positions=1024)
eng.load_state_dict(synth.make_state_dict(0))
prefix = torch.randn(32, 389, 576)
pd = eng._f32(prefix)
stop = False
samples = []


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            keep = [l.strip() for l in out.split("\n") if "sclk" in l or "ower" in l or "mclk" in l]
            samples.append((time.time(), keep))
        except Exception as ex:  # noqa: BLE001
            samples.append((time.time(), [repr(ex)]))
        time.sleep(0.3)


print("idle:", subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout[-900:])
eng.lm_prefill(pd, reserve=2)
th = threading.Thread(target=sampler)
th.start()
t0 = time.time()
n = 0
while time.time() - t0 < 8:
    eng.lm_prefill(pd, reserve=2)
    n += 1
stop = True
th.join()
print(f"{n} prefills in {time.time() - t0:.2f}s -> {(time.time() - t0) / n * 1e3:.2f} ms each")
for t, k in samples:
    print(f"{t - t0:6.2f}s", " | ".join(k))
