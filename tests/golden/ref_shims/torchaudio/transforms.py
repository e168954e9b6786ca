"""Shim (see __init__.py): `torchaudio.transforms.Resample(orig, new)` with torchaudio's defaults, served by the INDEPENDENT fp64
oracle of the published algorithm (oracle/resample_oracle.py: per-output-sample windowed-sinc sums, nothing shared with the
product's resamplers), rounded once to float32 like torchaudio's float32 output.  torchaudio itself is not installed here:
pinned to the published algorithm, not to torchaudio's binary."""
import os
import sys

import numpy as np
import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)


class Resample(torch.nn.Module):
    def __init__(self, orig_freq=16000, new_freq=16000):
        super().__init__()
        self.orig_freq, self.new_freq = int(orig_freq), int(new_freq)

    def forward(self, waveform):
        from oracle import resample_oracle
        y = resample_oracle.resample(waveform.detach().cpu().numpy(), self.orig_freq, self.new_freq)
        return torch.from_numpy(np.ascontiguousarray(y.astype(np.float32)))
