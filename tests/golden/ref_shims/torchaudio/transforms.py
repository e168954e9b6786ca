"""Shim (see __init__.py): torchaudio.transforms.Resample(orig, new) with torchaudio's defaults, restated in
mellow_amd/audio.py.  PARITY UNPINNED: torchaudio itself is not installed here."""
import torch


class Resample(torch.nn.Module):
    def __init__(self, orig_freq=16000, new_freq=16000):
        super().__init__()
        self.orig_freq, self.new_freq = int(orig_freq), int(new_freq)

    def forward(self, waveform):
        from mellow_amd import audio
        return audio.resample(waveform, self.orig_freq, self.new_freq)
