"""Shim used only by tests/golden/make_golden.py (case "example"): the two calls the reference makes into torchaudio
(wrapper.py:144-148).  `load` decodes 16-bit PCM WAV exactly like torchaudio.load(normalize=True) does (int16 / 32768 as
float32, shape (channels, frames)); `transforms.Resample` is the independent fp64 oracle of torchaudio's published default
sinc-Hann algorithm (oracle/resample_oracle.py; nothing of the product is imported), see ../README.md."""
import wave

import numpy as np
import torch

from . import transforms  # noqa: F401


def load(path):
    with wave.open(str(path), "rb") as w:
        sr, ch, sw, n = w.getframerate(), w.getnchannels(), w.getsampwidth(), w.getnframes()
        assert sw == 2, "the shim decodes 16-bit PCM only"
        data = np.frombuffer(w.readframes(n), dtype="<i2").reshape(-1, ch)
    return torch.from_numpy((data.astype(np.float32) / 32768.0).T.copy()), sr
