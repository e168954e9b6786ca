"""Shim for torchlibrosa.stft (0.1.0) — op sequence only; see ../README.md (parity unpinned)."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class STFT(nn.Module):
    def __init__(self, n_fft, hop_length, win_length, window, center, pad_mode, freeze_parameters):
        super().__init__()
        assert window == "hann" and win_length == n_fft
        self.n_fft, self.hop_length, self.center, self.pad_mode = n_fft, hop_length, center, pad_mode
        out = n_fft // 2 + 1
        self.conv_real = nn.Conv1d(1, out, n_fft, stride=hop_length, padding=0, dilation=1, groups=1, bias=False)
        self.conv_imag = nn.Conv1d(1, out, n_fft, stride=hop_length, padding=0, dilation=1, groups=1, bias=False)
        # constants are overwritten by the checkpoint (strict load); zero here on purpose
        nn.init.zeros_(self.conv_real.weight)
        nn.init.zeros_(self.conv_imag.weight)
        if freeze_parameters:
            for p in self.parameters():
                p.requires_grad = False

    def forward(self, input):
        x = input[:, None, :]
        if self.center:
            x = F.pad(x, pad=(self.n_fft // 2, self.n_fft // 2), mode=self.pad_mode)
        real = self.conv_real(x)
        imag = self.conv_imag(x)
        real = real[:, None, :, :].transpose(2, 3)
        imag = imag[:, None, :, :].transpose(2, 3)
        return real, imag


class Spectrogram(nn.Module):
    def __init__(self, n_fft=2048, hop_length=None, win_length=None, window="hann", center=True,
                 pad_mode="reflect", power=2.0, freeze_parameters=True):
        super().__init__()
        self.power = power
        self.stft = STFT(n_fft=n_fft, hop_length=hop_length, win_length=win_length, window=window,
                         center=center, pad_mode=pad_mode, freeze_parameters=True)

    def forward(self, input):
        real, imag = self.stft.forward(input)
        spectrogram = real ** 2 + imag ** 2
        if self.power == 2.0:
            pass
        else:
            spectrogram = spectrogram ** (self.power / 2.0)
        return spectrogram


class LogmelFilterBank(nn.Module):
    def __init__(self, sr=22050, n_fft=2048, n_mels=64, fmin=0.0, fmax=None, is_log=True, ref=1.0,
                 amin=1e-10, top_db=80.0, freeze_parameters=True):
        super().__init__()
        self.is_log, self.ref, self.amin, self.top_db = is_log, ref, amin, top_db
        self.melW = nn.Parameter(torch.zeros(n_fft // 2 + 1, n_mels))
        if freeze_parameters:
            for p in self.parameters():
                p.requires_grad = False

    def forward(self, input):
        mel_spectrogram = torch.matmul(input, self.melW)
        if self.is_log:
            return self.power_to_db(mel_spectrogram)
        return mel_spectrogram

    def power_to_db(self, input):
        ref_value = self.ref
        log_spec = 10.0 * torch.log10(torch.clamp(input, min=self.amin, max=float("inf")))
        log_spec -= 10.0 * math.log10(max(self.amin, ref_value))
        if self.top_db is not None:
            if self.top_db < 0:
                raise ValueError("top_db must be non-negative")
            log_spec = torch.clamp(log_spec, min=log_spec.max().item() - self.top_db, max=float("inf"))
        return log_spec
