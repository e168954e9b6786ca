"""Shim for torchlibrosa.augmentation — training-only module, identity here (never run in eval)."""
import torch.nn as nn


class SpecAugmentation(nn.Module):
    def __init__(self, time_drop_width, time_stripes_num, freq_drop_width, freq_stripes_num):
        super().__init__()

    def forward(self, input):
        return input
