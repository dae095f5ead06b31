"""TEST INFRASTRUCTURE ONLY. kornia==0.4.1 is pinned by the reference (requirements.txt:11) but is
neither installed nor vendored, so its RNG parameterisation cannot be reproduced: PARITY UNPINNED.
The reference's SCR agent (agents/scr.py:18-24) is imported with these identity modules; SCR parity
runs inject the same deterministic augmentation on both sides."""
import torch.nn as nn


class _Identity(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()

    def forward(self, x):
        return x


RandomResizedCrop = RandomHorizontalFlip = ColorJitter = RandomGrayscale = _Identity
