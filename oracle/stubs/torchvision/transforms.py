"""TEST INFRASTRUCTURE ONLY: ToTensor/Compose with torchvision semantics for uint8 HWC numpy input
(HWC u8 -> CHW f32 / 255), which is all the reference uses (utils/setup_elements.py:29-43)."""
import numpy as np
import torch


class ToTensor:
    def __call__(self, pic):
        arr = np.asarray(pic)
        if arr.ndim == 2:
            arr = arr[:, :, None]
        t = torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1)))
        if t.dtype == torch.uint8:
            return t.to(torch.float32).div(255)
        return t


class Compose:
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x
