"""Host side of the step's data path: continuum/data_utils.py:38-64 (dataset_transform + DataLoader) re-done for a
device-resident task.

The reference builds `DataLoader(dataset_transform(x, y, ToTensor()), batch, shuffle=True, drop_last=True,
num_workers=0)` (agents/exp_replay.py:21-23): per item HWC uint8 -> CHW float32 / 255 on the host.  Here the
uint8 task tensor is uploaded once, the *same* torch DataLoader machinery runs over bare indices (so the torch
RNG draws — one for the iterator's base seed, one for the RandomSampler seed — and the resulting batch order are
identical by construction), and each minibatch is produced on the GPU by one gather+convert kernel.
"""
import numpy as np
import torch
from torch.utils import data

from . import ops


class _IndexDataset(data.Dataset):
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return i


class DeviceLoader(object):
    """Iterable of (batch_x float32 [b,C,H,W] on GPU, batch_y int64 [b] on GPU); `last_y_host` holds the numpy
    labels of the batch just yielded (lets plugins keep host-side bookkeeping without a device sync)."""

    def __init__(self, x_u8_nhwc, y, batch_size, shuffle, drop_last=False, device=None):
        if isinstance(x_u8_nhwc, np.ndarray):
            if x_u8_nhwc.dtype != np.uint8:
                raise TypeError("DeviceLoader expects uint8 HWC images (what ToTensor() scales by 1/255)")
            x_u8_nhwc = torch.from_numpy(np.ascontiguousarray(x_u8_nhwc))
        device = device or torch.device("cuda", torch.cuda.current_device())
        self.x = x_u8_nhwc.to(device).contiguous()
        y_np = np.asarray(y).astype(np.int64)
        self.y_host = y_np
        self.y = torch.from_numpy(y_np).to(device)
        self.batch_size = batch_size
        self._index_loader = data.DataLoader(_IndexDataset(len(y_np)), batch_size=batch_size, shuffle=shuffle,
                                             num_workers=0, drop_last=drop_last)
        self.last_y_host = None
        self.last_index_host = None

    def __len__(self):
        return len(self._index_loader)

    def __iter__(self):
        # materialise the epoch's index batches first: both RNG draws happen here, exactly where the reference's
        # `for i, batch in enumerate(train_loader)` makes them (iterator creation + first next()).
        batches = [b for b in iter(self._index_loader)]
        if not batches:
            return
        sizes = [int(b.numel()) for b in batches]
        perm_host = torch.cat(batches)
        perm_dev = perm_host.to(self.x.device)
        perm_np = perm_host.numpy()
        start = 0
        for sz in sizes:
            idx = perm_dev[start:start + sz]
            self.last_index_host = perm_np[start:start + sz]
            self.last_y_host = self.y_host[self.last_index_host]
            bx = ops.gather_u8_images(self.x, idx)
            by = ops.gather_rows(self.y, idx)
            by.host = self.last_y_host   # numpy mirror of the labels (host-side bookkeeping without a device sync)
            start += sz
            yield bx, by


def setup_test_loader(test_data, params):
    """continuum/data_utils.py:57-64: one shuffled loader per task (test_batch, no drop_last)."""
    return [DeviceLoader(x_test, y_test, params.test_batch, shuffle=True, drop_last=False) for (x_test, y_test) in test_data]
