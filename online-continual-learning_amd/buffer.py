"""Device-resident replay memory behind the reference's `Buffer` surface (utils/buffer/buffer.py:8-41).

Attributes the plugins rely on: `buffer_img` float32 [mem, C, H, W] and `buffer_label` int64 [mem] (registered module buffers in
HBM, so `torch.save` keeps working), the fill / stream counters `current_index` and `n_seen_so_far`, `model`, `params`, `cuda`,
`device`.  `label_host` is a numpy mirror of `buffer_label` that the update plugins keep in step, so class-balanced sampling and
the class caches never synchronise the device; `buffer_tracker` (params.buffer_tracker) is the class -> slots index the match
retrievals read.  update() / retrieve() dispatch to the plugins named by `params.update` /
`params.retrieve` in `name_match`."""
import numpy as np
import torch

from . import name_match
from .setup_elements import input_size_match
from .utils import maybe_cuda


class Buffer(torch.nn.Module):
    def __init__(self, model, params):
        super().__init__()
        self.model, self.params = model, params
        self.cuda = params.cuda
        self.device = "cuda" if params.cuda else "cpu"
        self.current_index = self.n_seen_so_far = 0

        slots = params.mem_size
        print('buffer has %d slots' % slots)
        images = maybe_cuda(torch.zeros((slots,) + tuple(input_size_match[params.data]), dtype=torch.float32))
        if not images.is_cuda:
            raise RuntimeError("the replay buffer must live on the MI355X; no GPU is visible and there is no CPU path")
        self.register_buffer('buffer_img', images)
        self.register_buffer('buffer_label', maybe_cuda(torch.zeros(slots, dtype=torch.int64)))
        self.label_host = np.zeros(slots, dtype=np.int64)

        self.update_method = name_match.update_methods[params.update](params)
        self.retrieve_method = name_match.retrieve_methods[params.retrieve](params)

        if getattr(params, "buffer_tracker", False):   # utils/buffer/buffer.py:33-34
            from .plugins.buffer_utils import BufferClassTracker
            from .setup_elements import n_classes
            self.buffer_tracker = BufferClassTracker(n_classes[params.data], self.device)

    def sync_host_labels(self):
        """Re-read the label mirror after something other than the update plugins wrote buffer_label."""
        self.label_host = self.buffer_label.detach().cpu().numpy().copy()

    def retrieve(self, **kwargs):
        return self.retrieve_method.retrieve(buffer=self, **kwargs)

    def update(self, x, y, **kwargs):
        return self.update_method.update(buffer=self, x=x, y=y, **kwargs)
