"""Replay memory with the reference's attributes and plugin dispatch (utils/buffer/buffer.py:8-41), device-resident.

`buffer_img` [mem,C,H,W] float32 and `buffer_label` [mem] int64 are registered module buffers on the MI355X;
`label_host` is a numpy mirror of `buffer_label` that the update plugins keep in step so that class-balanced
sampling and cache bookkeeping never synchronise the device."""
import numpy as np
import torch

from . import name_match
from .setup_elements import input_size_match
from .utils import maybe_cuda


class Buffer(torch.nn.Module):
    def __init__(self, model, params):
        super().__init__()
        self.params = params
        self.model = model
        self.cuda = self.params.cuda
        self.current_index = 0
        self.n_seen_so_far = 0
        self.device = "cuda" if self.params.cuda else "cpu"

        # define buffer
        buffer_size = params.mem_size
        print('buffer has %d slots' % buffer_size)
        input_size = input_size_match[params.data]
        buffer_img = maybe_cuda(torch.FloatTensor(buffer_size, *input_size).fill_(0))
        buffer_label = maybe_cuda(torch.LongTensor(buffer_size).fill_(0))
        if not buffer_img.is_cuda:
            raise RuntimeError("the replay buffer must live on the MI355X; no GPU is visible and there is no CPU path")

        # registering as buffer allows us to save the object using `torch.save`
        self.register_buffer('buffer_img', buffer_img)
        self.register_buffer('buffer_label', buffer_label)
        self.label_host = np.zeros(buffer_size, dtype=np.int64)

        # define update and retrieve method
        self.update_method = name_match.update_methods[params.update](params)
        self.retrieve_method = name_match.retrieve_methods[params.retrieve](params)

        if getattr(self.params, "buffer_tracker", False):
            raise NotImplementedError("buffer_tracker (match / mem_match retrieval) is outside the HIP hot path")

    def sync_host_labels(self):
        """Re-read the label mirror after something other than the update plugins wrote buffer_label."""
        self.label_host = self.buffer_label.detach().cpu().numpy().copy()

    def update(self, x, y, **kwargs):
        return self.update_method.update(buffer=self, x=x, y=y, **kwargs)

    def retrieve(self, **kwargs):
        return self.retrieve_method.retrieve(buffer=self, **kwargs)
