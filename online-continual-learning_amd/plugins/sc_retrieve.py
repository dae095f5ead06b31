"""Label-matched retrieval -- the `retrieve_methods['match']` plugin (reference: utils/buffer/sc_retrieve.py:4-15): once more than
eps_mem_batch * warmup samples have been seen, one buffered sample of the same class for every item of the incoming batch
(plugins/buffer_utils.match_retrieve); before that, nothing."""
import torch

from .buffer_utils import match_retrieve


class Match_retrieve(object):
    def __init__(self, params):
        self.num_retrieve = params.eps_mem_batch
        self.warmup = params.warmup

    def retrieve(self, buffer, **kwargs):
        if buffer.n_seen_so_far <= self.num_retrieve * self.warmup:
            return torch.tensor([]), torch.tensor([])
        return match_retrieve(buffer, kwargs['y'])
