"""Random candidates plus their label-matched partners -- the `retrieve_methods['mem_match']` plugin (reference:
utils/buffer/mem_match.py:5-21).  Returns FOUR tensors (candidates and matches), as the reference does; the ER / SCR loops of the
reference unpack two, so this plugin serves agents that ask for the pair explicitly."""
import torch

from .buffer_utils import match_retrieve, random_retrieve


class MemMatch_retrieve(object):
    def __init__(self, params):
        self.num_retrieve = params.eps_mem_batch
        self.warmup = params.warmup

    def retrieve(self, buffer, **kwargs):
        empty = torch.tensor([])
        cand_x, cand_y, match_x, match_y = empty, empty, empty, empty
        if buffer.n_seen_so_far > self.num_retrieve * self.warmup:
            while match_x.size(0) == 0:   # redraw candidates until every one of them has a partner outside the draw
                cand_x, cand_y, drawn = random_retrieve(buffer, self.num_retrieve, return_indices=True)
                if cand_x.size(0) == 0:
                    break
                match_x, match_y = match_retrieve(buffer, cand_y, drawn)
        return cand_x, cand_y, match_x, match_y
