"""Uniform retrieval from the filled part of the replay memory -- the `retrieve_methods['random']` plugin
(reference: utils/buffer/random_retrieve.py:3-9; the draw itself is plugins/buffer_utils.random_retrieve)."""
from . import buffer_utils


class Random_retrieve:
    """`params.eps_mem_batch` rows per call, fewer while the memory holds fewer; x / y keyword arguments are accepted and ignored."""

    def __init__(self, params):
        self.num_retrieve = int(params.eps_mem_batch)

    def retrieve(self, buffer, **_unused):
        rows, labels = buffer_utils.random_retrieve(buffer, self.num_retrieve)
        return rows, labels
