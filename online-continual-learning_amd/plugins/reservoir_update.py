"""Reservoir sampling into the replay memory -- the `update_methods['random']` plugin (reference:
utils/buffer/reservoir_update.py:8-61).

Observable behaviour kept: while slots are free the stream fills them in order; afterwards every remaining item draws
j ~ U[0, n_seen_so_far) with ONE `FloatTensor(k).uniform_(0, n_seen).long()` call on the torch CPU generator (what the CPU
reference draws), is kept iff j < mem_size, and of several items drawing the same slot the last one wins while the slot keeps its
first position in the returned list (the reference's dict).  Counters, the returned slot list and the host label mirror are
maintained on the host; the overwrite itself is one gather + one scatter kernel per tensor."""
import numpy as np
import torch

from .. import debug
from .. import ops
from .buffer_utils import _host_labels


class Reservoir_update:
    def __init__(self, params):
        pass

    @staticmethod
    def _fill(buffer, x, y, y_host, count):
        """Append the first `count` items at current_index."""
        lo = buffer.current_index
        hi = lo + count
        buffer.buffer_img[lo:hi].copy_(x[:count])
        buffer.buffer_label[lo:hi].copy_(y[:count])
        buffer.label_host[lo:hi] = y_host[:count]
        buffer.current_index = hi
        buffer.n_seen_so_far += count
        return list(range(lo, hi))

    def update(self, buffer, x, y, **kwargs):
        capacity = buffer.buffer_img.size(0)
        y_host = _host_labels(y, kwargs.get("y_host"))
        n_items = x.size(0)

        free = max(0, capacity - buffer.current_index)
        if free:
            taken = min(free, n_items)
            slots = self._fill(buffer, x, y, y_host, taken)
            if taken == n_items:
                # the reference refreshes the tracker only when the whole batch fitted (:22-27; its TODO at :30 notes the gap)
                if getattr(buffer.params, "buffer_tracker", False):
                    buffer.buffer_tracker.update_cache(buffer.label_host, y_host[:taken], slots)
                debug.emit("reservoir", slots=list(slots))
                return slots
        # the part of the batch that did not fit (all of it once the memory is full)
        x, y, y_host = x[free:], y[free:], y_host[free:]

        draws = torch.FloatTensor(x.size(0)).uniform_(0, buffer.n_seen_so_far).long()
        buffer.n_seen_so_far += x.size(0)
        kept = torch.nonzero(draws < capacity).squeeze(-1)
        if kept.numel() == 0:
            debug.emit("reservoir", slots=[])
            return []

        winner = {}                      # slot -> item; a later item replaces an earlier one, the slot keeps its place
        for item, slot in zip(kept.tolist(), draws[kept].tolist()):
            assert 0 <= slot < capacity and 0 <= item < x.size(0)
            winner[slot] = item
        slots, items = list(winner.keys()), list(winner.values())

        if getattr(buffer.params, "buffer_tracker", False):   # before the overwrite: the tracker reads the labels being replaced (:55-57)
            buffer.buffer_tracker.update_cache(buffer.label_host, y_host[np.asarray(items, dtype=np.int64)], slots)
        dev = buffer.buffer_img.device
        slots_dev = ops.upload(torch.tensor(slots, dtype=torch.long), dev)
        items_dev = ops.upload(torch.tensor(items, dtype=torch.long), dev)
        ops.scatter_rows(buffer.buffer_img, slots_dev, ops.gather_rows(x.contiguous(), items_dev))
        ops.scatter_rows(buffer.buffer_label, slots_dev, ops.gather_rows(y.contiguous(), items_dev))
        buffer.label_host[np.asarray(slots, dtype=np.int64)] = y_host[np.asarray(items, dtype=np.int64)]
        debug.emit("reservoir", slots=list(slots))
        return slots
