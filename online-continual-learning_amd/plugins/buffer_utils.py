"""utils/buffer/buffer_utils.py: random_retrieve (:9-26), ClassBalancedRandomSampling (:74-160), get_grad_vector
(:58-71) — host bookkeeping kept in Python (numpy global RNG, torch CPU RNG, CPython set/dict order are part of
the reference's observable behaviour), data movement done by the HIP gather kernels."""
from collections import defaultdict

import numpy as np
import torch

from .. import ops
from .. import debug


def _host_labels(y, y_host=None):
    if y_host is not None:
        return np.asarray(y_host).astype(np.int64)
    return y.detach().cpu().numpy().astype(np.int64)


def random_retrieve(buffer, num_retrieve, excl_indices=None, return_indices=False):
    """buffer_utils.py:9-26: uniform sample without replacement of the filled slots (numpy global RNG)."""
    filled_indices = np.arange(buffer.current_index)
    if excl_indices is not None:
        excl_indices = list(excl_indices)
    else:
        excl_indices = []
    valid_indices = np.setdiff1d(filled_indices, np.array(excl_indices))
    num_retrieve = min(num_retrieve, valid_indices.shape[0])
    indices = torch.from_numpy(np.random.choice(valid_indices, num_retrieve, replace=False)).long()

    debug.emit("random_retrieve", indices=indices.numpy().copy())
    idx_dev = ops.upload(indices, buffer.buffer_img.device)
    x = ops.gather_rows(buffer.buffer_img, idx_dev)
    y = ops.gather_rows(buffer.buffer_label, idx_dev)
    y.host = buffer.label_host[indices.numpy()] if num_retrieve else np.zeros(0, dtype=np.int64)

    if return_indices:
        return x, y, indices
    else:
        return x, y


def match_retrieve(buffer, cur_y, exclud_idx=None):
    """buffer_utils.py:29-49: for every item of the batch one buffered sample of the same class, drawn per class with Python's
    global `random.sample` from the tracker's slot set (CPython set order is part of the behaviour); two empty tensors when some
    class of the batch has too few slots.  Labels come from the numpy mirror the loader attaches (`cur_y.host`)."""
    import random
    from collections import Counter
    ys = _host_labels(cur_y, getattr(cur_y, "host", None)).tolist()
    per_class = Counter(ys)
    positions = defaultdict(list)
    for pos, label in enumerate(ys):
        positions[label].append(pos)
    chosen = [None] * len(ys)
    for label in per_class:
        slots = buffer.buffer_tracker.class_index_cache[label]
        if exclud_idx is not None:
            slots = slots - set(exclud_idx.tolist())
        if not slots or len(slots) < per_class[label]:
            print('match retrieve attempt fail')
            return torch.tensor([]), torch.tensor([])
        for pos, slot in zip(positions[label], random.sample(list(slots), per_class[label])):
            chosen[pos] = slot
    indices = torch.tensor(chosen)
    debug.emit("match_retrieve", indices=indices.numpy().copy())
    idx_dev = ops.upload(indices, buffer.buffer_img.device)
    x = ops.gather_rows(buffer.buffer_img, idx_dev)
    y = ops.gather_rows(buffer.buffer_label, idx_dev)
    y.host = buffer.label_host[indices.numpy()]
    return x, y


def get_grad_vector(model):
    """buffer_utils.py:58-71: the flat gradient vector (zeros where a parameter has no gradient).  The engine's
    flat gradient array already has that layout; before any backward it is logically zero."""
    g = model.flat_grads()
    if model._grads_fresh:
        return torch.zeros_like(g)
    return g


class ClassBalancedRandomSampling:
    """buffer_utils.py:74-160.  Class-level caches, reset by the ASER plugins' constructors exactly as in the
    reference.  All index bookkeeping and RNG (torch CPU generator: one randperm per non-empty class) stay on the
    host; only the selected rows are gathered on the GPU."""
    class_index_cache = None
    class_num_cache = None

    @classmethod
    def sample(cls, buffer_x, buffer_y, n_smp_cls, excl_indices=None, device="cpu", label_host=None):
        if excl_indices is None:
            excl_indices = set()

        picks = []

        # Use cache to retrieve indices belonging to each class in buffer.  Same operations in the same order as the
        # reference (set difference -> CPython iteration order of the NEW set; one torch.randperm per non-empty class
        # on the CPU generator), but the selected indices are collected in a Python list and converted once: the
        # reference's per-class torch.tensor(list(...))[perm][:n] + torch.cat cost ~1 ms per call at 100 classes.
        randperm = torch.randperm
        for ind_set in cls.class_index_cache.values():
            if ind_set:
                # Exclude some indices
                valid_ind = ind_set - excl_indices
                # Auxiliary indices for permutation
                perm_ind = randperm(len(valid_ind))
                # Apply permutation, and select indices
                order = list(valid_ind)
                for j in perm_ind[:n_smp_cls].tolist():
                    picks.append(order[j])
        sample_ind = torch.tensor(picks, dtype=torch.long)

        idx_dev = ops.upload(sample_ind, buffer_x.device)
        x = ops.gather_rows(buffer_x, idx_dev)
        y = ops.gather_rows(buffer_y, idx_dev)
        if label_host is not None:
            y.host = label_host[sample_ind.numpy()]
        return x, y, sample_ind

    @classmethod
    def update_cache(cls, buffer_y_host, num_class, new_y=None, ind=None, device="cpu"):
        """buffer_y_host: numpy mirror of buffer_label; new_y / ind: host integer sequences."""
        if cls.class_index_cache is None:
            # Initialize caches
            cls.class_index_cache = defaultdict(set)
            cls.class_num_cache = torch.zeros(num_class, dtype=torch.long)

        if new_y is not None:
            # If ASER update is being used, keep updating existing caches
            ind = [int(i) for i in ind]
            new_y = [int(v) for v in new_y]
            orig_y = [int(buffer_y_host[i]) for i in ind]
            for i_int, ny_int, oy_int in zip(ind, new_y, orig_y):
                # Update dictionary according to new class label of index i
                if oy_int in cls.class_index_cache and i_int in cls.class_index_cache[oy_int]:
                    cls.class_index_cache[oy_int].remove(i_int)
                    cls.class_num_cache[oy_int] -= 1
                cls.class_index_cache[ny_int].add(i_int)
                cls.class_num_cache[ny_int] += 1
        else:
            # If only ASER retrieve is being used, reset cache and update it based on buffer
            cls_ind_cache = defaultdict(set)
            for i, c in enumerate(buffer_y_host):
                cls_ind_cache[int(c)].add(i)
            cls.class_index_cache = cls_ind_cache


class BufferClassTracker(object):
    """buffer_utils.py:163-203: per-buffer class -> set-of-slots index and per-class counts, maintained by the reservoir update
    (utils/buffer/reservoir_update.py:25-26,56-57) and read by match_retrieve.  `buffer_y` is the numpy label mirror."""

    def __init__(self, num_class, device="cpu"):
        self.class_index_cache = defaultdict(set)
        self.class_num_cache = np.zeros(num_class)

    def update_cache(self, buffer_y, new_y=None, ind=None):
        for slot, new_label in zip(ind, new_y):
            slot, new_label, old_label = int(slot), int(new_label), int(buffer_y[slot])
            if old_label in self.class_index_cache and slot in self.class_index_cache[old_label]:
                self.class_index_cache[old_label].remove(slot)
                self.class_num_cache[old_label] -= 1
            self.class_index_cache[new_label].add(slot)
            self.class_num_cache[new_label] += 1

    def check_tracker(self):
        print(self.class_num_cache.sum())
        print(len([k for i in self.class_index_cache.values() for k in i]))
