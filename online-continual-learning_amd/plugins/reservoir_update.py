"""utils/buffer/reservoir_update.py:8-61.  Same fill-then-reservoir logic and the same RNG call
(FloatTensor(n).uniform_(0, n_seen).long(), drawn on the CPU generator as the CPU reference does); the slot
overwrite is a HIP scatter."""
import numpy as np
import torch

from .. import ops
from .. import debug
from .buffer_utils import _host_labels


class Reservoir_update(object):
    def __init__(self, params):
        super().__init__()

    def update(self, buffer, x, y, **kwargs):
        batch_size = x.size(0)
        y_host = _host_labels(y, kwargs.get("y_host"))

        # add whatever still fits in the buffer
        place_left = max(0, buffer.buffer_img.size(0) - buffer.current_index)
        if place_left:
            offset = min(place_left, batch_size)
            buffer.buffer_img[buffer.current_index: buffer.current_index + offset].data.copy_(x[:offset])
            buffer.buffer_label[buffer.current_index: buffer.current_index + offset].data.copy_(y[:offset])
            buffer.label_host[buffer.current_index: buffer.current_index + offset] = y_host[:offset]

            buffer.current_index += offset
            buffer.n_seen_so_far += offset

            # everything was added
            if offset == x.size(0):
                filled_idx = list(range(buffer.current_index - offset, buffer.current_index, ))
                debug.emit("reservoir", slots=list(filled_idx))
                return filled_idx

        # remove what is already in the buffer
        x, y = x[place_left:], y[place_left:]
        y_host = y_host[place_left:]

        indices = torch.FloatTensor(x.size(0)).uniform_(0, buffer.n_seen_so_far).long()
        valid_indices = (indices < buffer.buffer_img.size(0)).long()

        idx_new_data = valid_indices.nonzero().squeeze(-1)
        idx_buffer = indices[idx_new_data]

        buffer.n_seen_so_far += x.size(0)

        if idx_buffer.numel() == 0:
            debug.emit("reservoir", slots=[])
            return []

        assert idx_buffer.max() < buffer.buffer_img.size(0)
        assert idx_buffer.max() < buffer.buffer_label.size(0)

        assert idx_new_data.max() < x.size(0)
        assert idx_new_data.max() < y.size(0)

        idx_map = {idx_buffer[i].item(): idx_new_data[i].item() for i in range(idx_buffer.size(0))}

        keys = list(idx_map.keys())
        vals = list(idx_map.values())
        dev = buffer.buffer_img.device
        keys_dev = ops.upload(torch.tensor(keys, dtype=torch.long), dev)
        vals_dev = ops.upload(torch.tensor(vals, dtype=torch.long), dev)
        # perform overwrite op
        ops.scatter_rows(buffer.buffer_img, keys_dev, ops.gather_rows(x.contiguous(), vals_dev))
        ops.scatter_rows(buffer.buffer_label, keys_dev, ops.gather_rows(y.contiguous(), vals_dev))
        buffer.label_host[np.asarray(keys, dtype=np.int64)] = y_host[np.asarray(vals, dtype=np.int64)]
        debug.emit("reservoir", slots=list(keys))
        return keys
