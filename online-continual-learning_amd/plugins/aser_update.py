"""utils/buffer/aser_update.py:8-112 — ASER update: reservoir fill, then kNN-SV ranked replacement of random
candidates by the current batch."""
import numpy as np
import torch

from .. import ops
from .. import debug
from ..setup_elements import n_classes
from ..utils import maybe_cuda, nonzero_indices
from .aser_utils import compute_knn_sv, add_minority_class_input
from .buffer_utils import ClassBalancedRandomSampling, random_retrieve, _host_labels
from .reservoir_update import Reservoir_update


class ASER_update(object):
    def __init__(self, params, **kwargs):
        super().__init__()
        self.device = "cuda" if torch.cuda.is_available() else "cpu"
        self.k = params.k
        self.mem_size = params.mem_size
        self.num_tasks = params.num_tasks
        self.out_dim = n_classes[params.data]
        self.n_smp_cls = int(params.n_smp_cls)
        self.n_total_smp = int(params.n_smp_cls * self.out_dim)
        self.reservoir_update = Reservoir_update(params)
        ClassBalancedRandomSampling.class_index_cache = None

    def update(self, buffer, x, y, **kwargs):
        model = buffer.model
        y_host = _host_labels(y, kwargs.get("y_host"))

        place_left = self.mem_size - buffer.current_index

        # If buffer is not filled, use available space to store whole or part of batch
        if place_left:
            x_fit = x[:place_left]
            y_fit = y[:place_left]
            y_fit_host = y_host[:place_left]

            ind = list(range(buffer.current_index, buffer.current_index + x_fit.size(0)))
            ClassBalancedRandomSampling.update_cache(buffer.label_host, self.out_dim,
                                                     new_y=y_fit_host, ind=ind, device=self.device)
            self.reservoir_update.update(buffer, x_fit, y_fit, y_host=y_fit_host)

        # If buffer is filled, update buffer by sv
        if buffer.current_index == self.mem_size:
            # remove what is already in the buffer
            cur_x, cur_y = x[place_left:], y[place_left:]
            self._update_by_knn_sv(model, buffer, cur_x, cur_y, y_host[place_left:])

    def _update_by_knn_sv(self, model, buffer, cur_x, cur_y, cur_y_host):
        """aser_update.py:43-112."""
        cur_x = maybe_cuda(cur_x).contiguous()
        cur_y = maybe_cuda(cur_y).contiguous()

        # Find minority class samples from current input batch
        minority_batch_x, minority_batch_y = add_minority_class_input(cur_x, cur_y, self.mem_size, self.out_dim,
                                                                      cur_y_host=cur_y_host)

        # Evaluation set
        eval_x, eval_y, eval_indices = \
            ClassBalancedRandomSampling.sample(buffer.buffer_img, buffer.buffer_label, self.n_smp_cls,
                                               device=self.device)

        # Concatenate minority class samples from current input batch to evaluation set
        eval_x = torch.cat((eval_x, minority_batch_x))
        eval_y = torch.cat((eval_y, minority_batch_y))

        # Candidate set
        cand_excl_indices = set(eval_indices.tolist())
        cand_x, cand_y, cand_ind = random_retrieve(buffer, self.n_total_smp, cand_excl_indices, return_indices=True)

        # Concatenate current input batch to candidate set
        cand_x = torch.cat((cand_x, cur_x))
        cand_y = torch.cat((cand_y, cur_y))

        dbg = debug.on()
        sv_matrix = compute_knn_sv(model, eval_x, eval_y, cand_x, cand_y, self.k, device=self.device, want_order=dbg)
        knn_order = None
        if dbg:
            sv_matrix, knn_order = sv_matrix
        sv = ops.col_reduce(sv_matrix, "sum")

        n_cur = cur_x.size(0)
        n_cand = cand_x.size(0)

        # Number of previously buffered instances in candidate set
        n_cand_buf = n_cand - n_cur

        # the cache / replacement bookkeeping below is host-side Python in the reference as well:
        # this is the step's one device->host synchronisation
        sv_arg_sort = ops.argsort_desc(sv).cpu()

        # Divide SV array into two segments
        # - large: candidate args to be retained; small: candidate args to be discarded
        sv_arg_large = sv_arg_sort[:n_cand_buf]
        sv_arg_small = sv_arg_sort[n_cand_buf:]

        # Extract args relevant to replacement operation
        ind_cur = sv_arg_large[nonzero_indices(sv_arg_large >= n_cand_buf)] - n_cand_buf
        arg_buffer = sv_arg_small[nonzero_indices(sv_arg_small < n_cand_buf)]
        ind_buffer = cand_ind[arg_buffer]

        buffer.n_seen_so_far += n_cur
        if debug.on():
            debug.emit("aser_update", eval_indices=eval_indices.numpy().copy(), cand_ind=cand_ind.numpy().copy(), sv=sv.cpu().numpy(),
                       order=sv_arg_sort.numpy().copy(), ind_buffer=ind_buffer.numpy().copy(), ind_cur=ind_cur.numpy().copy(),
                       n_minority=int(minority_batch_x.size(0)), knn_order=knn_order.cpu().numpy())

        # perform overwrite op
        y_upt_host = cur_y_host[ind_cur.numpy()]
        ClassBalancedRandomSampling.update_cache(buffer.label_host, self.out_dim,
                                                 new_y=y_upt_host, ind=ind_buffer.tolist(), device=self.device)
        if ind_buffer.numel():
            dev = buffer.buffer_img.device
            ind_cur_dev = ops.upload(ind_cur, dev)
            ind_buffer_dev = ops.upload(ind_buffer, dev)
            ops.scatter_rows(buffer.buffer_img, ind_buffer_dev, ops.gather_rows(cur_x, ind_cur_dev))
            ops.scatter_rows(buffer.buffer_label, ind_buffer_dev, ops.gather_rows(cur_y, ind_cur_dev))
            buffer.label_host[ind_buffer.numpy()] = y_upt_host
