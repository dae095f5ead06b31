"""ASER memory update -- the `update_methods['ASER']` plugin (reference: utils/buffer/aser_update.py:8-112).

While the memory has free slots the stream fills it like a reservoir (and the class cache learns every new slot).  Once it is full,
each incoming batch competes for slots: an evaluation set (class-balanced memory samples + the batch's minority-class items) scores
a candidate set (random memory samples + the whole batch) with kNN Shapley values; the candidates are ranked by total value, the
best `n_memory_candidates` keep / take a slot, and every batch item ranked among them replaces one of the memory candidates that fell
out.  Host side: the RNG draws and the class-cache bookkeeping (same calls, same order as the reference); device side: feature
extraction, the Shapley kernel, ranking, row moves.  One device->host copy per update (the ranking), which the class cache needs."""
import torch

from .. import debug
from .. import ops
from ..setup_elements import n_classes
from ..utils import maybe_cuda
from .aser_utils import add_minority_class_input, compute_knn_sv
from .buffer_utils import ClassBalancedRandomSampling, _host_labels, random_retrieve
from .reservoir_update import Reservoir_update


class ASER_update(object):
    def __init__(self, params, **kwargs):
        self.device = "cuda" if torch.cuda.is_available() else "cpu"
        self.k = params.k
        self.mem_size = params.mem_size
        self.num_tasks = params.num_tasks
        self.out_dim = n_classes[params.data]
        self.n_smp_cls = int(params.n_smp_cls)
        self.n_total_smp = int(params.n_smp_cls * self.out_dim)
        self.reservoir_update = Reservoir_update(params)
        self._pinned = None
        ClassBalancedRandomSampling.class_index_cache = None     # class-level state, reset per plugin instance (:20)
        # (every mutation of the class sets goes through update_cache: the C helper's memo is verified one draw in 64, buffer_utils.py)
        ClassBalancedRandomSampling.verify_every = int(__import__("os").environ.get("OCL_CBRS_VERIFY_EVERY", "64"))

    # ---- entry point ---------------------------------------------------------------------------------------------------------
    def update(self, buffer, x, y, **kwargs):
        self.update_finish(buffer, self.update_begin(buffer, x, y, **kwargs))

    # Two halves of update(), for a caller that has other GPU work to issue in between (agents/exp_replay.py: the first forward of the
    # next iteration): `update_begin` does everything up to and including the scoring kernels (all RNG draws, in order), `update_finish`
    # waits for the ranking, updates the class table and moves the rows.
    def update_begin(self, buffer, x, y, **kwargs):
        labels = _host_labels(y, kwargs.get("y_host"))
        free = self.mem_size - buffer.current_index
        if free:
            self._append(buffer, x[:free], y[:free], labels[:free])
        if buffer.current_index == self.mem_size:
            return self._score(buffer, x[free:], y[free:], labels[free:])
        return None

    def update_finish(self, buffer, pending):
        if pending is not None:
            self._replace(buffer, *pending)

    def _append(self, buffer, x, y, labels):
        """Fill phase (:27-36): the class cache first (it reads the labels being overwritten), then the reservoir append."""
        first = buffer.current_index
        ClassBalancedRandomSampling.update_cache(buffer.label_host, self.out_dim, new_y=labels, ind=list(range(first, first + x.size(0))),
                                                 device=self.device)
        self.reservoir_update.update(buffer, x, y, y_host=labels)

    # ---- full memory: Shapley-ranked replacement (:43-112) ------------------------------------------------------------------------
    def _score(self, buffer, cur_x, cur_y, cur_labels):
        cur_x, cur_y = maybe_cuda(cur_x).contiguous(), maybe_cuda(cur_y).contiguous()
        # RNG draws in the reference's order: minority threshold (torch CPU), evaluation set (one randperm per class), candidates (numpy)
        minor_x, minor_y = add_minority_class_input(cur_x, cur_y, self.mem_size, self.out_dim, cur_y_host=cur_labels)
        eval_x, eval_y, eval_slots = ClassBalancedRandomSampling.sample(buffer.buffer_img, buffer.buffer_label, self.n_smp_cls, device=self.device)
        mem_x, mem_y, mem_slots = random_retrieve(buffer, self.n_total_smp, set(eval_slots.tolist()), return_indices=True)
        n_mem, n_cur = mem_x.size(0), cur_x.size(0)

        trace = debug.on()
        values = compute_knn_sv(buffer.model, (eval_x, minor_x), torch.cat((eval_y, minor_y)), (mem_x, cur_x),
                                torch.cat((mem_y, cur_y)), self.k, device=self.device, want_order=trace)
        knn_order = None
        if trace:
            values, knn_order = values
        total = ops.col_reduce(values, "sum")
        ranking_dev = ops.argsort_desc(total)
        # the ranking travels to the host asynchronously, behind the scoring kernels and in front of whatever the caller issues next
        if self._pinned is None or self._pinned.numel() < ranking_dev.numel():
            self._pinned = torch.empty(max(1024, ranking_dev.numel()), dtype=ranking_dev.dtype).pin_memory()
        ranking_host = self._pinned[:ranking_dev.numel()]
        ranking_host.copy_(ranking_dev, non_blocking=True)
        arrived = torch.cuda.Event()
        arrived.record()
        buffer.n_seen_so_far += n_cur
        return (cur_x, cur_y, cur_labels, mem_slots, eval_slots, minor_x, total, (ranking_host, arrived, ranking_dev), knn_order, n_mem, n_cur)

    def _replace(self, buffer, cur_x, cur_y, cur_labels, mem_slots, eval_slots, minor_x, total, ranking_dev, knn_order, n_mem, n_cur):
        trace = debug.on()
        ranking_host, arrived, _keep = ranking_dev
        # the update's one synchronisation: the ranking has reached the host (polling the event instead of blocking on it: not faster,
        # profiles/r6_aser_spin_ab.txt)
        arrived.synchronize()
        ranking = ranking_host.clone()

        # the n_mem best-valued candidates hold a slot afterwards: batch items among them move in, memory items outside move out
        keep, drop = ranking[:n_mem], ranking[n_mem:]
        entering = keep[keep >= n_mem] - n_mem            # positions in the batch
        leaving = mem_slots[drop[drop < n_mem]]           # memory slots, paired with `entering` in ranking order
        if trace:
            debug.emit("aser_update", eval_indices=eval_slots.numpy().copy(), cand_ind=mem_slots.numpy().copy(), sv=total.cpu().numpy(),
                       order=ranking.numpy().copy(), ind_buffer=leaving.numpy().copy(), ind_cur=entering.numpy().copy(),
                       n_minority=int(minor_x.size(0)), knn_order=knn_order.cpu().numpy())

        new_labels = cur_labels[entering.numpy()]
        ClassBalancedRandomSampling.update_cache(buffer.label_host, self.out_dim, new_y=new_labels, ind=leaving.tolist(), device=self.device)
        if leaving.numel():
            dev = buffer.buffer_img.device
            dst = ops.upload(leaving, dev)
            new_x, new_y = ops.gather_pair(cur_x, cur_y, entering)
            ops.scatter_rows(buffer.buffer_img, dst, new_x)
            ops.scatter_rows(buffer.buffer_label, dst, new_y)
            buffer.label_host[leaving.numpy()] = new_labels
