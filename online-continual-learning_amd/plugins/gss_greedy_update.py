"""GSS-Greedy buffer update -- the `update_methods['GSS']` plugin (reference: utils/buffer/gss_greedy_update.py:6-122).

A sample's score is the largest cosine similarity between its loss gradient and the gradients of a few random mini-batches of
the memory; while the memory fills every incoming sample is stored with its score, afterwards a batch whose own gradient points
away from all memory gradients (max cosine < 0) competes, sample by sample and at random, against memory slots drawn in
proportion to their scores.

All gradients are those of an EVAL-mode forward (the plugin switches the model to eval() first): the engine records such a pass
with OCL_FWD_FROZEN_BN and back-propagates through the running-statistics BatchNorm.  Gradient vectors never leave the GPU: the
engine's flat gradient array is the get_grad_vector layout, `ocl_cosine_max` reduces a [k, n_params] stack against it.  What the
reference draws on the CPU generator is drawn there, so the decisions need the scores on the host: one small device->host copy per
similarity batch, as in the reference's `.cpu()` / `if batch_sim < 0`.

Generator parity is with the reference running on the CPU DEVICE (params.cuda = False, the configuration the oracle and the golden
run `er_gss` pin): there randperm and both multinomials consume the CPU generator.  In a CUDA run of the reference only the first
multinomial does (`buffer_score.cpu()`, gss_greedy_update.py:28-30); the second (`outcome`, :41) is drawn from CUDA tensors, i.e. from
the CUDA generator, whose stream this path neither has nor imitates -- here it consumes CPU-generator draws instead, so every later
`torch.randperm` / `uniform_` of a run is shifted relative to a CUDA reference run."""
import torch

from .. import debug
from .. import ops
from ..loss import cross_entropy_mean
from .buffer_utils import _host_labels


class GSSGreedyUpdate(object):
    def __init__(self, params):
        self.mem_strength = params.gss_mem_strength      # gradient vectors compared against (alg. 2, line 5)
        self.gss_batch_size = params.gss_batch_size
        self.buffer_score = torch.zeros(params.mem_size, dtype=torch.float32)   # host: every consumer is a host-side draw

    # ---- gradients ---------------------------------------------------------------------------------------------------------
    @staticmethod
    def _grad_into(model, x, y, out_row):
        """flat gradient of CE(model(x), y) in eval mode -> out_row (a [n_params] device view)."""
        model.mark_grads_zero()
        cross_entropy_mean(model.forward(x), y).backward()
        out_row.copy_(model.flat_grads())

    def _memory_grads(self, buffer):
        """get_rand_mem_grads (:82-104): gradients of up to mem_strength disjoint random mini-batches of the memory."""
        model = buffer.model
        bs = min(self.gss_batch_size, buffer.current_index)
        n_sub = min(self.mem_strength, buffer.current_index // bs)
        order = torch.randperm(buffer.current_index)
        order_dev = ops.upload(order, buffer.buffer_img.device)
        stack = torch.empty((n_sub, model.flat_params().numel()), dtype=torch.float32, device=buffer.buffer_img.device)
        for i in range(n_sub):
            pick = order_dev[i * bs:i * bs + bs].contiguous()
            self._grad_into(model, ops.gather_rows(buffer.buffer_img, pick), ops.gather_rows(buffer.buffer_label, pick), stack[i])
        return stack

    def _sample_scores(self, buffer, mem_grads, x, y):
        """get_each_batch_sample_sim (:106-122): per sample, the best cosine similarity with the memory gradients."""
        model = buffer.model
        scores = torch.empty(x.size(0), dtype=torch.float32, device=x.device)
        grad = torch.empty_like(model.flat_params())
        for i in range(x.size(0)):
            self._grad_into(model, x[i:i + 1].contiguous(), y[i:i + 1].contiguous(), grad)
            ops.cosine_max(mem_grads, grad, out=scores[i:i + 1])
        return scores.cpu()

    def _batch_score(self, buffer, x, y):
        """get_batch_sim (:66-80)."""
        mem_grads = self._memory_grads(buffer)
        grad = torch.empty_like(buffer.model.flat_params())
        self._grad_into(buffer.model, x, y, grad)
        return float(ops.cosine_max(mem_grads, grad).cpu()), mem_grads

    # ---- the plugin entry point -------------------------------------------------------------------------------------------------
    def update(self, buffer, x, y, **kwargs):
        model = buffer.model
        y_host = _host_labels(y, kwargs.get("y_host"))
        x, y = x.contiguous(), y.contiguous()
        model.eval()
        room = buffer.buffer_img.size(0) - buffer.current_index
        if room <= 0:
            batch_sim, mem_grads = self._batch_score(buffer, x, y)
            info = dict(batch_sim=batch_sim)
            if batch_sim < 0:
                held = self.buffer_score[:buffer.current_index]
                weights = (held - torch.min(held)) / ((torch.max(held) - torch.min(held)) + 0.01)
                slots = torch.multinomial(weights, x.size(0), replacement=False)          # candidates for replacement
                item_sim = self._sample_scores(buffer, mem_grads, x, y)
                odds = torch.cat((((item_sim + 1) / 2).unsqueeze(1), ((self.buffer_score[slots] + 1) / 2).unsqueeze(1)), dim=1)
                swap = torch.multinomial(odds, 1, replacement=False).squeeze(1).bool()     # 1: the newcomer takes the slot
                won = torch.arange(end=item_sim.size(0))[swap]
                if won.numel():
                    slots_dev = ops.upload(slots[swap], x.device)
                    won_dev = ops.upload(won, x.device)
                    ops.scatter_rows(buffer.buffer_img, slots_dev, ops.gather_rows(x, won_dev))
                    ops.scatter_rows(buffer.buffer_label, slots_dev, ops.gather_rows(y, won_dev))
                    buffer.label_host[slots[swap].numpy()] = y_host[won.numpy()]
                    self.buffer_score[slots[swap]] = item_sim[won].clone()
                info.update(index=slots.numpy().copy(), item_sim=item_sim.numpy().copy(), sub=swap.numpy().copy())
            debug.emit("gss", **info)
        else:
            take = min(room, x.size(0))
            x, y, y_host = x[:take], y[:take], y_host[:take]
            if buffer.current_index == 0:
                item_sim = torch.zeros(take) + 0.1                                         # first insertion (:50-51)
            else:
                item_sim = self._sample_scores(buffer, self._memory_grads(buffer), x, y)
            lo = buffer.current_index
            buffer.buffer_img[lo:lo + take].copy_(x)
            buffer.buffer_label[lo:lo + take].copy_(y)
            buffer.label_host[lo:lo + take] = y_host
            self.buffer_score[lo:lo + take] = item_sim
            buffer.current_index += take
            debug.emit("gss", fill=take, item_sim=item_sim.numpy().copy())
        model.train()
