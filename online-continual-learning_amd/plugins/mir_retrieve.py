"""utils/buffer/mir_retrieve.py:8-65 — Maximally Interfered Retrieval.

The reference deep-copies the model, overwrites its grads and applies theta - lr*grad (:34-47), then runs both
models on the 50 candidates in train mode under no_grad (:23-25).  Here the virtual parameters are ONE fused
kernel writing a shadow flat array (no deepcopy) and the second forward reads that array through the engine's
`params_override`; the per-sample CE difference and the top-k are HIP kernels."""
import torch

from .. import ops
from .. import debug
from .buffer_utils import random_retrieve, get_grad_vector


class MIR_retrieve(object):
    def __init__(self, params, **kwargs):
        super().__init__()
        self.params = params
        self.subsample = params.subsample
        self.num_retrieve = params.eps_mem_batch
        self._shadow = None

    def retrieve(self, buffer, **kwargs):
        sub_x, sub_y = random_retrieve(buffer, self.subsample)
        model = buffer.model
        grad_vector = get_grad_vector(model)
        flat = model.flat_params()
        if self._shadow is None or self._shadow.shape != flat.shape:
            self._shadow = torch.empty_like(flat)
        # theta' = theta - lr * grad   (get_future_step_parameters, :34-47)
        ops.sgd_step(flat, grad_vector, self.params.learning_rate, 0.0, 1.0, out=self._shadow)
        if sub_x.size(0) > 0:
            with torch.no_grad():
                logits_pre = model.forward(sub_x)
                logits_post = model.forward_with_params(sub_x, self._shadow)
                scores = ops.mir_scores(logits_pre, logits_post, sub_y)
                big_ind = ops.argsort_desc(scores)[:self.num_retrieve].contiguous()
                if debug.on():
                    debug.emit("mir", scores=scores.cpu().numpy(), big_ind=big_ind.cpu().numpy())
            return ops.gather_rows(sub_x, big_ind), ops.gather_rows(sub_y, big_ind)
        else:
            return sub_x, sub_y
