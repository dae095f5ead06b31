"""torch.optim.SGD (momentum 0) as one HIP kernel over the flat parameter array (K8).

Replaces `torch.optim.SGD.step` (utils/setup_elements.py:73-75; call sites agents/exp_replay.py:87,89,
agents/scr.py:60).  zero_grad() costs nothing: it only tells the engine that the next backward overwrites."""
import torch

from . import ops


class FusedSGD(torch.optim.Optimizer):
    def __init__(self, model, lr, weight_decay=0.0):
        if not hasattr(model, "flat_params"):
            raise RuntimeError("FusedSGD needs an engine-backed model (ocl_amd.resnet)")
        self.model = model
        super().__init__(list(model.parameters()), dict(lr=lr, weight_decay=weight_decay))

    def zero_grad(self, set_to_none=True):
        self.model.mark_grads_zero()

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        g = self.param_groups[0]
        m = self.model
        if m._grads_fresh:
            return None  # no backward since zero_grad(): torch.optim.SGD skips parameters whose grad is None
        ops.sgd_step(m.flat_params(), m.flat_grads(), g["lr"], g["weight_decay"], grad_scale)
        return None
