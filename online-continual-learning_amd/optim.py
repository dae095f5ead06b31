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
        # parameters that take no part in forward() never get a gradient, and torch.optim.SGD skips a parameter whose grad is None
        # entirely -- weight decay included.  SupConResNet carries one such pair: the encoder's classifier (models/resnet.py:144,157-160)
        self._no_grad_names = [n for n, _ in model.named_parameters() if n.startswith("encoder.linear.")] if hasattr(model, "head_kind") else []

    def zero_grad(self, set_to_none=True):
        self.model.mark_grads_zero()

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        g = self.param_groups[0]
        m = self.model
        if m._grads_fresh:
            return None  # no backward since zero_grad(): torch.optim.SGD skips parameters whose grad is None
        keep = None
        if g["weight_decay"] != 0 and self._no_grad_names:
            named = dict(m.named_parameters())
            keep = [(named[n], named[n].detach().clone()) for n in self._no_grad_names]
        ops.sgd_step(m.flat_params(), m.flat_grads(), g["lr"], g["weight_decay"], grad_scale)
        m.mark_weights_written()
        if keep is not None:      # undo the decay of the gradient-less tensors (two small device copies; weight_decay is 0 in every BASELINE config)
            for p, old in keep:
                p.data.copy_(old)
        return None
