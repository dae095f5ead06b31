"""utils/kd_manager.py:14-28 -- the teacher of the KD tricks.  The reference deep-copies the model at the end of every task
(agents/base.py:90-91) and forwards the copy under no_grad; here the teacher is a snapshot of the flat parameter array and
its forward is the engine's forward over that array (the mechanism of MIR's virtual model): same mode as the student (the
copy is taken, and stays, in train mode: batch statistics), and the student's BatchNorm running statistics are not touched
(the reference updates the copy's)."""
import torch

from .loss import loss_fn_kd


class KdManager:
    def __init__(self):
        self.teacher_model = None      # flat parameter snapshot (torch tensor on the device)
        self._owner = None

    def update_teacher(self, model):
        self._owner = model
        self.teacher_model = model.flat_params().detach().clone()

    def get_kd_loss(self, cur_model_logits, x):
        if self.teacher_model is not None:
            with torch.no_grad():
                prev_model_logits = self._owner.forward_with_params(x, self.teacher_model)
            dist_loss = loss_fn_kd(cur_model_logits, prev_model_logits)
        else:
            dist_loss = 0
        return dist_loss
