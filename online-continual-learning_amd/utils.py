"""Host helpers with the reference's names (utils/utils.py:4-16,25-42,93-107)."""
import torch


def maybe_cuda(what, use_cuda=True, **kw):
    """utils/utils.py:4-16. Moves `what` to the GPU when one is visible (and use_cuda is not False)."""
    if getattr(what, "is_cuda", False):      # already resident (the common case inside the replay loop)
        return what
    if use_cuda is not False and torch.cuda.is_available():
        what = what.cuda()
    return what


def boolean_string(s):
    if s not in {'False', 'True'}:
        raise ValueError('Not a valid boolean string')
    return s == 'True'


class AverageMeter(object):
    """utils/utils.py:25-42. Values may be device tensors; nothing is synchronised until avg() is read."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.sum = 0
        self.count = 0

    def update(self, val, n):
        if torch.is_tensor(val):
            val = val.detach()
        self.sum += val * n
        self.count += n

    def avg(self):
        if self.count == 0:
            return 0
        return float(self.sum) / self.count


def nonzero_indices(bool_mask_tensor):
    """utils/utils.py:105-107."""
    return bool_mask_tensor.nonzero(as_tuple=True)[0]


def mini_batch_deep_features(model, total_x, num):
    """utils/utils.py:45-90: eval-mode, no-grad features.  The reference chunks by 64; eval-mode BatchNorm is
    per-sample so the chunk size does not change results and the engine takes the largest chunk it can."""
    is_train = False
    if model.training:
        is_train = True
        model.eval()
    with torch.no_grad():
        # total_x: a tensor, or the pieces the reference concatenates (read where they are: ocl_net_forward_segments)
        src = total_x if isinstance(total_x, (list, tuple)) else total_x[:num]
        feats = model.features_batched(src).reshape((num, -1))
    if is_train:
        model.train()
    return feats
