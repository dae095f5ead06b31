"""Shape tables and factories with the reference's names (utils/setup_elements.py:11-82)."""
import torch
import torch.nn as nn

from .resnet import Reduced_ResNet18, SupConResNet
from .optim import FusedSGD

default_trick = {'labels_trick': False, 'kd_trick': False, 'separated_softmax': False,
                 'review_trick': False, 'ncm_trick': False, 'kd_trick_star': False}

input_size_match = {
    'cifar100': [3, 32, 32],
    'cifar10': [3, 32, 32],
    'core50': [3, 128, 128],
    'mini_imagenet': [3, 84, 84],
    'openloris': [3, 50, 50]
}

n_classes = {
    'cifar100': 100,
    'cifar10': 10,
    'core50': 50,
    'mini_imagenet': 100,
    'openloris': 69
}


def setup_architecture(params):
    """utils/setup_elements.py:46-68.  Construction order (and therefore torch-RNG consumption and the initial
    weights for a given seed) is identical to the reference."""
    nclass = n_classes[params.data]
    hw = tuple(input_size_match[params.data][1:])
    if params.agent in ['SCR', 'SCP']:
        if params.data == 'mini_imagenet':
            return SupConResNet(640, head=params.head, in_hw=hw)
        return SupConResNet(head=params.head, in_hw=hw)
    if params.data in ('cifar100', 'cifar10'):
        return Reduced_ResNet18(nclass, in_hw=hw)
    if params.data == 'mini_imagenet':
        model = Reduced_ResNet18(nclass, in_hw=hw)
        model.linear = nn.Linear(640, nclass, bias=True)
        return model
    raise NotImplementedError("dataset %r is outside the BASELINE configs of the HIP hot path" % (params.data,))


def setup_opt(optimizer, model, lr, wd):
    """utils/setup_elements.py:71-82.  'SGD' returns the fused single-kernel optimiser (same update rule as
    torch.optim.SGD with momentum 0)."""
    if optimizer == 'SGD':
        return FusedSGD(model, lr=lr, weight_decay=wd)
    elif optimizer == 'Adam':
        return torch.optim.Adam(model.parameters(), lr=lr, weight_decay=wd)
    else:
        raise Exception('wrong optimizer name')
