"""TEST INFRASTRUCTURE ONLY: placeholders; datasets are never downloaded (no network).  A test may install a synthetic source:
`torchvision.datasets.SYNTHETIC = fn(name, train) -> (uint8 [N,H,W,3], labels)`, which CIFAR10 / CIFAR100 then serve through the
attributes the reference reads (`.data`, `.targets`: continuum/dataset_scripts/cifar10.py:18-24)."""

SYNTHETIC = None


class _Synthetic:
    name = None

    def __init__(self, root=None, train=True, download=False, **kw):
        if SYNTHETIC is None:
            raise RuntimeError("datasets are not available in this container; use synthetic streams")
        self.data, targets = SYNTHETIC(self.name, train)
        self.targets = list(targets)


class CIFAR10(_Synthetic):
    name = "cifar10"


class CIFAR100(_Synthetic):
    name = "cifar100"
