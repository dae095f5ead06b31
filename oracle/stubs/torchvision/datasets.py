"""TEST INFRASTRUCTURE ONLY: placeholders; datasets are never downloaded (no network)."""


class _NoData:
    def __init__(self, *a, **k):
        raise RuntimeError("datasets are not available in this container; use synthetic streams")


CIFAR10 = CIFAR100 = _NoData
