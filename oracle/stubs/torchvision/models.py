"""TEST INFRASTRUCTURE ONLY."""


def resnet18(*a, **k):
    raise RuntimeError("torchvision.models is a stub")
