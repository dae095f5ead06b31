"""TEST INFRASTRUCTURE ONLY. Minimal stand-in for the `torchvision` package so that the read-only
reference tree (/root/reference) can be imported in the build container, where torchvision is not
installed. Only the names the reference imports at module import time exist."""
from . import transforms, datasets, models  # noqa: F401
