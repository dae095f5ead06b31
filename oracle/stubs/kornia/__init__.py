"""TEST INFRASTRUCTURE ONLY: stand-in for kornia==0.4.1 (absent). See augmentation.py."""
from . import augmentation  # noqa: F401
