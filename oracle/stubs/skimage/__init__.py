"""TEST INFRASTRUCTURE ONLY."""
from . import filters  # noqa: F401
