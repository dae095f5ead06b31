"""online-continual-learning_amd — MI355X-native replay step for RaptorMai/online-continual-learning.

Only the hot path named in BASELINE.json lives here: the ER / SCR inner loops, the ASER / MIR retrieve and
update hooks, SupCon loss, NCM classifier and the Reduced-ResNet18 forward/backward, all executed by
hand-written gfx950 HIP kernels (csrc/, C-ABI in include/ocl_hip.h) behind the reference's own
agents / retrieve_methods / update_methods registries (name_match.py).

The directory name contains a hyphen, so the package is imported as `ocl_amd` through the loader module
at the repository root (ocl_amd.py).
"""
__version__ = "0.1.0"

from . import name_match  # noqa: E402,F401  (the plugin registries: `ocl_amd.name_match.agents[...]` as INTEGRATION.md uses them)
