"""Import shim: the package directory is `online-continual-learning_amd/` (hyphens are not importable), so this
module loads it under the name `ocl_amd` and replaces itself in sys.modules.  `import ocl_amd`, then
`ocl_amd.name_match.agents['SCR']`, `from ocl_amd.agents.scr import SupContrastReplay`, ... work as usual."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "online-continual-learning_amd")
_spec = importlib.util.spec_from_file_location("ocl_amd", os.path.join(_pkg_dir, "__init__.py"),
                                               submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["ocl_amd"] = _mod
_spec.loader.exec_module(_mod)
