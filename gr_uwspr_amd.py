"""Import shim: the package directory is `gr-uwspr_amd/` (hyphen), which Python
cannot import by name.  `import gr_uwspr_amd` loads it under this module name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gr-uwspr_amd")
_spec = importlib.util.spec_from_file_location(
    "gr_uwspr_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["gr_uwspr_amd"] = _mod
_spec.loader.exec_module(_mod)
