"""Import shim: the package directory is named `lm.rs_amd` (not a valid Python identifier), so
`import lmrs_amd` loads lm.rs_amd/__init__.py under this name."""
import importlib.util as _u
import os as _os
import sys as _sys

_path = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "lm.rs_amd", "__init__.py")
_spec = _u.spec_from_file_location("lmrs_amd", _path, submodule_search_locations=[_os.path.dirname(_path)])
_mod = _u.module_from_spec(_spec)
_sys.modules["lmrs_amd"] = _mod
_spec.loader.exec_module(_mod)
