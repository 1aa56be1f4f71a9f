"""Import shim: the package directory is named `fletcherpenaltysolver.jl_amd` (the dot makes it
un-importable by name), so `import fps_amd` loads it from that directory under this alias."""
import importlib.util as _u
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "fletcherpenaltysolver.jl_amd")
_spec = _u.spec_from_file_location("fps_amd", _os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = _u.module_from_spec(_spec)
_sys.modules["fps_amd"] = _mod
_spec.loader.exec_module(_mod)
