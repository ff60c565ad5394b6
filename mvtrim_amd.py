"""Import shim: the package directory is named `motion-estimated-video-trimmer_amd`
(not a valid Python identifier), so `import mvtrim_amd` loads it from that directory
under this module name."""
import importlib.util
import os
import sys

_PKG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "motion-estimated-video-trimmer_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_PKG, "__init__.py"), submodule_search_locations=[_PKG])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
