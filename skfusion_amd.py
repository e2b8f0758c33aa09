"""Import alias for the package directory ``scikit-fusion_amd/`` (the name mandated for the
repository layout is not a valid Python identifier).  ``import skfusion_amd`` loads that
directory as the package ``skfusion_amd``; sub-modules resolve normally
(``skfusion_amd.fusion``, ``skfusion_amd._native`` ...)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scikit-fusion_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
