"""Import shim: the package directory is ``n-bodysimulation_amd`` (not a valid Python identifier),
so this module loads it under the name ``nbodysimulation_amd`` and re-exports it.

    import nbody_amd
    sim = nbody_amd.engine.Simulation(bodies)
"""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "n-bodysimulation_amd")
_NAME = "nbodysimulation_amd"

if _NAME not in sys.modules:
    _spec = importlib.util.spec_from_file_location(
        _NAME, os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR])
    _mod = importlib.util.module_from_spec(_spec)
    sys.modules[_NAME] = _mod
    _spec.loader.exec_module(_mod)

_pkg = sys.modules[_NAME]


def __getattr__(name):
    return getattr(_pkg, name)
