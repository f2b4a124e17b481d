"""MI355X-native all-pairs N-body force/integrate engine (drop-in for the GPU step of
LienoPC/N-BodySimulation). See DESIGN.md and include/nbody.h.

The directory name carries a hyphen, so import it through the repo-root shim:
``import nbody_amd`` (or ``importlib`` with this directory as the package path)."""
from ._lib import (DEFAULT_DT, DEFAULT_EPS2, KERNEL_FAST, KERNEL_ONESIDED, KERNEL_STRICT, KERNEL_SYMMETRIC, NBodyError, exported_symbols,  # noqa: F401
                   load)

__all__ = ["DEFAULT_DT", "DEFAULT_EPS2", "KERNEL_FAST", "KERNEL_ONESIDED", "KERNEL_STRICT", "KERNEL_SYMMETRIC", "NBodyError", "exported_symbols", "load"]


def __getattr__(name):
    # engine / sharded import torch; keep `import nbody_amd` light for symbol checks
    if name in ("engine", "sharded"):
        import importlib
        return importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(name)
