import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "rehearsal: several RCCL ranks sharing ONE GPU (fake host ids, loopback sockets); collected last")


def pytest_collection_modifyitems(config, items):
    """Rehearsals on borrowed terms go LAST, whatever the file order: with -x a flake there can never leave an oracle
    comparison or a drop-in test un-run (stable sort: everything else keeps its order)."""
    items.sort(key=lambda it: 1 if it.get_closest_marker("rehearsal") else 0)


@pytest.fixture(scope="session")
def oracle():
    """The CPU checker (test infrastructure)."""
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def nb():
    """The product package; the HIP library must already be built (no fallback)."""
    import nbody_amd
    lib = os.path.join(ROOT, "n-bodysimulation_amd", "libnbody_hip.so")
    drv = os.path.join(ROOT, "n-bodysimulation_amd", "bin", "nbody_headless")
    if not (os.path.exists(lib) and os.path.exists(drv)):
        import __graft_entry__            # a snapshot taken before build(): compile in-tree now (hipcc needs no GPU)
        __graft_entry__.build()
    nbody_amd.load()
    return nbody_amd


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same_bits(a, b):
    """Bitwise equality, except that +0 and -0 compare equal."""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return bool(np.all((bits(a) == bits(b)) | ((a == 0) & (b == 0))))
