import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` through gpurun)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; see oracle/smesh_oracle.cpp)."""
    from oracle import oracle as o
    o.lib()
    o.set_threads(1)
    o.set_accum_double(False)
    return o


@pytest.fixture(scope="session")
def sm():
    """The product package, with the HIP library loaded and a GPU present."""
    import semantic_meshes_amd as pkg
    from semantic_meshes_amd import _lib
    if _lib.device_count() < 1:
        pytest.fail("gpu test selected but no HIP device is visible (the product has no CPU fallback)")
    return pkg
