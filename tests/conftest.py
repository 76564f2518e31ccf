import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built_library():
    """The C-ABI shared library (built by __graft_entry__.build(); cross-compiles without a GPU)."""
    import __graft_entry__ as ge
    if not os.path.exists(ge.LIB):
        ge.build()
    from pl_fem_vectoriel_amd import _native
    return _native.load_library()


@pytest.fixture(scope="session")
def c1_geometry():
    from pl_fem_vectoriel_amd import MCFGeometry
    return MCFGeometry(7, 8.0, 1.5, 1.535, 1.0, wavelength_um=1.55)


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible (there is no CPU fallback)")
    return 0
