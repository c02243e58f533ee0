import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "b-cosification_amd")
for p in (PKG, REPO, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests are skipped (not failed) when no HIP device is present so that a plain `pytest tests/` works
    anywhere; the driver selects them explicitly on the GPU box."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def hip_lib():
    """Built C-ABI library (compiled on demand: hipcc cross-compiles for gfx950 without a GPU)."""
    from bcos_hip import lib
    lib.build()
    return lib.load()


@pytest.fixture(autouse=True)
def _library_options_restored():
    """Tests flip the library's development switches through bcos_set_option (include/bcos_hip.h); whatever a test left
    set -- also when it failed half way -- is put back to the values the library was loaded with."""
    yield
    mod = sys.modules.get("bcos_hip.lib")
    if mod is not None and getattr(mod, "_lib", None) is not None:
        mod.reset_options()
