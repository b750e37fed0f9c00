import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """Build (or reuse) the in-tree native libraries once per session. hipcc cross-compiles without a GPU."""
    from raytracingdenoiser_amd import build as b

    b.build_product(numerics="fast")   # lib/libNRD_hip.so: the product
    b.build_product(numerics="exact")  # lib/libNRD_hip_exact.so: the bit-exact regression build
    b.build_oracle()
    yield


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
