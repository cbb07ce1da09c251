import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """compile the in-tree native artefacts once per session (no-op when up to date)"""
    from typlonk_amd import build as B

    B.build_all()
    return True


@pytest.fixture(scope="session")
def ctx(built):
    import typlonk_amd

    c = typlonk_amd.Context(0)  # raises when the HIP library or the device is missing: no fallback
    yield c
    c.close()
