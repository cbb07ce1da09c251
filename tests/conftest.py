import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    config.addinivalue_line("markers", "slow: a heavy case (4-GiB vectors, 2^23-term MSMs, eight ranks at 2^22): part of the default "
                            "`-m gpu` run, deselect with `-m 'gpu and not slow'`; each one also skips itself when the box "
                            "lacks the host memory or free HBM it needs (need_resources)")


def need_resources(host_gib: float = 0.0, hbm_gib: float = 0.0):
    """skip the calling test unless the box has this much available host memory and free HBM (shared / smaller boxes)"""
    if host_gib:
        avail = None
        try:
            with open("/proc/meminfo") as f:
                for line in f:
                    if line.startswith("MemAvailable:"):
                        avail = int(line.split()[1]) / (1 << 20)
        except OSError:
            pass
        if avail is not None and avail < host_gib:
            pytest.skip(f"needs {host_gib} GiB of available host memory, the box has {avail:.1f}")
    if hbm_gib:
        try:
            import torch

            free, _ = torch.cuda.mem_get_info(0)
        except Exception:  # noqa: BLE001 -- no torch / no device: the test will say so itself
            return
        if free / (1 << 30) < hbm_gib:
            pytest.skip(f"needs {hbm_gib} GiB of free HBM, the device has {free / (1 << 30):.1f}")


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
