"""The C++ host-side mirror of the reference interface (typlonk_amd/host/typlonk_host.hpp):
CPU-only polynomial glue here; the reference's kzg / l0 tests restated in C++ run on the GPU box."""
import os
import subprocess

import pytest

from helpers import ROOT


def _run(name):
    exe = os.path.join(ROOT, "tests", "cpp", name)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().endswith("all ok"), r.stdout
    return r.stdout


def test_poly_glue_cpu(built):
    _run("test_poly_host")


@pytest.mark.gpu
def test_reference_kzg_and_l0_tests_cpp(built):
    out = _run("test_kzg_host")
    for t in ("commit ok", "scalar_mul ok", "l0 ok", "interpolate_then_commit ok"):
        assert t in out
