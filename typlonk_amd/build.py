"""Build recipe for the in-tree native artefacts (gfx950 only).

  typlonk_amd/libtyplonk_hip.so   HIP kernels + C ABI (include/typlonk.h)       -- hipcc
  tests/cpp/libff_host_shim.so    host shim over the shared arithmetic headers  -- g++
  oracle/liboracle.so             C restatement of the reference path (checker) -- gcc

Every target is rebuilt only when one of its sources is newer than the output.
"""
from __future__ import annotations

import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "typlonk_amd", "csrc")
LIB = os.path.join(ROOT, "typlonk_amd", "libtyplonk_hip.so")


def _stale(out: str, srcs: list[str]) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(s) > t for s in srcs if os.path.exists(s))


def _run(cmd: list[str]) -> None:
    print("+", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)


def hipcc_path() -> str:
    p = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(p):
        raise RuntimeError("hipcc not found: the HIP library cannot be built (there is no CPU fallback)")
    return p


def build_hip(force: bool = False) -> str:
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "typlonk.h")]
    if force or _stale(LIB, srcs):
        _run([hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
              os.path.join(CSRC, "capi.hip"), "-o", LIB])
    return LIB


def build_host_shim(force: bool = False) -> str:
    src = os.path.join(ROOT, "tests", "cpp", "ff_host_shim.cpp")
    out = os.path.join(ROOT, "tests", "cpp", "libff_host_shim.so")
    deps = [src, os.path.join(CSRC, "ff.hpp"), os.path.join(CSRC, "g1.hpp")]
    if force or _stale(out, deps):
        _run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", src, "-o", out])
    return out


def build_oracle(force: bool = False) -> str | None:
    src = os.path.join(ROOT, "oracle", "typlonk_oracle.c")
    out = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(src):
        return None
    if force or _stale(out, [src]):
        _run(["gcc", "-O2", "-std=c11", "-shared", "-fPIC", src, "-o", out])
    return out


def build_all(force: bool = False) -> None:
    build_hip(force)
    build_host_shim(force)
    build_oracle(force)


if __name__ == "__main__":
    build_all()
