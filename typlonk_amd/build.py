"""Build recipe for the in-tree native artefacts (gfx950 only).

  typlonk_amd/libtyplonk_hip.so   HIP kernels + C ABI (include/typlonk.h)       -- hipcc
  tests/cpp/hooks/libtyplonk_hip.so  the same library with the fault-injection hook (-DTYPLONK_TEST_HOOKS, tests only) -- hipcc
  tests/cpp/libff_host_shim.so    host shim over the shared arithmetic headers  -- g++
  tests/cpp/test_{poly,kzg,plonk}_host  tests of the C++ host mirror (typlonk_amd/host) -- g++

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


HIP_UNITS = ["ctx.hip", "ntt_host.hip", "msm_host.hip", "comm.hip", "prover.hip", "ntt_kernels.hip", "msm_sort.hip", "msm_accum.hip", "msm_reduce.hip", "srs_gen.hip", "quotient.hip", "plonk_ops.hip"]
# per-unit flags (none in use).  -DFQ30_ASM_CHAIN for msm_accum.hip was measured: the micro-benchmark's mixed-add ceiling
# rises 7.0 -> 7.3-7.5 G/s (profiles/r02_ubench2_chain.txt) but the real accumulation kernel does not move in a same-box
# A/B (profiles/r02_ab_chain_ntt.txt: 1.85-1.90 ms either way), so the compiler-scheduled form stays.
UNIT_FLAGS: dict[str, list[str]] = {}


def build_hip(force: bool = False) -> str:
    """one object per .hip unit (compiled in parallel, each only when stale), then one link"""
    from concurrent.futures import ThreadPoolExecutor

    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    hdrs.append(os.path.join(ROOT, "include", "typlonk.h"))
    objdir = os.path.join(CSRC, "_obj")
    os.makedirs(objdir, exist_ok=True)
    hipcc = hipcc_path()
    jobs, objs = [], []
    for u in HIP_UNITS:
        src = os.path.join(CSRC, u)
        obj = os.path.join(objdir, u.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *UNIT_FLAGS.get(u, []), "-c", src, "-o", obj])
    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(_run, jobs))
    if jobs or not os.path.exists(LIB):
        _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB])
    return LIB


def build_hip_variant(name: str, unit_flags: dict[str, list[str]]) -> str:
    """a second build of the library under tools/_ab/<name>/ with extra per-unit flags: same-box A/B measurements load it
    through TYPLONK_LIB_PATH (python -m typlonk_amd.build variant <name> <unit.hip>:<flag>[,<flag>] ...)"""
    from concurrent.futures import ThreadPoolExecutor

    out_dir = os.path.join(ROOT, "tools", "_ab", name)
    os.makedirs(out_dir, exist_ok=True)
    hipcc = hipcc_path()
    base_obj = os.path.join(CSRC, "_obj")
    jobs, objs = [], []
    for u in HIP_UNITS:
        if u in unit_flags:
            obj = os.path.join(out_dir, u.replace(".hip", ".o"))
            jobs.append([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *UNIT_FLAGS.get(u, []), *unit_flags[u], "-c",
                         os.path.join(CSRC, u), "-o", obj])
        else:
            obj = os.path.join(base_obj, u.replace(".hip", ".o"))   # the default build's object
        objs.append(obj)
    build_hip()
    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(_run, jobs))
    lib = os.path.join(out_dir, "libtyplonk_hip.so")
    _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib])
    return lib


def build_hip_test_hooks(force: bool = False) -> str:
    """tests/cpp/hooks/libtyplonk_hip.so: the library with comm.hip compiled under -DTYPLONK_TEST_HOOKS (the staging-failure
    injection of tests/test_gpu_dist.py).  Test-only: the shipped library carries no such switch; the dist tests load this
    build through TYPLONK_LIB_PATH in the rank that is meant to fail."""
    out_dir = os.path.join(ROOT, "tests", "cpp", "hooks")
    os.makedirs(out_dir, exist_ok=True)
    lib = os.path.join(out_dir, "libtyplonk_hip.so")
    hipcc = hipcc_path()
    build_hip()
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")] + [os.path.join(ROOT, "include", "typlonk.h")]
    src = os.path.join(CSRC, "comm.hip")
    obj = os.path.join(out_dir, "comm.o")
    base_obj = os.path.join(CSRC, "_obj")
    objs = [obj if u == "comm.hip" else os.path.join(base_obj, u.replace(".hip", ".o")) for u in HIP_UNITS]
    if force or _stale(obj, [src] + hdrs):
        _run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DTYPLONK_TEST_HOOKS", "-c", src, "-o", obj])
    if force or _stale(lib, objs):
        _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib])
    return lib


def build_host_shim(force: bool = False) -> str:
    src = os.path.join(ROOT, "tests", "cpp", "ff_host_shim.cpp")
    out = os.path.join(ROOT, "tests", "cpp", "libff_host_shim.so")
    deps = [src, os.path.join(CSRC, "ff.hpp"), os.path.join(CSRC, "g1.hpp"), os.path.join(CSRC, "fq30.hpp"), os.path.join(CSRC, "g1_host64.hpp"),
            os.path.join(CSRC, "fr30.hpp"), os.path.join(CSRC, "fr_inv.hpp"), os.path.join(CSRC, "transcript.hpp")]
    if force or _stale(out, deps):
        _run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", src, "-o", out])
    return out


def build_host_tests(force: bool = False) -> list[str]:
    """test binaries of the C++ host mirror (typlonk_amd/host/typlonk_host.hpp), linked to the HIP library"""
    outs = []
    hdr = [os.path.join(ROOT, "typlonk_amd", "host", "typlonk_host.hpp"), os.path.join(ROOT, "typlonk_amd", "host", "pairing_host.hpp"),
           os.path.join(ROOT, "tests", "cpp", "circuit_host.hpp"),
           os.path.join(CSRC, "ff.hpp"), os.path.join(CSRC, "fq30.hpp"), os.path.join(CSRC, "g1_host64.hpp"),
           os.path.join(CSRC, "transcript.hpp"), LIB]
    for name in ("test_poly_host", "test_kzg_host", "test_plonk_host", "test_pairing_host", "test_circuit_tables_host", "test_circuit_host",
                 "test_comm_host", "test_comm_ranks_host"):
        src = os.path.join(ROOT, "tests", "cpp", name + ".cpp")
        out = os.path.join(ROOT, "tests", "cpp", name)
        if force or _stale(out, [src] + hdr):
            _run(["g++", "-O2", "-std=c++17", src, "-o", out, "-L", os.path.join(ROOT, "typlonk_amd"), "-ltyplonk_hip",
                  "-Wl,-rpath," + os.path.join(ROOT, "typlonk_amd"), "-Wl,-rpath,$ORIGIN/../../typlonk_amd"])
        outs.append(out)
    return outs


def build_fake_rccl(force: bool = False) -> str:
    """tests/cpp/libfake_rccl.so: the test-only stand-in for librccl (N ranks on one GPU over POSIX shared memory) that
    tests/test_gpu_dist.py selects through TYPLONK_RCCL_LIB -- never loaded by the product on its own"""
    src = os.path.join(ROOT, "tests", "cpp", "fake_rccl.cpp")
    out = os.path.join(ROOT, "tests", "cpp", "libfake_rccl.so")
    if force or _stale(out, [src]):
        _run([hipcc_path(), "-O2", "-std=c++17", "-shared", "-fPIC", "-x", "hip", "--offload-arch=gfx950", src, "-o", out, "-lrt"])
    return out


def build_all(force: bool = False) -> None:
    build_hip(force)
    build_hip_test_hooks(force)
    build_host_shim(force)
    build_host_tests(force)
    build_fake_rccl(force)


if __name__ == "__main__":
    import sys

    if len(sys.argv) >= 3 and sys.argv[1] == "variant":
        flags: dict[str, list[str]] = {}
        for spec in sys.argv[3:]:
            unit, _, fl = spec.partition(":")
            flags[unit] = [f for f in fl.split(",") if f]
        print(build_hip_variant(sys.argv[2], flags))
    else:
        build_all()
