"""The C++ host-side mirror of the reference interface (typlonk_amd/host/typlonk_host.hpp):
CPU-only polynomial glue here; the reference's kzg / l0 tests restated in C++ run on the GPU box."""
import os
import subprocess

import pytest

from helpers import ROOT


def _run(name, *args):
    exe = os.path.join(ROOT, "tests", "cpp", name)
    r = subprocess.run([exe, *args], capture_output=True, text=True, timeout=600)
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


@pytest.mark.gpu
def test_prove_through_the_cpp_mirror(built):
    """plonk::CompiledCircuit::prove (typlonk_host.hpp -> typlonk_prove): squaring chain built in C++, r(zeta) == 0, every
    opening in its trapdoor form, a wrong witness throws"""
    out = _run("test_plonk_host")
    for t in ("prove ok", "openings ok", "commitments ok", "verify ok", "public inputs ok", "bad witness rejected ok"):
        assert t in out


def test_host_pairing_properties_and_equality_with_the_python_statement(built):
    """pairing_host.hpp (C++) against oracle/pairing.py (Python big ints): written separately from the same published
    definition, e(G1, G2) must agree coefficient by coefficient; the C++ side also checks order, bilinearity,
    non-degeneracy and the product form the KZG verifier uses"""
    from oracle import bls12_381 as O
    from oracle import pairing as PR

    out = _run("test_pairing_host")
    assert "g2 ok" in out and "pairing ok" in out
    got = [int(line.split("=")[1], 16) for line in out.splitlines() if line.startswith("e") and "=" in line]
    assert got == PR.pairing(O.G1, PR.G2)


def _cycles(perm):
    seen, out = set(), set()
    for s in range(len(perm)):
        if s in seen:
            continue
        cyc, k = [], s
        while k not in seen:
            seen.add(k)
            cyc.append(k)
            k = perm[k]
        out.add(frozenset(cyc))
    return out


def test_front_end_tables_equal_the_hand_laid_oracle_tables(built):
    """circuit_host.hpp: the recording run of the README circuit (CircuitDescription -> gates, copy constraints, sigma)
    against the tables oracle/plonk_oracle.py writes down by hand for the same circuit; the cycle ORDER is free (the
    reference iterates a HashMap), the partition and the selector rows are not"""
    from oracle import plonk_oracle as PO

    out = _run("test_circuit_tables_host")
    for t in ("permutation ok", "tables ok", "witness ok"):
        assert t in out
    lines = {l.split("=")[0]: l.split("=")[1] for l in out.splitlines() if "=" in l and not l.startswith("circuit2")}
    log_n, cols, q, perm = PO.pythagorean_circuit([3, 4, 5])
    n = 1 << log_n
    for k, name in enumerate(("q_l", "q_r", "q_o", "q_m", "q_c")):
        assert [int(x, 16) for x in lines[f"q{k}"].split(",")] == q[name]
    got_perm = [int(x) for x in lines["perm"].split(",")]
    assert _cycles(got_perm) == _cycles(perm)
    _, sig = PO.compile_permutation(got_perm, n, log_n)
    for i in range(3):
        assert [int(x, 16) for x in lines[f"sigma{i}"].split(",")] == sig[i]


@pytest.mark.parametrize("seed,ops", [(1, 5), (2, 12), (3, 40), (6, 6), (7, 20), (10, 6), (11, 200), (12, 200), (99, 1000), (100, 3000)])
def test_front_end_on_generated_circuits_against_its_python_oracle(built, seed, ops):
    """the C++ front end and oracle/frontend.py (a separate restatement of builder.rs / permutation lib.rs) interpret the
    same generated description -- additions, multiplications and assert_eq over earlier variables, reuse of one variable
    many times, equalities on variables that never enter a gate: same padded size, same gate rows, same partition of the
    cells into copy-constraint cycles, same witness columns; a description the reference would panic on fails in both"""
    from oracle import frontend as F

    exe = os.path.join(ROOT, "tests", "cpp", "test_circuit_tables_host")
    r = subprocess.run([exe, "random", str(seed), str(ops)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    prog = F.random_program(seed, 3, ops)
    try:
        rows, gates, _, perm = F.compile_circuit(F.run_program(prog), 3)
    except ValueError:
        assert r.stdout.strip() == "dangling"
        return
    lines = dict(l.split("=", 1) for l in r.stdout.splitlines() if "=" in l)
    assert int(lines["rows"]) == rows and lines["gates"] == "".join(gates)
    assert _cycles([int(x) for x in lines["perm"].split(",")]) == F.cycles(perm)
    adv = F.witness(F.run_program(prog), [3, 4, 5])
    for c in range(3):
        got = [int(x, 16) for x in lines[f"w{c}"].split(",")] if lines[f"w{c}"] else []
        assert got == adv[c]


@pytest.mark.gpu
def test_reference_circuit_tests_through_the_cpp_front_end(built):
    """plonk/src/builder/test.rs (circuit2_test, circuit2_test_bad_inputs, circuit1_test) written against the C++ mirror:
    build() -> prove() -> verify() with real pairings, the wrong witness refused, a 1000-gate loop circuit"""
    out = _run("test_circuit_host")
    for t in ("circuit2 ok", "circuit1 ok", "chain ok"):
        assert t in out


@pytest.mark.gpu
def test_a_2_20_row_circuit_through_the_cpp_front_end(built):
    """BASELINE config 3's size from the caller's side: 1,048,573 squarings written as a loop over a generic variable,
    compiled (tables, SRS, eight fixed commitments), proved on the GPU and verified with pairings, all from C++"""
    out = _run("test_circuit_host", "big")
    assert "rows=1048576" in out


@pytest.mark.gpu
def test_a_2_22_row_circuit_is_proved_and_accepted_by_the_pairing_verifier(built):
    """BASELINE config 5's size (n = 2^22): the same C++ program -- typlonk_prove on the GPU, then the restated
    plonk::proof::verify with real pairings accepts the proof and rejects a changed evaluation"""
    out = _run("test_circuit_host", "big", "22")
    assert "rows=4194304" in out and "all ok" in out


@pytest.mark.gpu
def test_rccl_exchange_from_a_plain_cpp_process(built):
    """typlonk_comm_* / typlonk_msm_g1_sharded_* called from C++ with neither Python nor PyTorch in the process: the
    library loads the system's librccl by itself (what a Rust host gets)"""
    _run("test_comm_host")


@pytest.mark.parametrize("name", ["test_poly_host", "test_circuit_tables_host"])
def test_host_mirror_under_address_and_ub_sanitizers(built, tmp_path, name):
    """the host-side C++ (typlonk_host.hpp, circuit_host.hpp + the shared field headers) compiled with ASan + UBSan on the
    CPU build (GPU sanitizers are not available on this pool): no report, same answers"""
    exe = str(tmp_path / (name + "_san"))
    lib = os.path.join(ROOT, "typlonk_amd")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           os.path.join(ROOT, "tests", "cpp", name + ".cpp"), "-o", exe, "-L", lib, "-ltyplonk_hip", "-Wl,-rpath," + lib]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0 and r.stdout.strip().endswith("all ok"), r.stdout[-1000:] + r.stderr[-2000:]
