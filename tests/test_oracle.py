"""CPU suite: pins both oracles (Python big-int and the C restatement) against the committed
golden vectors and the reference's own test identities.  No GPU, no HIP compute calls."""
import numpy as np
import pytest

from helpers import O, fr_pack, fr_unpack, g1_pack, g1_unpack_one, hex_pt, load_golden
from oracle import coracle as CO

KAT = load_golden("kat.json")
G = lambda v: np.array(O.fr_to_mont_limbs(v), dtype=np.uint64)  # noqa: E731


def test_public_constants():
    assert int(KAT["p"], 16) == O.P and int(KAT["r"], 16) == O.R
    assert O.g1_is_on_curve(O.G1)
    assert O.g1_mul(O.G1, O.R - 1) == O.g1_neg(O.G1)          # r * G = identity
    assert pow(O.FR_ROOT_OF_UNITY, 1 << 31, O.R) == O.R - 1   # primitive 2^32-th root
    assert pow(7, (O.R - 1) >> 32, O.R) == int(KAT["fr_root_of_unity_2_32"], 16)
    assert [hex(x) for x in O.fq_to_mont_limbs(O.GX)] == KAT["g1_x_mont_limbs"]
    # literals from SURVEY.md section 8c (arkworks / zkcrypto generator in Montgomery form)
    assert O.fq_to_mont_limbs(O.GX)[0] == 0x5CB38790FD530C16 and O.fq_to_mont_limbs(O.GY)[5] == 0x0BBC3EFC5008A26A
    assert O.domain_root(16) == 0x2155379D12180CAA88F39A78F1AEB57867A665AE1FCADC91D7118F85CD96B8AD


def test_reference_commit_test_python():
    """kzg/src/lib.rs:95-109"""
    srs = O.srs_from_secret(2, 10)
    assert len(srs) == 13
    c = O.kzg_commit(srs, [1, 2, 3])
    assert c == O.g1_mul(O.G1, O.poly_eval([1, 2, 3], 2)) == hex_pt(KAT["commit_1_2_3_s2"])
    assert c[0] == 0x1098F178F84FC753A76BB63709E9BE91EEC3FF5F7F3A5F4836F34FE8A1A6D6C5578D8FD820573CEF3A01E2BFEF3EAF3A
    assert O.poly_eval([1, 2, 3], 1) == 6
    w, y = O.kzg_open(srs, [1, 2, 3], 1)
    assert y == 6 and w == hex_pt(KAT["open_1_2_3_s2_z1"]["w"]) == O.g1_mul(O.G1, 11)
    # trapdoor form of the pairing check e(W, [s - z]) == e(C - yG, G2): (s - z) W == C - y G
    assert O.g1_mul(w, 2 - 1) == O.g1_add(c, O.g1_neg(O.g1_mul(O.G1, y)))


def test_reference_commit_test_c():
    xy, inf = CO.srs_from_secret(G(2), 13)
    assert [g1_unpack_one(xy[i], inf[i]) for i in range(13)] == O.srs_from_secret(2, 10)
    out, oi = CO.msm_reference(fr_pack([1, 2, 3]), xy, inf)
    assert g1_unpack_one(out, oi) == hex_pt(KAT["commit_1_2_3_s2"])
    q, y = CO.poly_div_linear(fr_pack([1, 2, 3]), G(1))
    assert fr_unpack(y) == [6] and fr_unpack(q) == [5, 3]
    out, oi = CO.msm_reference(q, xy, inf)
    assert g1_unpack_one(out, oi) == hex_pt(KAT["open_1_2_3_s2_z1"]["w"])
    with pytest.raises(AssertionError):   # assert!(srs.len() > polynomial.degree())
        CO.msm_reference(fr_pack([1] * 14), xy, inf)


def test_reference_scalar_mul_test():
    """kzg/src/lib.rs:160-171: commit(9 p) == 9 commit(p)"""
    xy, inf = CO.srs_from_secret(G(0xABCDEF0123), 8)
    p = O.random_frs(21, 5)
    a, ai = CO.msm_reference(fr_pack([9 * c % O.R for c in p]), xy, inf)
    b, bi = CO.msm_reference(fr_pack(p), xy, inf)
    assert g1_unpack_one(a, ai) == O.g1_mul(g1_unpack_one(b, bi), 9)


def test_golden_msm_both_oracles():
    for case in load_golden("msm.json"):
        s, n = int(case["secret"], 16), case["srs_len"]
        sc = [int(x, 16) for x in case["scalars"]]
        exp = hex_pt(case["expected"])
        srs = O.srs_from_secret_fast(s, n)
        assert O.msm_naive(sc, srs) == exp
        xy, inf = g1_pack(srs)
        out, oi = CO.msm_reference(fr_pack(sc) if sc else np.zeros((0, 4), dtype=np.uint64), xy, inf)
        assert g1_unpack_one(out, oi) == exp
        if exp is None:
            assert oi == 1 and not out[:6].any() and [int(x) for x in out[6:]] == O.fq_to_mont_limbs(1)


def test_fair_cpu_pippenger_equals_reference_path():
    """the bucket-method baseline (oracle_msm_pippenger, OpenMP) is the same function as evaluate_in_s: golden
    vectors, every window width used, adversarial scalar sets (zeros, r - 1, repeated, short) and an infinity base"""
    for case in load_golden("msm.json"):
        srs = O.srs_from_secret_fast(int(case["secret"], 16), case["srs_len"])
        sc = [int(x, 16) for x in case["scalars"]]
        xy, inf = g1_pack(srs)
        out, oi, ops, thr = CO.msm_pippenger(fr_pack(sc) if sc else np.zeros((0, 4), dtype=np.uint64), xy, inf, c=5)
        assert g1_unpack_one(out, oi) == hex_pt(case["expected"])
    xy, inf = CO.srs_from_secret(G(0x1234567), 300)
    inf[17] = 1
    sets = [O.random_frs(9, 300), [0] * 300, [O.R - 1] * 300, [7] * 300, [0, O.R - 1] * 150,
            [i % 5 for i in range(300)], [1] + [0] * 299]
    for sc in sets:
        ref, ri = CO.msm_reference(fr_pack(sc), xy, inf)
        for c in (1, 4, 13, 16):
            out, oi, ops, thr = CO.msm_pippenger(fr_pack(sc), xy, inf, c=c)
            assert (out == ref).all() and oi == ri, (c, sc[:3])
            assert thr >= 1
    with pytest.raises(AssertionError):
        CO.msm_pippenger(fr_pack([1] * 301), xy, inf)


def test_golden_ntt_both_oracles():
    assert [int(x, 16) for x in KAT["ntt4_1_2_3_4"]] == O.ntt([1, 2, 3, 4], 2) == fr_unpack(CO.ntt(fr_pack([1, 2, 3, 4]), 2))
    for case in load_golden("ntt.json"):
        L = case["log_n"]
        v = [int(x, 16) for x in case["input"]]
        for key, kw in (("forward", {}), ("inverse", {"inverse": True}), ("coset7_forward", {"coset": 7}),
                        ("coset7_inverse", {"inverse": True, "coset": 7})):
            exp = [int(x, 16) for x in case[key]]
            assert O.ntt(v, L, **kw) == exp
            ckw = {"inverse": kw.get("inverse", False), "coset": G(7) if "coset" in kw else None}
            assert fr_unpack(CO.ntt(fr_pack(v), L, **ckw)) == exp


def test_reference_l0_test_c_oracle():
    """plonk/src/utils.rs:161-177 at the reference's own size N = 2^16"""
    n = 1 << 16
    coeffs = np.tile(G(pow(n, -1, O.R)), (n, 1))
    ev = CO.ntt(coeffs, 16)
    assert (ev[0] == G(1)).all() and not ev[1:].any()


def test_c_oracle_ntt_roundtrip_and_domain_error():
    v = fr_pack(O.random_frs(5, 1 << 12))
    assert (CO.ntt(CO.ntt(v, 12), 12, inverse=True) == v).all()
    assert (CO.ntt(CO.ntt(v, 12, coset=G(7)), 12, inverse=True, coset=G(7)) == v).all()
    import ctypes
    assert CO.lib().oracle_ntt(None, ctypes.c_uint32(33), 0, None) == -3


def test_interpolate_trims_and_open_division():
    """from_coefficients_vec strips trailing zeros (sets MSM lengths); open() divides by X - z."""
    evals = O.ntt([5, 7, 0, 0], 2)
    assert O.interpolate(evals, 2) == [5, 7]
    assert O.interpolate([0, 0, 0, 0], 2) == []
    p = O.random_frs(31, 9)
    z = 12345
    q, y = O.poly_div_linear(p, z)
    # q(X) (X - z) + y == p(X) at a random point
    x = 987654321
    assert (O.poly_eval(q, x) * (x - z) + y) % O.R == O.poly_eval(p, x)


def test_srs_generators_agree():
    xy, inf = CO.srs_pow2_secret(1, 100)
    xy2, inf2 = CO.srs_from_secret(G(2), 100)
    assert (xy == xy2).all() and (inf == inf2).all()
    xy4, _ = CO.srs_pow2_secret(2, 10)
    assert g1_unpack_one(xy4[9], 0) == O.g1_mul(O.G1, pow(4, 9, O.R))


def test_quotient_oracle_satisfies_the_reference_asserts():
    """the identities plonk/src/proof.rs itself asserts: vanishes(line1) :321, vanishes(line4) :361,
    exact division by the vanishing polynomial, and the slice lengths n, n, n - 3"""
    from oracle import plonk_oracle as PO

    for log_n in (2, 3, 5):
        r = PO.prove_round_2_3(log_n, 0x1234567, 0xABCDEF01, 0x55AA55AA77)
        n = r["n"]
        assert r["rem"] == []
        assert PO.divide_by_vanishing_poly(r["line1"], n)[1] == []
        assert PO.divide_by_vanishing_poly(r["line4"], n)[1] == []
        assert len(r["t"]) <= 3 * n - 3
        if log_n >= 3:   # n = 4 has a single gate and a lower-degree quotient
            assert [len(s) for s in PO.slices(r["t"], n)] == [n, n, n - 3]
        # Z starts at 1 and the grand product closes (copy constraints hold)
        assert r["z_evals"][0] == 1
    # naive_mul / divide_by_vanishing_poly against direct evaluation
    a, b = O.random_frs(1, 7), O.random_frs(2, 9)
    x = 987654321
    assert O.poly_eval(PO.naive_mul(a, b), x) == O.poly_eval(a, x) * O.poly_eval(b, x) % O.R
    p = O.random_frs(3, 21)
    q, rem = PO.divide_by_vanishing_poly(p, 8)
    assert (O.poly_eval(q, x) * (pow(x, 8, O.R) - 1) + O.poly_eval(rem, x)) % O.R == O.poly_eval(p, x)


def test_readme_circuit_oracle():
    """SURVEY KAT-5: the README circuit accepts [3,4,5] (exact division, r(zeta) == 0) and rejects [3,4,6]"""
    from oracle import plonk_oracle as PO

    log_n, cols, q, perm = PO.pythagorean_circuit([3, 4, 5])
    assert [c[:5] for c in cols] == [[3, 4, 5, 9, 0], [3, 4, 5, 16, 0], [9, 16, 25, 25, 0]]
    n = 8
    flat = {0: 8, 8: 0, 1: 9, 9: 1, 2: 10, 10: 2, 16: 3, 3: 16, 17: 11, 11: 17, 19: 18, 18: 19}
    assert all(perm[k] == v for k, v in flat.items()) and sum(1 for i, p in enumerate(perm) if i != p) == 12
    srs = O.srs_from_secret_fast(5, n + 3)
    pr = PO.prove(log_n, cols, q, perm, [0] * n, (11, 22, 33), 44, lambda p: O.msm_naive(p, srs))
    assert pr["rem"] == [] and pr["r_open"][1] == 0
    _, bad, _, _ = PO.pythagorean_circuit([3, 4, 6])
    pb = PO.prove(log_n, bad, q, perm, [0] * n, (11, 22, 33), 44, lambda p: O.msm_naive(p, srs))
    assert pb["rem"] != [] or pb["r_open"][1] != 0


# ---- pairing + the reference's verifiers (oracle/pairing.py) ----------------------------------------------
def test_pairing_is_a_pairing():
    from oracle import pairing as PR
    assert PR.g2_is_on_curve(PR.G2) and PR.g2_mul(PR.G2, O.R - 1) == PR.g2_neg(PR.G2)
    e = PR.pairing(O.G1, PR.G2)
    one = PR.f12_one()
    assert e != one and PR.f12_pow(e, O.R) == one
    a, b = 0x1234567, 0xFEDCBA987
    assert PR.pairing(O.g1_mul(O.G1, a), PR.g2_mul(PR.G2, b)) == PR.f12_pow(e, a * b % O.R)
    assert PR.pairing(None, PR.G2) == one and PR.pairing(O.G1, None) == one


def test_reference_commit_test_with_the_real_verify():
    """kzg/src/lib.rs:95-109 in full: commit(1 + 2X + 3X^2) with s = 2, open at 1, verify -- and a wrong value fails"""
    from oracle import pairing as PR
    srs = O.srs_from_secret(2, 10)
    g2, g2s = PR.srs_g2(2)
    c = O.kzg_commit(srs, [1, 2, 3])
    w, y = O.kzg_open(srs, [1, 2, 3], 1)
    assert c == hex_pt(KAT["commit_1_2_3_s2"]) and y == 6
    assert PR.kzg_verify(c, (w, y), 1, g2, g2s)
    assert not PR.kzg_verify(c, (w, 7), 1, g2, g2s)
    assert not PR.kzg_verify(c, (w, y), 2, g2, g2s)


def _cpu_prove_verify(inputs):
    from oracle import pairing as PR
    from oracle import plonk_oracle as PO
    log_n, cols, q_evals, perm = PO.pythagorean_circuit(inputs)
    n = 1 << log_n
    secret = 0xC0FFEE
    srs = O.srs_from_secret_fast(secret, n + 3)
    commit = lambda p: O.kzg_commit(srs, p) if p else None      # noqa: E731
    ch, zeta = (0x1234567DEADBEEF, 0xABCDEF0123456789ABCDEF, 0x55AA55AA77), 0x0F1E2D3C4B5A69788796A5B4C3D2E1F0
    proof = PO.prove(log_n, cols, q_evals, perm, [0] * n, ch, zeta, commit)
    _, sig = PO.compile_permutation(perm, n, log_n)
    sigma_polys = [O.interpolate(s_, log_n) for s_ in sig]
    fixed = [commit(O.interpolate(q_evals[k], log_n)) for k in ("q_l", "q_r", "q_o", "q_m", "q_c")]
    g2, g2s = PR.srs_g2(secret)
    ok = PR.plonk_verify(log_n, proof, fixed, sigma_polys, [commit(p) for p in sigma_polys], PO.COSETS, [0] * n, ch, zeta,
                         g2, g2s)
    return ok, proof


def test_readme_circuit_prove_and_verify_on_cpu():
    """BASELINE config 1 / plonk/src/builder/test.rs:25-37: [3,4,5] proves and verifies, [3,4,6] does not"""
    ok, proof = _cpu_prove_verify([3, 4, 5])
    assert ok and proof["r_open"][1] == 0
    bad, _ = _cpu_prove_verify([3, 4, 6])
    assert not bad


def test_front_end_oracle_reproduces_kat5():
    """oracle/frontend.py on the README circuit: the tables SURVEY.md KAT-5 lists (4 gates -> 8 rows, Mul Mul Mul Add,
    the six copy constraints as transpositions, witness columns a, b, c) and the hand-laid ones of plonk_oracle.py"""
    from oracle import frontend as F
    from oracle import plonk_oracle as PO

    def readme(v):
        a, b, c = v
        a2, b2, c2 = a * a, b * b, c * c
        (a2 + b2).assert_eq(c2)

    rows, gates, sel, perm = F.compile_circuit(readme, 3)
    assert rows == 8 and "".join(gates) == "MMMADDDD"
    log_n, cols, q, pperm = PO.pythagorean_circuit([3, 4, 5])
    assert sel == [[q[k][j] for k in ("q_l", "q_r", "q_o", "q_m", "q_c")] for j in range(8)]
    assert F.cycles(perm) == F.cycles(pperm)
    assert {frozenset(c) for c in F.cycles(perm) if len(c) > 1} == {frozenset(p) for p in ((0, 8), (1, 9), (2, 10), (16, 3), (17, 11), (19, 18))}
    assert F.witness(readme, [3, 4, 5]) == ([3, 4, 5, 9], [3, 4, 5, 16], [9, 16, 25, 25])
    with pytest.raises(ValueError):
        F.compile_circuit(lambda v: v[0].assert_eq(v[1]), 2)


@pytest.mark.parametrize("log_n", [3, 5])
def test_fair_cpu_prover_equals_the_reference_shaped_oracle(log_n):
    """oracle/cpu_prover.py (NTT quotient, batch-inverted grand product, bucket MSMs, all cores: bench.py's cpu_fair leg)
    returns exactly the proof oracle/plonk_oracle.py's restatement of prove() returns (schoolbook quotient, one division
    per cell, per-term MSM): the two CPU paths the bench times are the same function"""
    import numpy as np

    from helpers import fr_pack, g1_pack, g1_unpack_one
    from oracle import cpu_prover as CP
    from oracle import plonk_oracle as PO

    n, cols, q_evals, perm = PO.squaring_chain(log_n)
    _, sig = PO.compile_permutation(perm, n, log_n)
    secret = 0x1D0C5EED
    srs = O.srs_from_secret(secret, n)            # n + 3 points
    alpha, beta, gamma, zeta = 0x1234567DEADBEEF, 0xABCDEF0123456789ABCDEF, 0x55AA55AA77, 0x0F1E2D3C4B5A6978
    ref = PO.prove(log_n, cols, q_evals, perm, [0] * n, (alpha, beta, gamma), zeta, lambda p: O.kzg_commit(srs, p))
    inputs = {"wires": [fr_pack(c) for c in cols], "selectors": [fr_pack(q_evals[k]) for k in ("q_l", "q_r", "q_o", "q_m", "q_c")],
              "sigma": [fr_pack(s) for s in sig], "cosets": PO.COSETS}
    xy, inf = g1_pack(srs)
    lim = lambda v: np.array(O.fr_to_mont_limbs(v), dtype=np.uint64)   # noqa: E731
    got = CP.prove(log_n, inputs, xy, inf, [lim(beta), lim(gamma), lim(alpha), lim(zeta)])
    pt = lambda t: g1_unpack_one(t[0], t[1])                            # noqa: E731
    fr = lambda a: O.fr_from_mont_limbs([int(x) for x in a])            # noqa: E731
    assert [pt(c) for c in got["commit"]] == ref["commit"] and pt(got["z_commit"]) == ref["z_commit"]
    assert [pt(c) for c in got["t_commit"]] == ref["t_commit"]
    opens = ref["open"] + [ref["z_open"], ref["zw_open"], ref["r_open"]]
    assert [pt(w) for w in got["witness"]] == [o[0] for o in opens]
    assert [fr(e) for e in got["evals"]] == [o[1] for o in opens] and fr(got["evals"][5]) == 0
