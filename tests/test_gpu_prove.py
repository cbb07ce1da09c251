"""End-to-end parity of the device-side prover (typlonk_prover_round1/2/3) against the oracle restatement
of plonk::proof::prove (/root/reference/plonk/src/proof.rs:96-194): every commitment, every opening
witness and every evaluation of the proof, bit for bit, for the squaring-chain circuit."""
import numpy as np
import pytest

from helpers import O, fr_pack, g1_pack, g1_unpack_one
from oracle import coracle as CO
from oracle import plonk_oracle as PO

pytestmark = pytest.mark.gpu

CH = (0x1234567DEADBEEF, 0xABCDEF0123456789ABCDEF, 0x55AA55AA77)   # alpha, beta, gamma
ZETA = 0x0F1E2D3C4B5A69788796A5B4C3D2E1F0


def _limbs(v):
    return np.array(O.fr_to_mont_limbs(v), dtype=np.uint64)


def _up(ctx, vals, n):
    b = ctx.alloc(n)
    b.upload(fr_pack(list(vals) + [0] * (n - len(vals))))
    return b


def _setup(ctx, log_n, x0=3):
    n, cols, q_evals, perm = PO.squaring_chain(log_n, x0)
    _, sig = PO.compile_permutation(perm, n, log_n)
    sel = [_up(ctx, O.interpolate(q_evals[k], log_n), n) for k in ("q_l", "q_r", "q_o", "q_m", "q_c")]
    sgm = [_up(ctx, O.interpolate(s, log_n), n) for s in sig]
    cid = ctx.circuit_load(log_n, sel, sgm)
    for b in sel + sgm:
        b.free()
    return n, cols, q_evals, perm, cid


V_BATCH = 0x7E57AB1E0F0F0F0F1234


def _gpu_prove(ctx, sid, cid, cols, n, pi_evals=None, batched=False):
    wires = [_up(ctx, c, n) for c in cols]
    pi = _up(ctx, pi_evals or [0] * n, n)
    alpha, beta, gamma = CH
    try:
        return ctx.prove(sid, cid, wires, pi, [_limbs(k) for k in PO.COSETS],
                         lambda commits: (_limbs(beta), _limbs(gamma)), lambda commits: (_limbs(alpha), _limbs(ZETA)),
                         challenge_v=(lambda evals: _limbs(V_BATCH)) if batched else None)
    finally:
        for b in wires + [pi]:
            b.free()


@pytest.mark.parametrize("log_n", [3, 4, 6])
def test_prove_equals_reference_flow(ctx, log_n):
    n, cols, q_evals, perm, cid = _setup(ctx, log_n)
    secret = 0x0123456789ABCDEF0123456789ABCDEF
    sid = ctx.srs_generate(_limbs(secret), n + 3)
    xy, inf = ctx.srs_download(sid)

    def commit(coeffs):
        out, oi = CO.msm_reference(fr_pack(coeffs) if coeffs else np.zeros((0, 4), dtype=np.uint64), xy, inf)
        return g1_unpack_one(out, oi)

    ref = PO.prove(log_n, cols, q_evals, perm, [0] * n, CH, ZETA, commit)
    assert ref["rem"] == [] and ref["r_open"][1] == 0
    got = _gpu_prove(ctx, sid, cid, cols, n)
    pt = lambda t: g1_unpack_one(t[0], t[1])              # noqa: E731
    fr = lambda a: O.fr_from_mont_limbs([int(v) for v in a])  # noqa: E731
    assert [pt(c) for c in got["commit"]] == ref["commit"]
    assert pt(got["z_commit"]) == ref["z_commit"]
    assert [pt(c) for c in got["t_commit"]] == ref["t_commit"]
    ref_w = [o[0] for o in ref["open"]] + [ref["z_open"][0], ref["zw_open"][0], ref["r_open"][0]]
    ref_e = [o[1] for o in ref["open"]] + [ref["z_open"][1], ref["zw_open"][1], ref["r_open"][1]]
    assert [pt(w) for w in got["witness"]] == ref_w
    assert [fr(e) for e in got["evals"]] == ref_e
    assert fr(got["evals"][5]) == 0                      # the verifier's r(zeta) == 0 (proof.rs:234-235)
    ctx.circuit_free(cid)
    ctx.srs_free(sid)


@pytest.mark.parametrize("log_n", [3, 5])
def test_batched_openings_equal_oracle_and_the_combination_of_the_six(ctx, log_n):
    """round3_evals + round4_batched: same commitments and evaluations as the six-opening proof; W[0] is
    the oracle's open() of a + v b + v^2 c + v^3 Z + v^4 r AND sum_i v^i W_i of the reference-shaped
    witnesses (division by X - zeta is linear); W[1] is the witness of Z at zeta*w"""
    n, cols, q_evals, perm, cid = _setup(ctx, log_n)
    sid = ctx.srs_generate(_limbs(0xFEEDFACE12345), n + 3)
    xy, inf = ctx.srs_download(sid)

    def commit(coeffs):
        out, oi = CO.msm_reference(fr_pack(coeffs) if coeffs else np.zeros((0, 4), dtype=np.uint64), xy, inf)
        return g1_unpack_one(out, oi)

    ref = PO.prove(log_n, cols, q_evals, perm, [0] * n, CH, ZETA, commit)
    got = _gpu_prove(ctx, sid, cid, cols, n, batched=True)
    six = _gpu_prove(ctx, sid, cid, cols, n)
    pt = lambda t: g1_unpack_one(t[0], t[1])              # noqa: E731
    assert got["batched"] and len(got["witness"]) == 2
    for key in ("commit", "t_commit"):
        assert [pt(c) for c in got[key]] == [pt(c) for c in six[key]] == ref[key]
    assert pt(got["z_commit"]) == ref["z_commit"]
    assert all((x == y).all() for x, y in zip(got["evals"], six["evals"]))
    w_ref, y_ref = PO.batched_opening(ref["wires"] + [ref["z"], ref["r"]], V_BATCH, ZETA, commit)
    assert pt(got["witness"][0]) == w_ref
    fr = lambda a: O.fr_from_mont_limbs([int(v) for v in a])  # noqa: E731
    ev = [fr(e) for e in got["evals"]]
    assert y_ref == sum(pow(V_BATCH, i, O.R) * ev[j] for i, j in enumerate((0, 1, 2, 3, 5))) % O.R
    comb = None
    for i, j in enumerate((0, 1, 2, 3, 5)):
        term = O.g1_mul(pt(six["witness"][j]), pow(V_BATCH, i, O.R))
        comb = term if comb is None else O.g1_add(comb, term)
    assert pt(got["witness"][0]) == comb
    assert pt(got["witness"][1]) == pt(six["witness"][4]) == ref["zw_open"][0]
    ctx.circuit_free(cid)
    ctx.srs_free(sid)


def test_batched_round_order_is_enforced(ctx):
    import ctypes as C
    from typlonk_amd.capi import TyplonkError, ERR_INVALID_ARG

    n, cols, q_evals, perm, cid = _setup(ctx, 3)
    sid = ctx.srs_generate(_limbs(2), n + 3)
    wires = [_up(ctx, c, n) for c in cols]
    lib = ctx.lib
    pr = C.c_void_p()
    cxy, cinf = ((C.c_uint64 * 12) * 3)(), (C.c_uint8 * 3)()
    w = (C.c_void_p * 3)(*[b.handle.value for b in wires])
    ctx._chk(lib.typlonk_prover_round1(ctx.h, sid, cid, w, None, C.byref(pr), C.byref(cxy), C.byref(cinf)))
    v = _limbs(5)
    from typlonk_amd.capi import ProofBatched
    pb = ProofBatched()
    rc = lib.typlonk_prover_round4_batched(pr, v.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(pb))
    assert rc == ERR_INVALID_ARG
    lib.typlonk_prover_free(pr)
    for b in wires:
        b.free()
    ctx.circuit_free(cid)
    ctx.srs_free(sid)


def test_prove_2_12_self_checks(ctx):
    """n = 2^12: the full oracle flow is too slow (schoolbook quotient), so check what a verifier
    with the trapdoor can check: r(zeta) == 0, and every opening (s - z) W == C - y G with s = 2."""
    log_n = 12
    n, cols, q_evals, perm, cid = _setup(ctx, log_n, x0=11)
    sid = ctx.srs_generate(_limbs(2), n + 3)
    got = _gpu_prove(ctx, sid, cid, cols, n)
    pt = lambda t: g1_unpack_one(t[0], t[1])              # noqa: E731
    fr = lambda a: O.fr_from_mont_limbs([int(v) for v in a])  # noqa: E731
    assert fr(got["evals"][5]) == 0
    w = O.domain_root(log_n)
    checks = [(got["commit"][0], got["witness"][0], got["evals"][0], ZETA),
              (got["commit"][1], got["witness"][1], got["evals"][1], ZETA),
              (got["commit"][2], got["witness"][2], got["evals"][2], ZETA),
              (got["z_commit"], got["witness"][3], got["evals"][3], ZETA),
              (got["z_commit"], got["witness"][4], got["evals"][4], ZETA * w % O.R)]
    for commit, wit, ev, z in checks:
        lhs = O.g1_mul(pt(wit), (2 - z) % O.R)
        rhs = O.g1_add(pt(commit), O.g1_neg(O.g1_mul(O.G1, fr(ev))))
        assert lhs == rhs
    # wire commitments against the identity commit(p) == [p(s)]G with the C oracle's Horner
    for col, c in zip(cols, got["commit"]):
        coeffs = CO.ntt(fr_pack(col), log_n, inverse=True)
        exp, einf = CO.g1_mul_generator(CO.poly_eval(coeffs, _limbs(2)))
        assert (c[0] == exp).all() and c[1] == einf
    ctx.circuit_free(cid)
    ctx.srs_free(sid)


def test_null_public_inputs_equal_zero_column(ctx):
    """public_inputs = NULL is the zero polynomial: identical proof to an explicit all-zero column"""
    n, cols, q_evals, perm, cid = _setup(ctx, 5)
    sid = ctx.srs_generate(_limbs(7), n + 3)
    a = _gpu_prove(ctx, sid, cid, cols, n)                       # explicit zero column
    wires = [_up(ctx, c, n) for c in cols]
    alpha, beta, gamma = CH
    b = ctx.prove(sid, cid, wires, None, [_limbs(k) for k in PO.COSETS],
                  lambda c: (_limbs(beta), _limbs(gamma)), lambda c: (_limbs(alpha), _limbs(ZETA)))
    for key in ("commit", "t_commit", "witness"):
        for (x, xi), (y, yi) in zip(a[key], b[key]):
            assert (x == y).all() and xi == yi
    for x, y in zip(a["evals"], b["evals"]):
        assert (x == y).all()
    for w in wires:
        w.free()
    ctx.circuit_free(cid)
    ctx.srs_free(sid)


def test_prove_rejects_wrong_round_order_and_short_srs(ctx):
    from typlonk_amd.capi import TyplonkError, ERR_LENGTH

    n, cols, q_evals, perm, cid = _setup(ctx, 3)
    sid = ctx.srs_generate(_limbs(2), n - 1)            # too short: the reference's assert! in commit
    with pytest.raises(TyplonkError) as e:
        _gpu_prove(ctx, sid, cid, cols, n)
    assert e.value.code == ERR_LENGTH
    ctx.circuit_free(cid)
    ctx.srs_free(sid)


def test_device_built_squaring_chain_matches_oracle_tables(ctx):
    """typlonk_amd/circuits.py builds the benchmark circuit in HBM; its witness / sigma tables are the
    oracle's (with the same blinders injected) and its proof verifies r(zeta) == 0"""
    from typlonk_amd.circuits import SquaringChain
    from helpers import fr_unpack

    log_n = 5
    sc = SquaringChain(ctx, log_n, x0=3)
    n = sc.n
    cols_dev = [fr_unpack(b.download()) for b in sc.wire_evals]
    bl = [c[n - 3:] for c in cols_dev]
    _, cols, q_evals, perm = PO.squaring_chain(log_n, 3, blinders=bl)
    assert cols_dev == cols
    sid = ctx.srs_generate(_limbs(2), n + 3)
    alpha, beta, gamma = CH
    got = ctx.prove(sid, sc.circuit, sc.wire_evals, sc.pi_evals, sc.cosets,
                    lambda c: (_limbs(beta), _limbs(gamma)), lambda c: (_limbs(alpha), _limbs(ZETA)))
    xy, inf = ctx.srs_download(sid)

    def commit(coeffs):
        out, oi = CO.msm_reference(fr_pack(coeffs) if coeffs else np.zeros((0, 4), dtype=np.uint64), xy, inf)
        return g1_unpack_one(out, oi)

    ref = PO.prove(log_n, cols, q_evals, perm, [0] * n, CH, ZETA, commit)
    pt = lambda t: g1_unpack_one(t[0], t[1])   # noqa: E731
    assert [pt(c) for c in got["commit"]] == ref["commit"]
    assert pt(got["z_commit"]) == ref["z_commit"]
    assert [pt(c) for c in got["t_commit"]] == ref["t_commit"]
    assert O.fr_from_mont_limbs([int(v) for v in got["evals"][5]]) == 0
    sc.free()
    ctx.srs_free(sid)


def _load_circuit(ctx, log_n, q_evals, perm):
    n = 1 << log_n
    _, sig = PO.compile_permutation(perm, n, log_n)
    sel = [_up(ctx, O.interpolate(q_evals[k], log_n), n) for k in ("q_l", "q_r", "q_o", "q_m", "q_c")]
    sgm = [_up(ctx, O.interpolate(s, log_n), n) for s in sig]
    cid = ctx.circuit_load(log_n, sel, sgm)
    for b in sel + sgm:
        b.free()
    return cid


def test_readme_pythagorean_circuit(ctx):
    """BASELINE config 1 / README.md:10-34 / plonk/src/builder/test.rs:25-37: inputs [3,4,5] prove and
    'verify' (r(zeta) == 0, every element equal to the oracle's prove()); [3,4,6] must not"""
    log_n, cols, q_evals, perm = PO.pythagorean_circuit([3, 4, 5])
    n = 8
    cid = _load_circuit(ctx, log_n, q_evals, perm)
    sid = ctx.srs_generate(_limbs(0xC0FFEE), n + 3)          # Srs::random(n): 11 points
    assert ctx.srs_len(sid) == 11
    xy, inf = ctx.srs_download(sid)

    def commit(coeffs):
        out, oi = CO.msm_reference(fr_pack(coeffs) if coeffs else np.zeros((0, 4), dtype=np.uint64), xy, inf)
        return g1_unpack_one(out, oi)

    ref = PO.prove(log_n, cols, q_evals, perm, [0] * n, CH, ZETA, commit)
    assert ref["rem"] == [] and ref["r_open"][1] == 0
    got = _gpu_prove(ctx, sid, cid, cols, n)
    pt = lambda t: g1_unpack_one(t[0], t[1])                   # noqa: E731
    fr = lambda a: O.fr_from_mont_limbs([int(v) for v in a])   # noqa: E731
    assert [pt(c) for c in got["commit"]] == ref["commit"] and pt(got["z_commit"]) == ref["z_commit"]
    assert [pt(c) for c in got["t_commit"]] == ref["t_commit"]
    assert [pt(w) for w in got["witness"]] == [o[0] for o in ref["open"]] + [ref["z_open"][0], ref["zw_open"][0], ref["r_open"][0]]
    assert fr(got["evals"][5]) == 0
    # bad inputs (circuit2_test_bad_inputs): the copy constraint c_3 ~ c_2 fails, the verifier's r(zeta) != 0
    # (the reference test is #[should_panic]; here the prover reports TYPLONK_ERR_UNSATISFIED, in both proof shapes,
    #  and the context is usable afterwards)
    from typlonk_amd.capi import ERR_UNSATISFIED, TyplonkError
    _, bad_cols, _, _ = PO.pythagorean_circuit([3, 4, 6])
    for batched in (False, True):
        with pytest.raises(TyplonkError) as e:
            _gpu_prove(ctx, sid, cid, bad_cols, n, batched=batched)
        assert e.value.code == ERR_UNSATISFIED
    again = _gpu_prove(ctx, sid, cid, cols, n)
    assert [pt(c) for c in again["commit"]] == ref["commit"] and fr(again["evals"][5]) == 0
    ctx.circuit_free(cid)
    ctx.srs_free(sid)


def test_gpu_proof_passes_the_reference_verifier_with_real_pairings(ctx):
    """End to end: the proof typlonk_prover_round1/2/3 produces for the squaring chain (n = 32) is accepted by the
    restated plonk::proof::verify (oracle/pairing.py: 12 pairings, linearisation commitment built from the GPU's
    own selector / sigma commitments); a tampered evaluation or witness is rejected"""
    from oracle import pairing as PR

    log_n = 5
    n, cols, q_evals, perm, cid = _setup(ctx, log_n)
    secret = 0x5EC2E7D00D
    sid = ctx.srs_generate(_limbs(secret), n + 3)
    got = _gpu_prove(ctx, sid, cid, cols, n)
    pt = lambda t: g1_unpack_one(t[0], t[1])                   # noqa: E731
    fr = lambda a: O.fr_from_mont_limbs([int(v) for v in a])   # noqa: E731
    ev = [fr(e) for e in got["evals"]]
    wit = [pt(w) for w in got["witness"]]
    proof = {"commit": [pt(c) for c in got["commit"]], "open": [(wit[i], ev[i]) for i in range(3)],
             "z_commit": pt(got["z_commit"]), "z_open": (wit[3], ev[3]), "zw_open": (wit[4], ev[4]),
             "t_commit": [pt(c) for c in got["t_commit"]], "r_open": (wit[5], ev[5])}
    _, sig = PO.compile_permutation(perm, n, log_n)
    sigma_polys = [O.interpolate(s, log_n) for s in sig]
    gpu_commit = lambda cf: pt(ctx.msm(sid, fr_pack(cf) if cf else np.zeros((0, 4), dtype=np.uint64)))   # noqa: E731
    fixed = [gpu_commit(O.interpolate(q_evals[k], log_n)) for k in ("q_l", "q_r", "q_o", "q_m", "q_c")]
    sigma_c = [gpu_commit(p) for p in sigma_polys]
    g2, g2s = PR.srs_g2(secret)
    alpha, beta, gamma = CH
    args = (fixed, sigma_polys, sigma_c, PO.COSETS, [0] * n, (alpha, beta, gamma), ZETA, g2, g2s)
    assert PR.plonk_verify(log_n, proof, *args)
    bad = dict(proof, open=[(wit[0], (ev[0] + 1) % O.R)] + proof["open"][1:])
    assert not PR.plonk_verify(log_n, bad, *args)
    bad = dict(proof, r_open=(O.g1_add(wit[5], O.G1), ev[5]))
    assert not PR.plonk_verify(log_n, bad, *args)
    ctx.circuit_free(cid)
    ctx.srs_free(sid)


def test_prove_with_the_transcript_and_verify_by_recomputing_challenges(ctx):
    """prove() without injected challenges squeezes them natively (csrc/transcript.hpp, the reference's ChallengeGenerator); the
    verifier side recomputes them from the commitments as verify_challenges does (proof.rs:236-246)"""
    from oracle import pairing as PR
    import transcript_ref as T

    log_n = 4
    n, cols, q_evals, perm, cid = _setup(ctx, log_n)
    secret = 0x77AA55
    sid = ctx.srs_generate(_limbs(secret), n + 3)
    wires = [_up(ctx, c, n) for c in cols]
    got = ctx.prove(sid, cid, wires, None, [_limbs(k) for k in PO.COSETS])
    pt = lambda t: g1_unpack_one(t[0], t[1])                   # noqa: E731
    fr = lambda a: O.fr_from_mont_limbs([int(v) for v in a])   # noqa: E731
    beta, gamma = [fr(x) for x in T.challenge12(got["commit"])]
    alpha, zeta = [fr(x) for x in T.challenge34(got["commit"] + [got["z_commit"]])]
    ev = [fr(e) for e in got["evals"]]
    wit = [pt(w) for w in got["witness"]]
    assert ev[5] == 0
    proof = {"commit": [pt(c) for c in got["commit"]], "open": [(wit[i], ev[i]) for i in range(3)],
             "z_commit": pt(got["z_commit"]), "z_open": (wit[3], ev[3]), "zw_open": (wit[4], ev[4]),
             "t_commit": [pt(c) for c in got["t_commit"]], "r_open": (wit[5], ev[5])}
    _, sig = PO.compile_permutation(perm, n, log_n)
    sigma_polys = [O.interpolate(s, log_n) for s in sig]
    gpu_commit = lambda cf: pt(ctx.msm(sid, fr_pack(cf) if cf else np.zeros((0, 4), dtype=np.uint64)))   # noqa: E731
    fixed = [gpu_commit(O.interpolate(q_evals[k], log_n)) for k in ("q_l", "q_r", "q_o", "q_m", "q_c")]
    g2, g2s = PR.srs_g2(secret)
    assert PR.plonk_verify(log_n, proof, fixed, sigma_polys, [gpu_commit(p) for p in sigma_polys], PO.COSETS, [0] * n,
                           (alpha, beta, gamma), zeta, g2, g2s)
    for b in wires:
        b.free()
    ctx.circuit_free(cid)
    ctx.srs_free(sid)


@pytest.mark.parametrize("log_n", [4, 7])
def test_native_prove_equals_the_round_by_round_flow(ctx, log_n):
    """typlonk_prove (one native call, transcript in csrc/transcript.hpp) returns exactly what Context.prove assembles
    from the three rounds with the Python statement of the transcript -- commitments, witnesses, evaluations -- and the
    challenges it used are the ones recomputed from its own commitments (verify_challenges, proof.rs:236-246)"""
    import transcript_ref as T
    from typlonk_amd.capi import ERR_UNSATISFIED, TyplonkError

    n, cols, q_evals, perm, cid = _setup(ctx, log_n)
    sid = ctx.srs_generate(_limbs(0x1CEB00DA), n + 3)
    wires = [_up(ctx, c, n) for c in cols]
    ks = [_limbs(k) for k in PO.COSETS]
    ref = ctx.prove(sid, cid, wires, None, ks)          # Python transcript between the rounds
    got = ctx.prove_native(sid, cid, wires, None, ks)
    same = lambda a, b: bool((np.asarray(a[0]) == np.asarray(b[0])).all() and int(a[1]) == int(b[1]))   # noqa: E731
    for key in ("commit", "t_commit", "witness"):
        assert len(got[key]) == len(ref[key]) and all(same(a, b) for a, b in zip(got[key], ref[key])), key
    assert same(got["z_commit"], ref["z_commit"])
    assert all((a == b).all() for a, b in zip(got["evals"], ref["evals"]))
    assert not got["evals"][5].any()
    beta, gamma = T.challenge12(got["commit"])
    alpha, zeta = T.challenge34(got["commit"] + [got["z_commit"]])
    for name, want in (("beta", beta), ("gamma", gamma), ("alpha", alpha), ("zeta", zeta)):
        assert (got["challenges"][name] == want).all(), name
    # a witness that violates a gate: the reference panics, the native call reports it
    bad_cols = [list(c) for c in cols]
    bad_cols[2][1] = (bad_cols[2][1] + 1) % O.R
    bad = [_up(ctx, c, n) for c in bad_cols]
    with pytest.raises(TyplonkError) as e:
        ctx.prove_native(sid, cid, bad, None, ks)
    assert e.value.code == ERR_UNSATISFIED
    for b in wires + bad:
        b.free()
    ctx.circuit_free(cid)
    ctx.srs_free(sid)


@pytest.mark.parametrize("log_n,tables", [(4, False), (10, False), (16, True)])
def test_prove_from_host_columns_equals_prove_from_device_buffers(ctx, log_n, tables):
    """typlonk_prove_host: the padded, blinded columns handed over in HOST memory -- how the reference holds them when
    prove() starts (plonk/src/proof.rs:43-53) and what the Rust layer's Backend::prove now passes -- uploaded column by column
    beside round 1's kernels.  Same proof as typlonk_prove on device buffers, element for element, with and without a
    public-input column; a null column is refused."""
    from typlonk_amd.capi import ERR_INVALID_ARG, TyplonkError
    from typlonk_amd.circuits import SquaringChain

    n = 1 << log_n
    sid = ctx.srs_generate(_limbs(0xBEEF5), n + 3)
    if tables:
        ctx.srs_precompute(sid, 0)
    chain = SquaringChain(ctx, log_n)
    try:
        host_w = [b.download() for b in chain.wire_evals]
        host_pi = np.zeros((n, 4), dtype=np.uint64)          # the circuit's public inputs are [0]: an explicit zero column
        dev_pi = ctx.alloc(n)
        dev_pi.upload(host_pi)
        same = lambda a, b: bool((np.asarray(a[0]) == np.asarray(b[0])).all() and int(a[1]) == int(b[1]))   # noqa: E731
        for with_pi in (True, False):
            ref = ctx.prove_native(sid, chain.circuit, chain.wire_evals, dev_pi if with_pi else None, chain.cosets)
            got = ctx.prove_native_host(sid, chain.circuit, host_w, host_pi if with_pi else None, chain.cosets)
            for key in ("commit", "t_commit", "witness"):
                assert all(same(a, b) for a, b in zip(got[key], ref[key])), (key, with_pi)
            assert same(got["z_commit"], ref["z_commit"])
            assert all((a == b).all() for a, b in zip(got["evals"], ref["evals"]))
            assert all((got["challenges"][k] == ref["challenges"][k]).all() for k in ("beta", "gamma", "alpha", "zeta"))
            assert not got["evals"][5].any()
        import ctypes as C

        w = (C.POINTER(C.c_uint64) * 3)()          # three null columns
        from typlonk_amd.capi import Proof

        ks = ((C.c_uint64 * 4) * 3)()
        rc = ctx.lib.typlonk_prove_host(ctx.h, sid, chain.circuit, w, None, C.byref(ks), C.byref(Proof()))
        assert rc == ERR_INVALID_ARG
        dev_pi.free()
    finally:
        chain.free()
        ctx.srs_free(sid)


@pytest.mark.parametrize("pipe", ["0", "1"])
def test_round3_queueing_switch_does_not_change_a_bit(built, monkeypatch, pipe):
    """TYPLONK_PROVER_PIPE: round 3's nine commitments behind one fence (round 5, the default) or queued as rounds 1-4
    did -- scheduling only: the proof of a 2^14-row squaring chain over a table-mode SRS is the same, element for element,
    as the one the default context of this process produces"""
    import typlonk_amd
    from typlonk_amd.circuits import SquaringChain

    monkeypatch.setenv("TYPLONK_PROVER_PIPE", pipe)
    log_n = 14
    proofs = []
    for fresh in (True, False):
        if not fresh:
            monkeypatch.delenv("TYPLONK_PROVER_PIPE")
        c2 = typlonk_amd.Context(0)
        try:
            sid = c2.srs_generate(_limbs(0xFACE), (1 << log_n) + 3)
            c2.srs_precompute(sid, 0)
            chain = SquaringChain(c2, log_n)
            proofs.append(c2.prove_native(sid, chain.circuit, chain.wire_evals, chain.pi_evals, chain.cosets))
        finally:
            c2.close()
    a, b = proofs
    same = lambda x, y: bool((np.asarray(x[0]) == np.asarray(y[0])).all() and int(x[1]) == int(y[1]))   # noqa: E731
    for key in ("commit", "t_commit", "witness"):
        assert all(same(x, y) for x, y in zip(a[key], b[key])), key
    assert same(a["z_commit"], b["z_commit"]) and all((x == y).all() for x, y in zip(a["evals"], b["evals"]))
    assert not a["evals"][5].any()


def test_config5_prove_at_2_22_both_shapes(ctx):
    """BASELINE config 5's size, n = 2^22 (quotient domain 2^24): a full prove() in both proof shapes on one GPU.
    r(zeta) = 0; the wire commitments equal [p(s)]G with p = iNTT(wire column) evaluated by the C oracle (the reference's
    test identity, kzg/src/lib.rs:102-105); every opening witness satisfies the verifier's equation in trapdoor form,
    (s - z) W = C - y G (kzg/src/lib.rs:66-81 with the pairing replaced by the known secret) -- for [r] with the
    verifier's own linearisation commitment (proof.rs:441-503) built from the proof's and the circuit's commitments;
    the batched witness is the v-combination of the six-opening proof's witnesses.  (The pairing verifier itself runs
    on a 2^22 proof in tests/test_host_mirror.py.)"""
    from oracle import coracle as CO
    from oracle import pairing as PR
    from typlonk_amd.circuits import SquaringChain

    log_n = 22
    n = 1 << log_n
    secret = 0x0123456789ABCDEF0123456789ABCDEF
    s_l = _limbs(secret)
    chain = SquaringChain(ctx, log_n, keep_host=True)
    sid = ctx.srs_generate(s_l, n + 3)
    ctx.srs_precompute(sid, 20)
    alpha, beta, gamma = CH
    v = 0x1F2E3D4C5B6A7988
    chal = (lambda c: (_limbs(beta), _limbs(gamma)), lambda c: (_limbs(alpha), _limbs(ZETA)))
    six = ctx.prove(sid, chain.circuit, chain.wire_evals, chain.pi_evals, chain.cosets, *chal)
    bat = ctx.prove(sid, chain.circuit, chain.wire_evals, chain.pi_evals, chain.cosets, *chal, challenge_v=lambda e: _limbs(v))
    pt = lambda t: g1_unpack_one(t[0], t[1])                   # noqa: E731
    fr = lambda a: O.fr_from_mont_limbs([int(x) for x in a])   # noqa: E731
    ev = [fr(e) for e in six["evals"]]
    assert ev[5] == 0 and [fr(e) for e in bat["evals"]] == ev
    host = chain.host_inputs()

    def at_s(evals):      # p(s) for p = interpolate(evals)
        coeffs = CO.ntt(evals, log_n, inverse=True, threads=0)
        return coeffs, fr(CO.poly_eval(coeffs, s_l))

    commits = [pt(c) for c in six["commit"]]
    for i in range(3):
        _, ps = at_s(host["wires"][i])
        assert commits[i] == O.g1_mul(O.G1, ps), f"wire commitment {i}"
    assert [pt(c) for c in bat["commit"]] == commits and pt(bat["z_commit"]) == pt(six["z_commit"])
    assert [pt(c) for c in bat["t_commit"]] == [pt(c) for c in six["t_commit"]]
    # openings in trapdoor form
    w = O.domain_root(log_n)
    wit = [pt(x) for x in six["witness"]]
    zc = pt(six["z_commit"])
    trapdoor = lambda W, C, z, y: O.g1_mul(W, (secret - z) % O.R) == O.g1_add(C, O.g1_neg(O.g1_mul(O.G1, y)))   # noqa: E731
    for i in range(3):
        assert trapdoor(wit[i], commits[i], ZETA, ev[i]), f"opening {i}"
    assert trapdoor(wit[3], zc, ZETA, ev[3]) and trapdoor(wit[4], zc, ZETA * w % O.R, ev[4])
    # [r]: the verifier's linearisation commitment from commitments only
    sel_c, sel_s = at_s(host["selectors"][2])                  # q_o = q_m; q_l = q_r = q_c = 0
    q_pt = O.g1_mul(O.G1, sel_s)
    fixed = [None, None, q_pt, q_pt, None]
    sig = [at_s(x) for x in host["sigma"]]
    sigma_c = [O.g1_mul(O.G1, ps) for _, ps in sig]
    sigma_ev = [fr(CO.poly_eval(cf, _limbs(ZETA))) for cf, _ in sig]
    r_commit = PR.linearisation_commitment(log_n, fixed, sigma_c, sigma_ev, PO.COSETS, ev[:3], zc, (ev[3], ev[4]), ZETA,
                                           [pt(c) for c in six["t_commit"]], (alpha, beta, gamma), 0)
    assert trapdoor(wit[5], r_commit, ZETA, 0), "opening of r"
    # batched shape: W = W_a + v W_b + v^2 W_c + v^3 W_Z + v^4 W_r (division by X - zeta is linear), W_zw unchanged
    bw = [pt(x) for x in bat["witness"]]
    comb = None
    for k, i in enumerate((0, 1, 2, 3, 5)):
        comb = O.g1_add(comb, O.g1_mul(wit[i], pow(v, k, O.R)))
    assert bw[0] == comb and bw[1] == wit[4]
    chain.free()
    ctx.srs_free(sid)
