"""GPU parity of typlonk_quotient_dev against the reference's quotient_polynomial
(plonk/src/proof.rs:292-375) restated with its schoolbook products in oracle/plonk_oracle.py."""
import numpy as np
import pytest

from helpers import O, fr_pack, fr_unpack
from oracle import plonk_oracle as PO

pytestmark = pytest.mark.gpu

ALPHA, BETA, GAMMA = 0x1234567DEADBEEF, 0xABCDEF0123456789ABCDEF, 0x55AA55AA77


def _limbs(v):
    return np.array(O.fr_to_mont_limbs(v), dtype=np.uint64)


def _upload(ctx, coeffs, n):
    b = ctx.alloc(n)
    b.upload(fr_pack(list(coeffs) + [0] * (n - len(coeffs))))
    return b


def _run(ctx, r, log_n):
    n = 1 << log_n
    wires = [_upload(ctx, w, n) for w in r["wires"]]
    z = _upload(ctx, r["z"], n)
    sel = [_upload(ctx, r["q"][k], n) for k in ("q_l", "q_r", "q_o", "q_m", "q_c")]
    sig = [_upload(ctx, s, n) for s in r["sigma"]]
    pi = _upload(ctx, r["pi"], n)
    t_out = ctx.alloc(4 * n)
    ctx.quotient_dev(log_n, wires, z, sel, sig, pi, _limbs(ALPHA), _limbs(BETA), _limbs(GAMMA),
                     [_limbs(k) for k in PO.COSETS], t_out)
    t = fr_unpack(t_out.download())
    for b in wires + [z] + sel + sig + [pi, t_out]:
        b.free()
    return t


@pytest.mark.parametrize("log_n", [2, 3, 4, 6, 8])
def test_quotient_equals_reference_schoolbook(ctx, log_n):
    r = PO.prove_round_2_3(log_n, ALPHA, BETA, GAMMA)
    assert r["rem"] == []                       # valid witness: the reference's division is exact
    n = r["n"]
    t = _run(ctx, r, log_n)
    assert len(t) == 4 * n
    assert O.poly_trim(t) == r["t"]             # bit-exact, all 3n - 3 coefficients
    assert not any(t[3 * n - 3:])               # nothing above degree 3n - 4
    # SlicedPoly::<3>::from_poly slices (the three MSM inputs of proof.rs:181)
    assert PO.slices(t, n) == PO.slices(r["t"], n)


def test_quotient_with_nonzero_public_inputs_and_other_seed(ctx):
    """a PI polynomial that vanishes nowhere breaks divisibility, so use PI = 0 on the gate rows but a
    different witness seed; also checks the Z(wX) index shift against the oracle's rotated interpolation"""
    r = PO.prove_round_2_3(5, ALPHA, BETA, GAMMA, x0=0x1337)
    t = _run(ctx, r, 5)
    assert O.poly_trim(t) == r["t"]
    w = O.domain_root(5)
    assert O.poly_eval(r["zw"], 12345) == O.poly_eval(r["z"], 12345 * w % O.R)


def test_quotient_identity_at_2_12(ctx):
    """size-independent property: t(x) (x^n - 1) == numerator(x) at random points, n = 2^12 (the
    schoolbook oracle would need 12 products of 4096 x 4096 terms; Horner evaluations are enough)"""
    log_n = 12
    n = 1 << log_n
    _, cols, q_evals, perm = PO.squaring_chain(log_n, x0=5)
    ids, sig = PO.compile_permutation(perm, n, log_n)
    acc = PO.grand_product(cols, ids, sig, BETA, GAMMA, n)
    assert acc[n] == 1
    r = {"wires": [O.interpolate(c, log_n) for c in cols], "z": O.interpolate(acc[:n], log_n),
         "q": {k: O.interpolate(v, log_n) for k, v in q_evals.items()},
         "sigma": [O.interpolate(s, log_n) for s in sig], "pi": []}
    t = _run(ctx, r, log_n)
    assert not any(t[3 * n - 3:])
    wroot = O.domain_root(log_n)
    for x in (0x1234567, 0xFEDCBA9876543210FEDCBA):
        ev = lambda p: O.poly_eval(p, x)  # noqa: E731
        a, b, c = (ev(p) for p in r["wires"])
        z, zw = ev(r["z"]), O.poly_eval(r["z"], x * wroot % O.R)
        q = {k: ev(v) for k, v in r["q"].items()}
        s = [ev(p) for p in r["sigma"]]
        line1 = q["q_l"] * a + q["q_r"] * b - q["q_o"] * c + q["q_m"] * a * b + q["q_c"]
        line2 = (a + BETA * 2 * x + GAMMA) * (b + BETA * 3 * x + GAMMA) * (c + BETA * 4 * x + GAMMA) * z
        line3 = (a + BETA * s[0] + GAMMA) * (b + BETA * s[1] + GAMMA) * (c + BETA * s[2] + GAMMA) * zw
        zh = pow(x, n, O.R) - 1
        l0 = zh * pow(n * (x - 1), -1, O.R)
        num = (line1 + ALPHA * (line2 - line3) + ALPHA * ALPHA * (z - 1) * l0) % O.R
        assert O.poly_eval(t, x) * zh % O.R == num


def test_quotient_with_cached_circuit(ctx):
    """typlonk_circuit_load caches the per-circuit coset evaluations; two proofs (different witnesses,
    different challenges) against one loaded circuit equal the uncached results"""
    log_n, n = 6, 64
    r1 = PO.prove_round_2_3(log_n, ALPHA, BETA, GAMMA, x0=3)
    sel = [_upload(ctx, r1["q"][k], n) for k in ("q_l", "q_r", "q_o", "q_m", "q_c")]
    sig = [_upload(ctx, s, n) for s in r1["sigma"]]
    cid = ctx.circuit_load(log_n, sel, sig)
    for b in sel + sig:
        b.free()       # the circuit keeps its own transformed copies
    for x0, (al, be, ga) in ((3, (ALPHA, BETA, GAMMA)), (0x77, (5, 6, 7))):
        r = PO.prove_round_2_3(log_n, al, be, ga, x0=x0)
        wires = [_upload(ctx, w, n) for w in r["wires"]]
        z, pi, t_out = _upload(ctx, r["z"], n), _upload(ctx, r["pi"], n), ctx.alloc(4 * n)
        ctx.quotient_dev(log_n, wires, z, None, None, pi, _limbs(al), _limbs(be), _limbs(ga),
                         [_limbs(k) for k in PO.COSETS], t_out, circuit=cid)
        assert O.poly_trim(fr_unpack(t_out.download())) == r["t"]
        for b in wires + [z, pi, t_out]:
            b.free()
    ctx.circuit_free(cid)


def test_quotient_argument_errors(ctx):
    from typlonk_amd.capi import TyplonkError, ERR_RANGE

    r = PO.prove_round_2_3(3, ALPHA, BETA, GAMMA)
    n = 8
    bufs = [_upload(ctx, [1], n) for _ in range(13)]
    small = ctx.alloc(3 * n)
    with pytest.raises(TyplonkError) as e:
        ctx.quotient_dev(3, bufs[0:3], bufs[3], bufs[4:9], bufs[9:12], bufs[12], _limbs(1), _limbs(1), _limbs(1),
                         [_limbs(k) for k in PO.COSETS], small)
    assert e.value.code == ERR_RANGE
    for b in bufs + [small]:
        b.free()
