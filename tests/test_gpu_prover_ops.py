"""GPU parity of the O(n) prover steps (grand product, open, linear combinations) against the
reference-faithful oracle (oracle/plonk_oracle.py, oracle/bls12_381.py)."""
import numpy as np
import pytest

from helpers import O, fr_pack, fr_unpack
from oracle import plonk_oracle as PO

pytestmark = pytest.mark.gpu

BETA, GAMMA = 0xABCDEF0123456789ABCDEF, 0x55AA55AA77


def _limbs(v):
    return np.array(O.fr_to_mont_limbs(v), dtype=np.uint64)


def _up(ctx, vals):
    b = ctx.alloc(len(vals))
    b.upload(fr_pack(vals))
    return b


@pytest.mark.parametrize("log_n", [1, 3, 6, 11, 13])
def test_grand_product_equals_reference_scan(ctx, log_n):
    """CompiledPermutation::prove (one division per cell, sequential) vs prefix/suffix product scans"""
    n = 1 << log_n
    if log_n >= 2:
        _, cols, _, perm = PO.squaring_chain(log_n, x0=7)
    else:
        cols, perm = [[5, 6], [7, 8], [9, 10]], list(range(6))
    ids, sig = PO.compile_permutation(perm, n, log_n)
    ref = PO.grand_product(cols, ids, sig, BETA, GAMMA, n)
    wires = [_up(ctx, c) for c in cols]
    sigma = [_up(ctx, s) for s in sig]
    z = ctx.alloc(n)
    ctx.grand_product_dev(log_n, wires, sigma, _limbs(BETA), _limbs(GAMMA), [_limbs(k) for k in PO.COSETS], z)
    got = fr_unpack(z.download())
    assert got == ref[:n]
    assert got[0] == 1
    if log_n >= 2:
        assert ref[n] == 1          # valid copy constraints: the product closes
    # an invalid witness (one cell changed) still matches the reference scan value for value
    cols[0][0] = (cols[0][0] + 1) % O.R
    ref2 = PO.grand_product(cols, ids, sig, BETA, GAMMA, n)
    wires[0].upload(fr_pack(cols[0]))
    ctx.grand_product_dev(log_n, wires, sigma, _limbs(BETA), _limbs(GAMMA), [_limbs(k) for k in PO.COSETS], z)
    assert fr_unpack(z.download()) == ref2[:n]
    for b in wires + sigma + [z]:
        b.free()


@pytest.mark.parametrize("m", [1, 2, 7, 8, 9, 255, 2048, 2049, 5000, (1 << 16) - 1, 1 << 16])
def test_open_equals_reference_horner_and_division(ctx, m):
    """kzg/src/lib.rs:55-61: y = p(z), q = (p - y) / (X - z)"""
    p = O.random_frs(0x0BE2 + m, m)
    z = O.random_frs(77, 1)[0]
    q_ref, y_ref = O.poly_div_linear(p, z)
    pb = _up(ctx, p)
    qb = ctx.alloc(max(m - 1, 1))
    y = ctx.open_dev(pb, m, _limbs(z), qb)
    assert O.fr_from_mont_limbs([int(v) for v in y]) == y_ref == O.poly_eval(p, z)
    if m > 1:
        got = fr_unpack(qb.download(0, m - 1))
        assert O.poly_trim(got) == q_ref
    # evaluation only (no quotient buffer), and an offset sub-range
    y2 = ctx.open_dev(pb, m, _limbs(z))
    assert (y2 == y).all()
    if m > 3:
        y3 = ctx.open_dev(pb, m - 3, _limbs(z), offset=2)
        assert O.fr_from_mont_limbs([int(v) for v in y3]) == O.poly_eval(p[2:m - 1], z)
    pb.free()
    qb.free()


def test_open_kat_and_errors(ctx):
    """the reference's own open in kzg::commit: p = 1 + 2X + 3X^2 at z = 1 -> y = 6, q = 5 + 3X"""
    from typlonk_amd.capi import TyplonkError, ERR_LENGTH

    pb = _up(ctx, [1, 2, 3])
    qb = ctx.alloc(2)
    y = ctx.open_dev(pb, 3, _limbs(1), qb)
    assert O.fr_from_mont_limbs([int(v) for v in y]) == 6 and fr_unpack(qb.download()) == [5, 3]
    with pytest.raises(TyplonkError) as e:     # `.expect("at least 1")`
        ctx.open_dev(pb, 0, _limbs(1))
    assert e.value.code == ERR_LENGTH
    pb.free()
    qb.free()


def test_lincomb(ctx):
    n = 1000
    polys = [O.random_frs(900 + k, n) for k in range(5)]
    scal = O.random_frs(55, 5)
    const = 0x1234
    bufs = [_up(ctx, p) for p in polys]
    out = ctx.alloc(n)
    ctx.lincomb_dev(bufs, [_limbs(s) for s in scal], n, out, constant=_limbs(const))
    exp = [sum(s * p[i] for s, p in zip(scal, polys)) % O.R for i in range(n)]
    exp[0] = (exp[0] + const) % O.R
    assert fr_unpack(out.download()) == exp
    for b in bufs + [out]:
        b.free()
