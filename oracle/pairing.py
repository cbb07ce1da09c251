"""CPU oracle (TEST INFRASTRUCTURE ONLY): BLS12-381 pairing and the reference's two verifiers.

  * kzg::KzgScheme::verify                 /root/reference/kzg/src/lib.rs:66-81
        pairing(W, [s]G2 - z G2) == pairing(C - y G1, G2)
  * Srs::g2                                /root/reference/kzg/src/srs.rs:26-34
  * plonk::proof::verify / verify_openings / linearisation_commitment
                                           /root/reference/plonk/src/proof.rs:195-281, 441-503
  * SlicedPoly::compact_commitment         /root/reference/plonk/src/utils.rs:96-109
  * CompiledPermutation::sigma_evals / sigma_commitments   /root/reference/permutation/src/lib.rs:165-194

The pairing itself lives in ark-ec 0.3.0 / ark-bls12-381 0.3.0 (Cargo.lock:17-18, 28-29), which are not in
this container: it is restated from the published definition -- optimal ate pairing, Miller loop over
|x| = 0xd201000000010000, M-type sextic twist E': y^2 = x^3 + 4(1 + u), final exponentiation (p^12 - 1)/r --
with Fq12 = Fq[w]/(w^12 - 2 w^6 + 2) (w^6 = 1 + u).  Parity unpinned against the reference (no Rust here);
pinned by what a pairing must satisfy and tests/test_oracle.py checks: the G2 generator is on E' and has order
r, e(aP, bQ) == e(P, Q)^(ab), e(P, Q) != 1, e(P, Q)^r == 1.  The verifier's answer does not depend on the
normalisation of e (any non-degenerate bilinear map gives the same accept/reject).
Fiat-Shamir (verify_challenges, proof.rs:236-246) is out of scope: the caller supplies the challenges.
"""
from __future__ import annotations

from . import bls12_381 as O
from . import plonk_oracle as PO

P, R = O.P, O.R
ATE_LOOP = 0xD201000000010000  # |x|; x = -ATE_LOOP

# ---- Fq2 = Fq[u]/(u^2 + 1), elements (a, b) = a + b u ---------------------------------------------------
def f2_add(x, y): return ((x[0] + y[0]) % P, (x[1] + y[1]) % P)
def f2_sub(x, y): return ((x[0] - y[0]) % P, (x[1] - y[1]) % P)
def f2_neg(x): return (-x[0] % P, -x[1] % P)
def f2_mul(x, y): return ((x[0] * y[0] - x[1] * y[1]) % P, (x[0] * y[1] + x[1] * y[0]) % P)
def f2_scalar(x, k): return (x[0] * k % P, x[1] * k % P)


def f2_inv(x):
    d = pow(x[0] * x[0] + x[1] * x[1], P - 2, P)
    return (x[0] * d % P, -x[1] * d % P)


F2_ZERO, F2_ONE = (0, 0), (1, 0)
B2 = (4, 4)  # 4 (1 + u)

# G2 generator (ark-bls12-381 g2::G2_GENERATOR_X/Y, the standard BLS12-381 generator)
G2 = ((0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
       0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E),
      (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
       0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE))


# ---- E'(Fq2), affine, None = identity -----------------------------------------------------------------
def g2_is_on_curve(q):
    if q is None:
        return True
    x, y = q
    return f2_mul(y, y) == f2_add(f2_mul(f2_mul(x, x), x), B2)


def g2_neg(q):
    return None if q is None else (q[0], f2_neg(q[1]))


def g2_add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    if a[0] == b[0]:
        if a[1] != b[1] or a[1] == F2_ZERO:
            return None
        lam = f2_mul(f2_scalar(f2_mul(a[0], a[0]), 3), f2_inv(f2_scalar(a[1], 2)))
    else:
        lam = f2_mul(f2_sub(b[1], a[1]), f2_inv(f2_sub(b[0], a[0])))
    x3 = f2_sub(f2_sub(f2_mul(lam, lam), a[0]), b[0])
    return (x3, f2_sub(f2_mul(lam, f2_sub(a[0], x3)), a[1]))


def g2_mul(q, k: int):
    k %= R
    acc = None
    for bit in bin(k)[2:] if k else "":
        acc = g2_add(acc, acc)
        if bit == "1":
            acc = g2_add(acc, q)
    return acc


# ---- Fq12 = Fq[w]/(w^12 - 2 w^6 + 2): lists of 12 coefficients ------------------------------------------
def f12_one():
    return [1] + [0] * 11


def f12_mul(a, b):
    t = [0] * 23
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                if y:
                    t[i + j] += x * y
    for i in range(22, 11, -1):  # w^12 = 2 w^6 - 2
        c = t[i]
        if c:
            t[i - 6] += 2 * c
            t[i - 12] -= 2 * c
    return [x % P for x in t[:12]]


def f12_pow(a, e: int):
    acc = f12_one()
    for bit in bin(e)[2:]:
        acc = f12_mul(acc, acc)
        if bit == "1":
            acc = f12_mul(acc, a)
    return acc


def f12_conj(a):
    """w -> -w: the Fq6-conjugate; the inverse of a unitary element (anything after the final exponentiation)"""
    return [(-x % P) if i & 1 else x for i, x in enumerate(a)]


def _line(lam, xt, yt, px, py):
    """w^3 * l(P) for the line of slope lam/w through the untwisted psi(T) = (xT / w^2, yT / w^3):
    (lam xT - yT) - lam xP w^2 + yP w^3, with a + b u = (a - b) + b w^6.  The factor w^3 lies in Fq4 and dies in the
    final exponentiation."""
    c0 = f2_sub(f2_mul(lam, xt), yt)
    c2 = f2_scalar(lam, -px % P)
    out = [0] * 12
    out[0], out[6] = (c0[0] - c0[1]) % P, c0[1]
    out[2], out[8] = (c2[0] - c2[1]) % P, c2[1]
    out[3] = py % P
    return out


def miller_loop(p1, q2):
    if p1 is None or q2 is None:
        return f12_one()
    px, py = p1
    f = f12_one()
    t = q2
    for bit in bin(ATE_LOOP)[3:]:
        lam = f2_mul(f2_scalar(f2_mul(t[0], t[0]), 3), f2_inv(f2_scalar(t[1], 2)))
        f = f12_mul(f12_mul(f, f), _line(lam, t[0], t[1], px, py))
        t = g2_add(t, t)
        if bit == "1":
            lam = f2_mul(f2_sub(q2[1], t[1]), f2_inv(f2_sub(q2[0], t[0])))
            f = f12_mul(f, _line(lam, t[0], t[1], px, py))
            t = g2_add(t, q2)
    return f


FINAL_EXP = (P ** 12 - 1) // R


def pairing(p1, q2):
    """e(P, Q), P in G1 (affine ints or None), Q in G2 (affine Fq2 pairs or None); x < 0 -> conjugate"""
    return f12_conj(f12_pow(miller_loop(p1, q2), FINAL_EXP))


# ---- the reference's verifiers ---------------------------------------------------------------------------
def srs_g2(s: int):
    """Srs::g2: (G2, [s]G2)"""
    return G2, g2_mul(G2, s)


def kzg_verify(commitment, opening, z: int, g2, g2s) -> bool:
    """KzgScheme::verify, kzg/src/lib.rs:66-81; opening = (W, y)"""
    w, y = opening
    a = g2_add(g2s, g2_neg(g2_mul(g2, z)))
    b = O.g1_add(commitment, O.g1_neg(O.g1_mul(O.G1, y)))
    return pairing(w, a) == pairing(b, g2)


def g1_lincomb(terms):
    acc = None
    for pt, k in terms:
        acc = O.g1_add(acc, O.g1_mul(pt, k % R))
    return acc


def linearisation_commitment(log_n, fixed_commitments, sigma_commitments, sigma_evals, cosets, advice, acc, acc_evals,
                             zeta, quotient, challenges, public_eval):
    """plonk/src/proof.rs:441-503.  fixed_commitments = [q_l, q_r, q_o, q_m, q_c]; quotient = [t_lo, t_mid, t_hi]"""
    n = 1 << log_n
    alpha, beta, gamma = challenges
    a, b, c = advice
    q_l, q_r, q_o, q_m, q_c = fixed_commitments
    line1 = g1_lincomb([(q_l, a), (q_r, b), (q_o, -c), (q_m, a * b), (q_c, 1)])
    l2 = 1
    for k, ev in zip(cosets, advice):
        l2 = l2 * (ev + beta * k * zeta + gamma) % R
    l0_eval = O.poly_eval(PO.l0_poly(n), zeta)
    line2 = O.g1_mul(acc, (l2 * alpha + l0_eval * alpha * alpha) % R)
    l3 = 1
    for s_ev, ev in list(zip(sigma_evals, advice))[:2]:
        l3 = l3 * (ev + beta * s_ev + gamma) % R
    line3 = O.g1_mul(sigma_commitments[2], l3 * alpha * beta * acc_evals[1] % R)
    zn = pow(zeta, n, R)
    compact = g1_lincomb([(quotient[0], 1), (quotient[1], zn), (quotient[2], zn * zn)])   # utils.rs:96-109
    vanish = (zn - 1) % R
    line5 = O.g1_mul(compact, vanish)
    constant = (alpha * (l3 * (c + gamma) % R * acc_evals[1]) + l0_eval * alpha * alpha + public_eval) % R
    inner = O.g1_add(line3, O.g1_mul(O.G1, constant))
    return O.g1_add(O.g1_add(line1, O.g1_add(line2, O.g1_neg(inner))), O.g1_neg(line5))


def plonk_verify(log_n, proof, fixed_commitments, sigma_polys, sigma_commitments, cosets, pi_evals, challenges, zeta,
                 g2, g2s) -> bool:
    """plonk/src/proof.rs:195-234 with the challenges supplied.  proof: dict with commit[3], open[3] = (W, y),
    z_commit, z_open, zw_open, t_commit[3], r_open -- the shape oracle/plonk_oracle.prove returns."""
    w = O.domain_root(log_n)
    public_eval = O.poly_eval(O.interpolate(pi_evals, log_n), zeta)
    for cm, op in zip(proof["commit"], proof["open"]):                      # verify_openings, :247-272
        if not kzg_verify(cm, op, zeta, g2, g2s):
            return False
    if not kzg_verify(proof["z_commit"], proof["z_open"], zeta, g2, g2s):
        return False
    if not kzg_verify(proof["z_commit"], proof["zw_open"], zeta * w % R, g2, g2s):
        return False
    advice = [op[1] for op in proof["open"]]
    sigma_evals = [O.poly_eval(s, zeta) for s in sigma_polys]                # permutation/src/lib.rs:165-176
    r = linearisation_commitment(log_n, fixed_commitments, sigma_commitments, sigma_evals, cosets, advice,
                                 proof["z_commit"], [proof["z_open"][1], proof["zw_open"][1]], zeta, proof["t_commit"],
                                 challenges, public_eval)
    return kzg_verify(r, proof["r_open"], zeta, g2, g2s) and proof["r_open"][1] == 0
