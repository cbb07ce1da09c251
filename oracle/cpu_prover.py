"""A "fair CPU" prove() -- TEST INFRASTRUCTURE / bench.py's cpu_fair leg only (the product never imports oracle/).

plonk::proof::prove (/root/reference/plonk/src/proof.rs:96-194) as a competent CPU implementation would run it on all
host cores: the same rounds, formulas and outputs as the reference, with its asymptotically slow steps replaced by what
the GPU path also uses --

  * the quotient through coset NTTs on the 4n domain instead of twelve schoolbook products (proof.rs:292-375),
  * one batch inversion per chunk instead of 3n field divisions in the grand product (permutation/src/proving.rs:7-31),
  * bucket-method MSMs (oracle_msm_pippenger) instead of a 255-step double-and-add per term (kzg/src/lib.rs:41-54),
  * Z(wX) as an index shift on the coset instead of a second interpolation (proof.rs:121-126).

Everything heavy is an OpenMP loop in oracle/typlonk_oracle.c; this module only sequences the calls and computes the
handful of scalars of the linearisation polynomial (proof.rs:376-439) with Python integers.  Its output equals the
schoolbook oracle's (oracle/plonk_oracle.py, tests/test_oracle.py) and the GPU prover's bit for bit, which is what makes
its time a baseline for the same job.
"""
from __future__ import annotations

import ctypes as C
import time

import numpy as np

from . import bls12_381 as O
from . import coracle as CO

R = O.R


def _limbs(v: int) -> np.ndarray:
    return np.array(O.fr_to_mont_limbs(v % R), dtype=np.uint64)


def _int(l) -> int:
    return O.fr_from_mont_limbs([int(x) for x in np.asarray(l).reshape(4)])


def _p64(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def _ptrs(arrs):
    return (C.POINTER(C.c_uint64) * len(arrs))(*[_p64(a) for a in arrs])


def lincomb(polys, scalars, n, constant=None):
    polys = [np.ascontiguousarray(p, dtype=np.uint64).reshape(-1, 4) for p in polys]
    sc = np.ascontiguousarray(np.stack([_limbs(s) for s in scalars]), dtype=np.uint64)
    out = np.zeros((n, 4), dtype=np.uint64)
    cst = _limbs(constant) if constant is not None else None
    CO.lib().oracle_fr_lincomb(_ptrs(polys), _p64(sc), C.c_size_t(len(polys)), C.c_size_t(n), _p64(cst) if cst is not None else None,
                               _p64(out))
    return out


def quotient(log_n: int, wires, z, pi, sel, sig, alpha: int, beta: int, gamma: int, ks, threads: int = 0, stage=None):
    """t(X) as plonk::proof::quotient_polynomial defines it (proof.rs:292-375), computed the way the GPU path computes it:
    every input polynomial (n coefficients, Montgomery limbs) evaluated on the coset 7 H_4n, the formula applied point
    by point with the division by X^n - 1, one inverse coset transform.  Returns the 4n coefficients (the top n are zero
    exactly when the numerator vanishes on H, i.e. for a satisfying witness).  alpha / beta / gamma / ks: integers."""
    lib = CO.lib()
    n = 1 << log_n
    stage = {} if stage is None else stage
    t0 = time.perf_counter()

    def mark(name):
        nonlocal t0
        t = time.perf_counter()
        stage[name] = stage.get(name, 0.0) + (t - t0) * 1e3
        t0 = t

    k_l = np.ascontiguousarray(np.stack([_limbs(k) for k in ks]))
    l0 = np.tile(_limbs(pow(n, -1, R)), (n, 1))                                                         # utils.rs:150-159
    g7 = _limbs(7)

    def extend(p):
        buf = np.zeros((4 * n, 4), dtype=np.uint64)
        buf[:n] = np.asarray(p, dtype=np.uint64).reshape(n, 4)
        return CO.ntt(buf, log_n + 2, coset=g7, threads=threads)

    ext = [extend(p) for p in list(wires) + [z, pi] + list(sel) + list(sig) + [l0]]
    mark("ntt")
    t_ev = np.zeros((4 * n, 4), dtype=np.uint64)
    rc = lib.oracle_quotient_pointwise(_ptrs(ext), C.c_uint32(log_n), _p64(_limbs(alpha)), _p64(_limbs(beta)), _p64(_limbs(gamma)),
                                       _p64(k_l), _p64(g7), _p64(t_ev))
    assert rc == 0
    del ext
    mark("quotient_pointwise")
    t = CO.ntt(t_ev, log_n + 2, inverse=True, coset=g7, threads=threads)
    mark("ntt")
    return t


def prove(log_n: int, inputs: dict, srs_xy, srs_inf, challenges, threads: int = 0):
    """inputs = SquaringChain.host_inputs() (wire / selector / sigma EVALUATIONS, Montgomery limbs; public inputs [0]);
    challenges = [beta, gamma, alpha, zeta] as limb arrays (what bench.py injects into the GPU prover as well).
    Returns the proof in Context.prove's layout plus per-stage wall times."""
    lib = CO.lib()
    n = 1 << log_n
    beta, gamma, alpha, zeta = [_int(c) for c in challenges]
    ks = [int(k) for k in inputs["cosets"]]
    k_l = np.ascontiguousarray(np.stack([_limbs(k) for k in ks]))
    srs_xy = np.ascontiguousarray(srs_xy, dtype=np.uint64).reshape(-1, 12)
    srs_inf = None if srs_inf is None else np.ascontiguousarray(srs_inf, dtype=np.uint8)
    stage, t0 = {}, time.perf_counter()
    nthreads = [1]

    def mark(name):
        nonlocal t0
        t = time.perf_counter()
        stage[name] = stage.get(name, 0.0) + (t - t0) * 1e3
        t0 = t

    def intt(ev):
        return CO.ntt(ev, log_n, inverse=True, threads=threads)

    def commit(coeffs, m):
        xy, inf, _, thr = CO.msm_pippenger(coeffs[:m], srs_xy[:m], None if srs_inf is None else srs_inf[:m],
                                           c=max(8, min(13, log_n - 5)))   # tools/cpu_msm_sweep.py
        nthreads[0] = max(nthreads[0], thr)
        return xy, inf

    # ---- round 1 (:96-110): interpolate the wire columns, commit ---------------------------------------------------
    wires_ev = [np.ascontiguousarray(w, dtype=np.uint64).reshape(n, 4) for w in inputs["wires"]]
    wires = [intt(w) for w in wires_ev]
    mark("ntt")
    commits = [commit(w, n) for w in wires]
    mark("msm")
    # ---- round 2 (:113-129): grand product, [Z] --------------------------------------------------------------------
    sig_ev = [np.ascontiguousarray(s, dtype=np.uint64).reshape(n, 4) for s in inputs["sigma"]]
    z_ev = np.zeros((n, 4), dtype=np.uint64)
    last = np.zeros(4, dtype=np.uint64)
    rc = lib.oracle_grand_product(_ptrs(wires_ev), _ptrs(sig_ev), _p64(_limbs(beta)), _p64(_limbs(gamma)), _p64(k_l), C.c_uint32(log_n),
                                  _p64(z_ev), _p64(last))
    assert rc == 0 and _int(last) == 1, "copy constraints not satisfied"
    mark("grand_product")
    z = intt(z_ev)
    mark("ntt")
    z_commit = commit(z, n)
    mark("msm")
    # ---- round 3: quotient on the coset 7 H_4n (:292-375) ----------------------------------------------------------
    sel = [intt(np.ascontiguousarray(s, dtype=np.uint64).reshape(n, 4)) for s in inputs["selectors"]]   # builder.rs:84-88
    sig = [intt(s) for s in sig_ev]                                                                     # proof.rs:334-338
    mark("ntt")
    pi = np.zeros((n, 4), dtype=np.uint64)                                                              # public inputs [0]
    t = quotient(log_n, wires, z, pi, sel, sig, alpha, beta, gamma, ks, threads=threads, stage=stage)
    t0 = time.perf_counter()
    assert not t[3 * n:].any(), "the quotient has degree >= 3n: the witness does not satisfy the circuit"
    t_sl = [t[0:n], t[n:2 * n], t[2 * n:3 * n]]
    # ---- openings (:147-175; kzg/src/lib.rs:55-64) -------------------------------------------------------------------
    w = O.domain_root(log_n)
    zl, zwl = _limbs(zeta), _limbs(zeta * w % R)

    def open_(p, at):
        q, y = CO.poly_div_linear(p, at)
        return q, _int(y)

    qa, qb, qc = [open_(p, zl) for p in wires]
    a_e, b_e, c_e = qa[1], qb[1], qc[1]
    qz, qzw = open_(z, zl), open_(z, zwl)
    s0, s1 = _int(CO.poly_eval(sig[0], zl)), _int(CO.poly_eval(sig[1], zl))
    zn = pow(zeta, n, R)
    zh = (zn - 1) % R
    l0z = zh * pow(n * (zeta - 1) % R, -1, R) % R if zeta != 1 else 1
    l2 = 1
    for e, k in zip((a_e, b_e, c_e), ks):
        l2 = l2 * (e + k * beta * zeta + gamma) % R
    abz = (a_e + beta * s0 + gamma) * (b_e + beta * s1 + gamma) % R * qzw[1] % R
    # r (:376-439) as one linear combination; PI(zeta) = 0
    r = lincomb([sel[0], sel[1], sel[2], sel[3], sel[4], z, sig[2], t_sl[0], t_sl[1], t_sl[2]],
                [a_e, b_e, -c_e, a_e * b_e, 1, alpha * l2 + alpha * alpha * l0z, -alpha * beta * abz, -zh, -zh * zn, -zh * zn * zn],
                n, constant=-(alpha * (gamma + c_e) * abz) - alpha * alpha * l0z)
    qr = open_(r, zl)
    mark("openings")
    witness = [commit(q, n - 1) for q, _ in (qa, qb, qc, qz, qzw, qr)]
    t_commit = [commit(t_sl[0], n), commit(t_sl[1], n), commit(t_sl[2], max(n - 3, 0))]          # :181
    mark("msm")
    evals = [_limbs(v) for v in (a_e, b_e, c_e, qz[1], qzw[1], qr[1])]
    return {"commit": commits, "z_commit": z_commit, "t_commit": t_commit, "witness": witness, "evals": evals,
            "stage_ms": {k: round(v, 2) for k, v in stage.items()}, "threads": nthreads[0]}
