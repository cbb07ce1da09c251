"""CPU oracle (TEST INFRASTRUCTURE ONLY) for the quotient polynomial of TyPLONK's prover.

Reference-faithful Python big-int restatement of

  * plonk::builder  gate rows / padding            /root/reference/plonk/src/builder.rs:47-58, 317-325
  * permutation::Permutation::compile              /root/reference/permutation/src/lib.rs:101-128, 141-154
  * permutation::CompiledPermutation::prove        /root/reference/permutation/src/proving.rs:7-31
  * Z / Z(wX) construction in prove()              /root/reference/plonk/src/proof.rs:113-128
  * plonk::proof::quotient_polynomial              /root/reference/plonk/src/proof.rs:292-375
    (12 schoolbook `naive_mul` products, divide_by_vanishing_poly, SlicedPoly::from_poly)
  * plonk::utils::l0_poly                          /root/reference/plonk/src/utils.rs:150-159

Same pinning status as oracle/bls12_381.py: the reference cannot be run here, so this module is
pinned only by the algebraic identities the reference itself asserts (vanishes(line1), vanishes(line4),
the row-1 recurrence assert at proof.rs:350-353, remainder == 0) -- all checked in tests/test_oracle.py.
Only tests/ may import it.
"""
from __future__ import annotations

from . import bls12_381 as O

R = O.R


# ---- dense polynomials over Fr (ark-poly DensePolynomial semantics) ------------------------------------
def trim(c):
    return O.poly_trim(c)


def p_add(a, b):
    n = max(len(a), len(b))
    return trim([((a[i] if i < len(a) else 0) + (b[i] if i < len(b) else 0)) % R for i in range(n)])


def p_sub(a, b):
    n = max(len(a), len(b))
    return trim([((a[i] if i < len(a) else 0) - (b[i] if i < len(b) else 0)) % R for i in range(n)])


def p_scale(a, k):
    return trim([x * k % R for x in a])


def naive_mul(a, b):
    """DensePolynomial::naive_mul: result[i + j] += a[i] * b[j]"""
    if not a or not b:
        return []
    out = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                out[i + j] = (out[i + j] + x * y) % R
    return trim(out)


def divide_by_vanishing_poly(p, n):
    """(q, rem) with p = q * (X^n - 1) + rem  (ark-poly divide_by_vanishing_poly)"""
    if len(p) <= n:
        return [], trim(p)
    q = list(p[n:])
    for i in range(1, len(p) // n):
        for j in range(len(q) - n * i):
            q[j] = (q[j] + p[n * (i + 1) + j]) % R
    rem = list(p[:n])
    for j in range(n):
        if j < len(q):
            rem[j] = (rem[j] + q[j]) % R
    return trim(q), trim(rem)


def l0_poly(n):
    """(X^n - 1) / (n (X - 1)) = (1/n) sum_{i<n} X^i   (plonk/src/utils.rs:150-159)"""
    return [pow(n, -1, R)] * n


def slices(t, n):
    """SlicedPoly::<3>::from_poly(t, n): three coefficient slices of length n (last one shorter)"""
    return [trim(t[0:n]), trim(t[n:2 * n]), trim(t[2 * n:3 * n])]


# ---- a valid circuit instance: the squaring chain of SURVEY.md section 8d -----------------------------
COSETS = [2, 3, 4]  # Permutation::cosets(rows): first three k with k^n != 1 (1 is excluded)


def squaring_chain(log_n: int, x0: int = 3, blinders=None):
    """gates = n - 3 multiplications x_{j+1} = x_j * x_j; returns the circuit tables and a valid
    witness.  Columns are padded to n - 3 rows and receive 3 blinding rows (proof.rs:43-49)."""
    n = 1 << log_n
    g = n - 3
    a, b, c = [], [], []
    x = x0 % R
    for _ in range(g):
        a.append(x)
        b.append(x)
        x = x * x % R
        c.append(x)
    bl = blinders or [[(7 * i + 3 * j + 11) % R for j in range(3)] for i in range(3)]
    cols = [a + bl[0], b + bl[1], c + bl[2]]
    # selectors: Mul = [q_l, q_r, q_o, q_m, q_c] = [0, 0, 1, 1, 0]; Dummy rows all zero (builder.rs:318-324)
    q = {"q_l": [0] * n, "q_r": [0] * n, "q_o": [1] * g + [0] * 3, "q_m": [1] * g + [0] * 3, "q_c": [0] * n}
    # copy constraints as a permutation of the 3n cells (flat index j + i*n): (0,j)~(1,j), (2,j)~(0,j+1)~(1,j+1)
    perm = list(range(3 * n))

    def cyc(cells):
        for u, v in zip(cells, cells[1:] + cells[:1]):
            perm[u] = v

    cyc([0, n])                                        # a_0, b_0
    for j in range(g - 1):
        cyc([2 * n + j, j + 1, n + j + 1])             # c_j, a_{j+1}, b_{j+1}
    return n, cols, q, perm


def compile_permutation(perm, n, log_n):
    """per cell (tag, value): tag = k_i w^j, value = k_i' w^j' of the cell it maps to (lib.rs:108-119)"""
    w = O.domain_root(log_n)
    roots = [pow(w, j, R) for j in range(n)]
    ids = [[COSETS[i] * roots[j] % R for j in range(n)] for i in range(3)]
    sig = [[COSETS[perm[j + i * n] // n] * roots[perm[j + i * n] % n] % R for j in range(n)] for i in range(3)]
    return ids, sig


def grand_product(cols, ids, sig, beta, gamma, n):
    """CompiledPermutation::prove: n + 1 values starting at 1 (one field division per cell)"""
    acc = [1]
    state = 1
    for j in range(n):
        row = 1
        for i in range(3):
            num = (cols[i][j] + beta * ids[i][j] + gamma) % R
            den = (cols[i][j] + beta * sig[i][j] + gamma) % R
            row = row * num * pow(den, -1, R) % R
        state = state * row % R
        acc.append(state)
    return acc


def quotient_polynomial(log_n, wires, z, zw, q, sigma_polys, alpha, beta, gamma, pi):
    """plonk/src/proof.rs:292-375 with the reference's schoolbook products.  All arguments are
    coefficient vectors.  Returns (t, remainder); the reference discards the remainder."""
    n = 1 << log_n
    a, b, c = wires
    line1 = p_add(p_add(p_add(p_sub(p_add(naive_mul(q["q_l"], a), naive_mul(q["q_r"], b)), naive_mul(q["q_o"], c)),
                              naive_mul(naive_mul(q["q_m"], a), b)), q["q_c"]), pi)
    f = [p_add(w, trim([gamma, k * beta % R])) for w, k in zip(wires, COSETS)]
    line2 = naive_mul(naive_mul(naive_mul(f[0], f[1]), f[2]), z)
    g = [p_add(p_add(w, p_scale(s, beta)), trim([gamma])) for w, s in zip(wires, sigma_polys)]
    line3 = naive_mul(naive_mul(naive_mul(g[0], g[1]), g[2]), zw)
    zm1 = list(z)
    zm1[0] = (zm1[0] - 1) % R
    line4 = naive_mul(trim(zm1), l0_poly(n))
    target = p_add(p_add(p_add(line1, p_scale(line2, alpha)), p_scale(p_scale(line3, R - 1), alpha)),
                   p_scale(line4, alpha * alpha % R))
    return divide_by_vanishing_poly(target, n), (line1, line4)


def prove_round_2_3(log_n, alpha, beta, gamma, x0=3, pi_evals=None):
    """Everything prove() does between the wire commitments and the quotient commitment, for the
    squaring-chain circuit: returns the coefficient vectors the GPU path consumes and the reference
    quotient."""
    n, cols, q_evals, perm = squaring_chain(log_n, x0)
    ids, sig = compile_permutation(perm, n, log_n)
    wires = [O.interpolate(col, log_n) for col in cols]                 # proof.rs:50
    # proof.rs:113-115 re-evaluates the wire polynomials; identical to `cols`
    acc = grand_product(cols, ids, sig, beta, gamma, n)
    assert acc[n] == 1, "copy constraints not satisfied"
    evals = acc[:n]                                                     # evals.pop()
    z = O.interpolate(evals, log_n)
    zw = O.interpolate(evals[1:] + evals[:1], log_n)                    # rotate_left(1)
    q = {k: O.interpolate(v, log_n) for k, v in q_evals.items()}       # builder.rs:84-88
    sigma_polys = [O.interpolate(s, log_n) for s in sig]                # proof.rs:334-338
    pi = O.interpolate(pi_evals or [0] * n, log_n)
    (t, rem), (line1, line4) = quotient_polynomial(log_n, wires, z, zw, q, sigma_polys, alpha, beta, gamma, pi)
    return {"n": n, "wires": wires, "z": z, "zw": zw, "q": q, "sigma": sigma_polys, "pi": pi, "t": t, "rem": rem,
            "line1": line1, "line4": line4, "cols": cols, "z_evals": evals}


# ---- the whole prove(): rounds 1-5 with injected Fiat-Shamir challenges ------------------------------------
def add_to_poly(p, x):
    """plonk/src/utils.rs:13-20"""
    if not p:
        return trim([x % R])
    q = list(p)
    q[0] = (q[0] + x) % R
    return trim(q)


def linearisation_poly(log_n, q, sigma_polys, cosets, adv, z_eval_w, z, challenges, zeta, t_slices, public_eval):
    """plonk/src/proof.rs:376-439.  adv = (a(zeta), b(zeta), c(zeta)); z_eval_w = Z(zeta w)."""
    n = 1 << log_n
    a, b, c = adv
    alpha, beta, gamma = challenges
    line1 = p_add(p_add(p_scale(q["q_l"], a), p_sub(p_scale(q["q_r"], b), p_scale(q["q_o"], c))),
                  p_add(p_scale(q["q_m"], a * b % R), q["q_c"]))
    line1 = add_to_poly(line1, public_eval)
    l2 = 1
    for k, e in zip(cosets, adv):
        l2 = l2 * ((e + k * beta * zeta + gamma) % R) % R
    line2 = p_scale(z, l2)
    sig_ev = [O.poly_eval(s, zeta) for s in sigma_polys]
    ab = (a + beta * sig_ev[0] + gamma) * (b + beta * sig_ev[1] + gamma) % R
    cp_c = add_to_poly(p_scale(sigma_polys[2], beta), (gamma + c) % R)
    line3 = p_scale(p_scale(cp_c, ab), z_eval_w)
    copy_constrain = p_sub(line2, line3)
    l0_eval = O.poly_eval(l0_poly(n), zeta)
    line4 = p_scale(add_to_poly(z, R - 1), l0_eval)
    compact = []
    for idx, sl in enumerate(t_slices):                   # SlicedPoly::compact, degree = n
        compact = p_add(compact, p_scale(sl, pow(zeta, n * idx, R)))
    line5 = p_scale(compact, (pow(zeta, n, R) - 1) % R)   # evaluate_vanishing_polynomial
    return p_sub(p_add(p_add(line1, p_scale(copy_constrain, alpha)), p_scale(line4, alpha * alpha % R)), line5)


def prove(log_n, cols, q_evals, perm, pi_evals, challenges, zeta, commit):
    """plonk/src/proof.rs:96-194.  `commit(coeffs)` is the KZG commitment function (an MSM against the
    SRS).  Returns the proof elements in the order the reference produces them."""
    n = 1 << log_n
    alpha, beta, gamma = challenges
    w = O.domain_root(log_n)
    ids, sig = compile_permutation(perm, n, log_n)
    wires = [O.interpolate(col, log_n) for col in cols]
    pi = O.interpolate(pi_evals, log_n)
    commitments = [commit(p) for p in wires]                                  # round1, :107-110
    acc = grand_product(cols, ids, sig, beta, gamma, n)
    evals = acc[:n]
    z = O.interpolate(evals, log_n)
    zw = O.interpolate(evals[1:] + evals[:1], log_n)
    z_commit = commit(z)                                                      # :129
    q = {k: O.interpolate(v, log_n) for k, v in q_evals.items()}
    sigma_polys = [O.interpolate(s, log_n) for s in sig]
    public_eval = O.poly_eval(pi, zeta)                                       # :138
    (t, rem), _ = quotient_polynomial(log_n, wires, z, zw, q, sigma_polys, alpha, beta, gamma, pi)
    t_slices = slices(t, n)

    def open_(p, x):                                                          # kzg/src/lib.rs:55-64
        qq, y = O.poly_div_linear(p, x)
        return commit(qq), y

    openings = [open_(p, zeta) for p in wires]                                # :147-154
    adv = [o[1] for o in openings]
    z_open = open_(z, zeta)                                                   # :162
    zw_open = open_(z, zeta * w % R)                                          # :163
    r = linearisation_poly(log_n, q, sigma_polys, COSETS, adv, zw_open[1], z, challenges, zeta, t_slices, public_eval)
    r_open = open_(r, zeta)                                                   # :175
    t_commit = [commit(s) for s in t_slices]                                  # :181
    return {"commit": commitments, "open": openings, "z_commit": z_commit, "z_open": z_open, "zw_open": zw_open,
            "t_commit": t_commit, "r_open": r_open, "r": r, "t": t, "rem": rem, "wires": wires, "z": z}


# ---- the README circuit (a*a + b*b == c*c), SURVEY.md KAT-5 --------------------------------------------
def pythagorean_circuit(inputs, blinders=None):
    """README.md:16-27 / plonk/src/builder/test.rs:25-37: gates Mul, Mul, Mul, Add -> n = 8 rows.
    Returns (log_n, cols, q_evals, perm) exactly as CircuitBuilder::compile + ComputeVar produce them
    (witness columns before blinding a=[x,y,z,x^2,0] b=[x,y,z,y^2,0] c=[x^2,y^2,z^2,x^2+y^2,0])."""
    x, y, z = [v % R for v in inputs]
    log_n, n = 3, 8
    a = [x, y, z, x * x % R, 0]
    b = [x, y, z, y * y % R, 0]
    c = [x * x % R, y * y % R, z * z % R, (x * x + y * y) % R, 0]
    bl = blinders or [[11 + 3 * i + j for j in range(3)] for i in range(3)]
    cols = [a + bl[0], b + bl[1], c + bl[2]]
    mul, add, dummy = [0, 0, 1, 1, 0], [1, 1, 1, 0, 0], [0, 0, 0, 0, 0]
    rows = [mul, mul, mul, add] + [dummy] * 4                      # builder.rs:318-324
    q = {name: [rows[j][k] for j in range(n)] for k, name in enumerate(("q_l", "q_r", "q_o", "q_m", "q_c"))}
    perm = list(range(3 * n))
    # copy constraints (col,row): (0,0)~(1,0) (0,1)~(1,1) (0,2)~(1,2) (2,0)~(0,3) (2,1)~(1,3) (2,3)~(2,2)
    for (ci, ri), (cj, rj) in (((0, 0), (1, 0)), ((0, 1), (1, 1)), ((0, 2), (1, 2)), ((2, 0), (0, 3)), ((2, 1), (1, 3)),
                               ((2, 3), (2, 2))):
        u, v = ri + ci * n, rj + cj * n
        perm[u], perm[v] = perm[v], perm[u]
    return log_n, cols, q, perm


def batched_opening(polys, v, zeta, commit):
    """Batched KZG opening of `polys` (coefficient lists) at zeta with challenge v -- the reference's to-do
    (/root/reference/README.md:4 "opening batching"), stated with the reference's own open()
    (/root/reference/kzg/src/lib.rs:55-64) applied to F = sum_i v^i p_i.  Returns (W, F(zeta))."""
    f, pw = [], 1
    for p in polys:
        f = p_add(f, p_scale(p, pw))
        pw = pw * v % R
    q, y = O.poly_div_linear(f, zeta)
    return commit(q), y
