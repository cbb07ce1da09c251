"""CPU oracle (TEST INFRASTRUCTURE ONLY) for TyPLONK's MSM + NTT hot path.

Plain Python big-int restatement of the arithmetic the reference reaches through
arkworks 0.3.0 (un-vendored; pins in /root/reference/Cargo.lock:17-18, 28-29, 42-43, 82-83):

  * kzg::KzgScheme::evaluate_in_s / commit / open / identity   kzg/src/lib.rs:37-64, 82-85
  * kzg::srs::Srs::from_secret                                  kzg/src/srs.rs:15-34
  * Evaluations::interpolate / evaluate_over_domain (radix-2)   call sites plonk/src/proof.rs:50,106,115,125,128

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product path (typlonk_amd/) never does.

PARITY PINNING: the reference is Rust-only and cannot be compiled or run in this environment
(no cargo/rustc), and its arithmetic lives in un-vendored crates.  This oracle is therefore pinned
against (i) the reference's own test identities -- kzg `commit` (kzg/src/lib.rs:95-109),
`scalar_mul` (:160-171), plonk utils `l0` (plonk/src/utils.rs:161-177) -- and (ii) the public
BLS12-381 constants.  No vector produced by running the reference exists: beyond those identities
parity is UNPINNED.

Everything here works on canonical integers.  `to_mont_limbs`/`from_mont_limbs` convert to the
arkworks in-memory form (little-endian u64 limbs, Montgomery domain) used at the C-ABI.
"""
from __future__ import annotations

# ---------------------------------------------------------------------------------------------
# constants (public BLS12-381 definition; checked in tests/test_oracle.py)
# ---------------------------------------------------------------------------------------------
P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
B_COEFF = 4
GX = 0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB
GY = 0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1
G1 = (GX, GY)
INF = None  # affine identity; arkworks encodes it (x=0, y=1, infinity=true)

FR_TWO_ADICITY = 32
FR_GENERATOR = 7
# ark-bls12-381 FrParameters::TWO_ADIC_ROOT_OF_UNITY = 7^((r-1)/2^32)
FR_ROOT_OF_UNITY = pow(FR_GENERATOR, (R - 1) >> FR_TWO_ADICITY, R)

FR_MONT_R = (1 << 256) % R
FQ_MONT_R = (1 << 384) % P


# ---------------------------------------------------------------------------------------------
# arkworks in-memory limb form  (ark-ff 0.3.0 Fp256/Fp384: Montgomery residue, LE u64 limbs)
# ---------------------------------------------------------------------------------------------
def fr_to_mont_limbs(x: int) -> list[int]:
    v = (x % R) * FR_MONT_R % R
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def fr_from_mont_limbs(limbs) -> int:
    v = sum(int(l) << (64 * i) for i, l in enumerate(limbs))
    return v * pow(FR_MONT_R, -1, R) % R


def fq_to_mont_limbs(x: int) -> list[int]:
    v = (x % P) * FQ_MONT_R % P
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(6)]


def fq_from_mont_limbs(limbs) -> int:
    v = sum(int(l) << (64 * i) for i, l in enumerate(limbs))
    return v * pow(FQ_MONT_R, -1, P) % P


def g1_to_limbs(pt):
    """affine point -> (12 u64 limbs x||y Montgomery, inf flag).  Identity is (0, 1, inf=1) as in
    ark-ec 0.3.0 GroupAffine::zero()."""
    if pt is INF:
        return fq_to_mont_limbs(0) + fq_to_mont_limbs(1), 1
    return fq_to_mont_limbs(pt[0]) + fq_to_mont_limbs(pt[1]), 0


def g1_from_limbs(limbs, inf):
    if inf:
        return INF
    return (fq_from_mont_limbs(limbs[:6]), fq_from_mont_limbs(limbs[6:12]))


# ---------------------------------------------------------------------------------------------
# G1 affine arithmetic (independent of the Jacobian formulas used by the C restatement and
# of the XYZZ formulas used on the GPU)
# ---------------------------------------------------------------------------------------------
def g1_is_on_curve(pt) -> bool:
    if pt is INF:
        return True
    x, y = pt
    return (y * y - x * x * x - B_COEFF) % P == 0


def g1_neg(pt):
    if pt is INF:
        return INF
    return (pt[0], (-pt[1]) % P)


def g1_add(a, b):
    if a is INF:
        return b
    if b is INF:
        return a
    x1, y1 = a
    x2, y2 = b
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return INF
        lam = 3 * x1 * x1 * pow(2 * y1, -1, P) % P
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    y3 = (lam * (x1 - x3) - y1) % P
    return (x3, y3)


def g1_mul(pt, k: int):
    """k*pt, MSB-first double-and-add as ark-ec 0.3.0 AffineCurve::mul -> mul_bits
    (call site kzg/src/lib.rs:49).  The scalar is the canonical integer of the Fr element."""
    k %= R
    acc = INF
    for bit in bin(k)[2:] if k else "":
        acc = g1_add(acc, acc)
        if bit == "1":
            acc = g1_add(acc, pt)
    return acc


# ---------------------------------------------------------------------------------------------
# kzg crate
# ---------------------------------------------------------------------------------------------
def srs_from_secret(s: int, gates: int):
    """kzg/src/srs.rs:15-24, 30-34: g1 = [G, sG, s^2 G, ...] of length gates+3, affine."""
    out = [G1]
    sx = s % R
    for _ in range(gates + 3 - 1):
        out.append(g1_mul(G1, sx))
        sx = sx * s % R
    return out[: gates + 3]


def srs_from_secret_fast(s: int, length: int):
    """Same vector as srs_from_secret(s, length-3) but built as successive multiplications
    by s of the previous point (exact same group elements; only used to make big fixtures)."""
    out = [G1]
    for _ in range(length - 1):
        out.append(g1_mul(out[-1], s))
    return out


def poly_trim(coeffs):
    """DensePolynomial::from_coefficients_vec strips trailing zeros (ark-poly 0.3.0)."""
    c = [x % R for x in coeffs]
    while c and c[-1] == 0:
        c.pop()
    return c


def poly_degree(coeffs) -> int:
    return 0 if not coeffs else len(coeffs) - 1


def poly_eval(coeffs, x: int) -> int:
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % R
    return acc


def msm_naive(scalars, points):
    """kzg/src/lib.rs:44-52: zip (truncates to the shorter), per-term scalar mul, sum."""
    acc = INF
    for k, pt in zip(scalars, points):
        acc = g1_add(acc, g1_mul(pt, k))
    return acc


def kzg_commit(srs_g1, coeffs):
    """kzg/src/lib.rs:37-54.  Panics (AssertionError) when srs.len() <= degree (:43)."""
    assert len(srs_g1) > poly_degree(coeffs)
    return msm_naive(coeffs, srs_g1)


def poly_div_linear(coeffs, z: int):
    """(p(X) - p(z)) / (X - z) by synthetic division -- kzg/src/lib.rs:57-61."""
    y = poly_eval(coeffs, z)
    n = len(coeffs)
    if n <= 1:
        return [], y
    q = [0] * (n - 1)
    carry = 0
    for i in range(n - 1, 0, -1):
        carry = (coeffs[i] + carry * z) % R
        q[i - 1] = carry
    return poly_trim(q), y


def kzg_open(srs_g1, coeffs, z: int):
    """kzg/src/lib.rs:55-64 -> (W, y)."""
    assert len(coeffs) >= 1  # `.expect("at least 1")` at :58
    q, y = poly_div_linear(coeffs, z)
    return kzg_commit(srs_g1, q), y


# ---------------------------------------------------------------------------------------------
# ark-poly 0.3.0 Radix2EvaluationDomain<Fr>
# ---------------------------------------------------------------------------------------------
def domain_root(log_n: int) -> int:
    """group_gen = TWO_ADIC_ROOT_OF_UNITY squared (32 - log_n) times."""
    assert 0 <= log_n <= FR_TWO_ADICITY
    w = FR_ROOT_OF_UNITY
    for _ in range(FR_TWO_ADICITY - log_n):
        w = w * w % R
    return w


def dft_naive(vals, w: int):
    n = len(vals)
    out = []
    for k in range(n):
        wk = pow(w, k, R)
        acc, x = 0, 1
        for v in vals:
            acc = (acc + v * x) % R
            x = x * wk % R
        out.append(acc)
    return out


def _ntt_rec(vals, w):
    n = len(vals)
    if n == 1:
        return vals
    e = _ntt_rec(vals[0::2], w * w % R)
    o = _ntt_rec(vals[1::2], w * w % R)
    out = [0] * n
    x = 1
    for k in range(n // 2):
        t = x * o[k] % R
        out[k] = (e[k] + t) % R
        out[k + n // 2] = (e[k] - t) % R
        x = x * w % R
    return out


def ntt(vals, log_n: int, inverse: bool = False, coset: int | None = None):
    """Natural order in / natural order out.
    forward : out[k] = sum_i vals[i] * (g w^k)^i            (EvaluationDomain::fft / coset_fft)
    inverse : coefficients, multiplied by n^-1 (and g^-i)   (EvaluationDomain::ifft / coset_ifft)"""
    n = 1 << log_n
    v = [x % R for x in vals] + [0] * (n - len(vals))
    assert len(v) == n
    w = domain_root(log_n)
    if not inverse:
        if coset is not None:
            g, x = coset % R, 1
            for i in range(n):
                v[i] = v[i] * x % R
                x = x * g % R
        return _ntt_rec(v, w)
    out = _ntt_rec(v, pow(w, -1, R))
    ninv = pow(n, -1, R)
    out = [x * ninv % R for x in out]
    if coset is not None:
        gi, x = pow(coset, -1, R), 1
        for i in range(n):
            out[i] = out[i] * x % R
            x = x * gi % R
    return out


def interpolate(evals, log_n: int):
    """Evaluations::interpolate = from_coefficients_vec(ifft(evals)) (trims trailing zeros)."""
    return poly_trim(ntt(evals, log_n, inverse=True))


def evaluate_over_domain(coeffs, log_n: int):
    return ntt(coeffs, log_n)


# ---------------------------------------------------------------------------------------------
# deterministic input generator shared by tests / bench
# ---------------------------------------------------------------------------------------------
class Xoshiro256ss:
    M = 0xFFFFFFFFFFFFFFFF

    def __init__(self, seed: int):
        # splitmix64 seeding
        self.s = []
        x = seed & self.M
        for _ in range(4):
            x = (x + 0x9E3779B97F4A7C15) & self.M
            z = x
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & self.M
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & self.M
            self.s.append(z ^ (z >> 31))

    @staticmethod
    def _rotl(x, k):
        return ((x << k) | (x >> (64 - k))) & 0xFFFFFFFFFFFFFFFF

    def next(self) -> int:
        s = self.s
        result = (self._rotl((s[1] * 5) & self.M, 7) * 9) & self.M
        t = (s[1] << 17) & self.M
        s[2] ^= s[0]
        s[3] ^= s[1]
        s[1] ^= s[2]
        s[0] ^= s[3]
        s[2] ^= t
        s[3] = self._rotl(s[3], 45)
        return result

    def fr(self) -> int:
        """uniform canonical integer in [0, r): 4 x u64 (limb 0 first), clear the top bit, reject >= r."""
        while True:
            limbs = [self.next() for _ in range(4)]
            limbs[3] &= 0x7FFFFFFFFFFFFFFF
            v = sum(l << (64 * i) for i, l in enumerate(limbs))
            if v < R:
                return v


def random_frs(seed: int, n: int):
    g = Xoshiro256ss(seed)
    return [g.fr() for _ in range(n)]
