"""GPU parity of the NTT path (typlonk_ntt_fr*) against the CPU oracle, through the C ABI.
Bit-exact: every comparison is integer equality of canonical Fr values / raw limbs."""
import numpy as np
import pytest

from helpers import O, fr_pack, fr_unpack

pytestmark = pytest.mark.gpu


def rand_limbs(seed, n):
    """n valid Montgomery residues (any 256-bit value < r) straight from numpy"""
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * 2 + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
    return a


@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 4, 5, 7, 8, 10, 11, 12, 13])
def test_ntt_forward_inverse_vs_oracle(ctx, log_n):
    n = 1 << log_n
    v = O.random_frs(0x5EED + log_n, n)
    f = ctx.ntt(fr_pack(v), log_n)
    assert fr_unpack(f) == O.ntt(v, log_n)
    i = ctx.ntt(fr_pack(v), log_n, inverse=True)
    assert fr_unpack(i) == O.ntt(v, log_n, inverse=True)


@pytest.mark.parametrize("log_n", [1, 3, 8, 10, 12])
def test_coset_ntt_vs_oracle(ctx, log_n):
    n = 1 << log_n
    v = O.random_frs(0xC05E7 + log_n, n)
    g = np.array(O.fr_to_mont_limbs(7), dtype=np.uint64)
    f = ctx.ntt(fr_pack(v), log_n, coset=g)
    assert fr_unpack(f) == O.ntt(v, log_n, coset=7)
    i = ctx.ntt(fr_pack(v), log_n, inverse=True, coset=g)
    assert fr_unpack(i) == O.ntt(v, log_n, inverse=True, coset=7)


def test_kat3_ntt4(ctx):
    # SURVEY.md KAT-3 / tests/golden: NTT_4([1,2,3,4]) with arkworks' omega_4
    out = fr_unpack(ctx.ntt(fr_pack([1, 2, 3, 4]), 2))
    assert out == [0xA,
                   0x73EDA753299D7D4718963E6B1D9BCE637BB7A3FE13F85BFEFFFDFFFEFFFFFFFF,
                   0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFEFFFFFFFF,
                   0x11AA3999CEC0609A1D8060004EC0600000001FFFFFFFFFFFE]


def test_l0_identity_2_16(ctx):
    """reference test plonk/src/utils.rs:161-177: L0 = (1/N) sum X^i evaluates to [1,0,...,0] on the
    2^16 domain (sum of evaluations == 1)."""
    log_n, n = 16, 1 << 16
    ninv = pow(n, -1, O.R)
    coeffs = np.tile(np.array(O.fr_to_mont_limbs(ninv), dtype=np.uint64), (n, 1))
    ev = ctx.ntt(coeffs, log_n)
    one = np.array(O.fr_to_mont_limbs(1), dtype=np.uint64)
    assert (ev[0] == one).all()
    assert not ev[1:].any()


@pytest.mark.parametrize("log_n", [14, 16, 17, 20, 22, 24])
def test_roundtrip_and_spot_check_large(ctx, log_n):
    n = 1 << log_n
    x = rand_limbs(log_n, n)
    f = ctx.ntt(x, log_n)
    back = ctx.ntt(f, log_n, inverse=True)
    assert (back == x).all()
    # spot-check a few outputs against the definition X[k] = sum_i x[i] w^(ik), on a sparse input
    sparse = np.zeros((n, 4), dtype=np.uint64)
    idxs = [0, 1, 5, n // 3, n - 1]
    vals = O.random_frs(99 + log_n, len(idxs))
    for i, v in zip(idxs, vals):
        sparse[i] = O.fr_to_mont_limbs(v)
    fs = ctx.ntt(sparse, log_n)
    w = O.domain_root(log_n)
    for k in [0, 1, 2, n // 2 + 3, n - 1, 12345 % n]:
        exp = sum(v * pow(w, i * k, O.R) for i, v in zip(idxs, vals)) % O.R
        assert O.fr_from_mont_limbs([int(t) for t in fs[k]]) == exp


def _to_int(row):
    return sum(int(row[j]) << (64 * j) for j in range(4))


def test_linearity_2_20(ctx):
    """NTT(a + b) == NTT(a) + NTT(b) at 2^20 (Montgomery form is linear, so limbs add mod r)."""
    log_n, n = 20, 1 << 20
    a, b = rand_limbs(1, n), rand_limbs(2, n)
    # a, b < 2^254 so a + b < 2^255: add as Python ints per element would be slow; do it with
    # object arrays in bulk
    ai = (a[:, 0].astype(object) + (a[:, 1].astype(object) << 64) + (a[:, 2].astype(object) << 128)
          + (a[:, 3].astype(object) << 192))
    bi = (b[:, 0].astype(object) + (b[:, 1].astype(object) << 64) + (b[:, 2].astype(object) << 128)
          + (b[:, 3].astype(object) << 192))
    si = (ai + bi) % O.R
    s = np.empty((n, 4), dtype=np.uint64)
    mask = (1 << 64) - 1
    for j in range(4):
        s[:, j] = ((si >> (64 * j)) & mask).astype(np.uint64)
    fa, fb, fs = ctx.ntt(a, log_n), ctx.ntt(b, log_n), ctx.ntt(s, log_n)
    rng = np.random.default_rng(7)
    for k in [0, 1, n - 1] + [int(x) for x in rng.integers(0, n, size=64)]:
        assert (_to_int(fa[k]) + _to_int(fb[k])) % O.R == _to_int(fs[k])


def test_errors(ctx):
    from typlonk_amd.capi import TyplonkError, ERR_DOMAIN

    with pytest.raises(TyplonkError) as e:
        ctx.ntt_devptr(1, 33)
    assert e.value.code == ERR_DOMAIN


def test_golden_ntt_fixtures(ctx):
    from helpers import load_golden

    g = np.array(O.fr_to_mont_limbs(7), dtype=np.uint64)
    for case in load_golden("ntt.json"):
        L = case["log_n"]
        v = fr_pack([int(x, 16) for x in case["input"]])
        assert fr_unpack(ctx.ntt(v, L)) == [int(x, 16) for x in case["forward"]]
        assert fr_unpack(ctx.ntt(v, L, inverse=True)) == [int(x, 16) for x in case["inverse"]]
        assert fr_unpack(ctx.ntt(v, L, coset=g)) == [int(x, 16) for x in case["coset7_forward"]]
        assert fr_unpack(ctx.ntt(v, L, inverse=True, coset=g)) == [int(x, 16) for x in case["coset7_inverse"]]


@pytest.mark.parametrize("log_n", [16, 20, 22, 24, 25])
def test_full_size_vs_c_oracle(ctx, log_n):
    """BASELINE config sizes: whole-vector equality with the C restatement of ark-poly's radix-2 FFT; 2^25 is the first
    size that takes four passes (checked by hand up to 2^27, forward, round trip and inverse coset: all equal)"""
    from oracle import coracle as CO

    x = rand_limbs(1000 + log_n, 1 << log_n)
    assert (ctx.ntt(x, log_n) == CO.ntt(x, log_n)).all()
    g = np.array(O.fr_to_mont_limbs(7), dtype=np.uint64)
    if log_n <= 20:
        assert (ctx.ntt(x, log_n, inverse=True, coset=g) == CO.ntt(x, log_n, inverse=True, coset=g)).all()
    if log_n == 24:   # the quotient domain of a 2^22-row circuit: forward on the coset g = 7
        assert (ctx.ntt(x, log_n, coset=g) == CO.ntt(x, log_n, coset=g)).all()


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_both_butterfly_kernels_give_the_c_oracles_words(built, mode, monkeypatch):
    """The NTT has two pass kernels: 8 x 32-bit words (ff.hpp) and nine 30-bit limbs with twiddles in the 2^270 domain
    (fr30.hpp; since round 4 the default wherever its full twiddle tables exist, i.e. up to 2^24 points).
    TYPLONK_NTT_FR30 = 0 / 1 / 2 selects the 8 x 32 kernel everywhere / the default policy / the 30-bit kernel everywhere; each
    mode must reproduce the C restatement of ark-poly's radix-2 FFT word for word -- forward, inverse (n^-1), coset and
    inverse coset, one to three passes, the extremes of the input range (0, r - 1) included."""
    import typlonk_amd
    from oracle import coracle as CO

    monkeypatch.setenv("TYPLONK_NTT_FR30", str(mode))
    c2 = typlonk_amd.Context(0)
    try:
        g = np.array(O.fr_to_mont_limbs(7), dtype=np.uint64)
        g2 = np.array(O.fr_to_mont_limbs(0x123456789ABCDEF), dtype=np.uint64)
        for log_n in (1, 2, 5, 9, 10, 11, 14, 17, 19, 20, 21):
            x = rand_limbs(7000 + 31 * log_n + mode, 1 << log_n)
            x[0] = 0
            x[-1] = np.array(O.fr_to_mont_limbs(O.R - 1), dtype=np.uint64)
            assert (c2.ntt(x, log_n) == CO.ntt(x, log_n)).all(), (mode, log_n, "forward")
            assert (c2.ntt(x, log_n, inverse=True) == CO.ntt(x, log_n, inverse=True)).all(), (mode, log_n, "inverse")
            assert (c2.ntt(x, log_n, coset=g) == CO.ntt(x, log_n, coset=g)).all(), (mode, log_n, "coset")
            assert (c2.ntt(x, log_n, inverse=True, coset=g2) == CO.ntt(x, log_n, inverse=True, coset=g2)).all(), (mode, log_n, "icoset")
            # all r - 1: the largest canonical input in every position
            top = np.tile(np.array(O.fr_to_mont_limbs(O.R - 1), dtype=np.uint64), (1 << log_n, 1))
            if log_n <= 17:
                assert (c2.ntt(top, log_n) == CO.ntt(top, log_n)).all(), (mode, log_n, "all r-1")
    finally:
        c2.close()


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_batched_transforms_equal_single_calls_in_every_kernel_mode(built, mode, monkeypatch):
    """typlonk_ntt_fr_batch_devptr: `count` vectors of one size / direction / coset, every pass ONE launch carrying count x
    the tiles of a vector.  The reference transforms in such groups (the three wire columns plonk/src/proof.rs:50, their
    re-evaluation :113-115, the three sigmas :334-338, the five selectors plonk/src/builder.rs:84-88).  Must be bit-identical
    to `count` single calls -- and to the C restatement of ark-poly's FFT -- in all three TYPLONK_NTT_FR30 modes, for
    count = 1, 3, 5, 8 and 11 (11 = two launch groups), forward, inverse, coset and inverse coset, one to three passes and
    the two-pass 2^20 plan; count = 0 is a no-op; overlapping vectors and a null entry are refused."""
    import typlonk_amd
    from oracle import coracle as CO
    from typlonk_amd.capi import ERR_INVALID_ARG, TyplonkError

    monkeypatch.setenv("TYPLONK_NTT_FR30", str(mode))
    c2 = typlonk_amd.Context(0)
    try:
        g = np.array(O.fr_to_mont_limbs(7), dtype=np.uint64)
        c2.ntt_batch_devptr([], 10)                                   # count = 0
        for log_n, counts in ((0, (3,)), (1, (3,)), (6, (5,)), (10, (1, 8)), (13, (3, 11)), (17, (3,)), (20, (3, 5))):
            n = 1 << log_n
            for count in counts:
                buf = c2.alloc(n * count)
                x = rand_limbs(9100 + 17 * log_n + count + mode, n * count)
                x[0] = 0
                x[-1] = np.array(O.fr_to_mont_limbs(O.R - 1), dtype=np.uint64)
                ptrs = [buf.devptr + 32 * n * v for v in range(count)]
                for inverse, coset in ((False, None), (True, None), (False, g), (True, g)):
                    if log_n == 20 and coset is not None and inverse:
                        continue                                        # (keeps the test inside a few seconds of CPU oracle)
                    buf.upload(x)
                    c2.ntt_batch_devptr(ptrs, log_n, inverse=inverse, coset=coset)
                    got = buf.download()
                    for v in range(count):
                        xv = x[v * n:(v + 1) * n]
                        single = c2.ntt(xv, log_n, inverse=inverse, coset=coset)
                        assert (got[v * n:(v + 1) * n] == single).all(), (mode, log_n, count, v, inverse, coset is not None)
                        if v == 0 or log_n <= 13:
                            assert (single == CO.ntt(xv, log_n, inverse=inverse, coset=coset)).all()
        buf = c2.alloc(3 << 10)
        with pytest.raises(TyplonkError) as e:
            c2.ntt_batch_devptr([buf.devptr, buf.devptr + 32 * 512], 10)   # the second vector starts inside the first
        assert e.value.code == ERR_INVALID_ARG
        with pytest.raises(TyplonkError) as e:
            c2.ntt_batch_devptr([buf.devptr, 0], 10)
        assert e.value.code == ERR_INVALID_ARG
    finally:
        c2.close()


@pytest.mark.parametrize("big", [0, 2])
def test_batched_2_20_both_plans(built, big, monkeypatch):
    """the 2^20 batch in both pass plans (TYPLONK_NTT_BIG = 0: three passes on 1024-element tiles, 2: the two-pass plan on
    4096-element tiles): same words as the C oracle, all three vectors"""
    import typlonk_amd
    from oracle import coracle as CO

    monkeypatch.setenv("TYPLONK_NTT_BIG", str(big))
    c2 = typlonk_amd.Context(0)
    try:
        n, count = 1 << 20, 3
        x = rand_limbs(9900 + big, n * count)
        buf = c2.alloc(n * count)
        for inverse in (False, True):
            buf.upload(x)
            c2.ntt_batch_devptr([buf.devptr + 32 * n * v for v in range(count)], 20, inverse=inverse)
            got = buf.download()
            for v in range(count):
                assert (got[v * n:(v + 1) * n] == CO.ntt(x[v * n:(v + 1) * n], 20, inverse=inverse)).all(), (big, inverse, v)
    finally:
        c2.close()


@pytest.mark.slow
@pytest.mark.parametrize("log_n", [26, 27])
def test_beyond_the_full_table_limit_vs_c_oracle(ctx, log_n):
    """typlonk_ntt_fr accepts every two-adic size; above 2^24 points the inter-pass twiddles are composed from two-level
    tables instead of read from a full table, and four passes are needed.  Whole-vector equality with the C restatement
    (its OpenMP form: the one-thread form takes a minute here) at 2^26 and 2^27 (a 4-GiB vector), forward, and the
    inverse round trip."""
    from conftest import need_resources
    from oracle import coracle as CO

    need_resources(host_gib=5 * (32 << log_n) / (1 << 30) + 2, hbm_gib=3 * (32 << log_n) / (1 << 30) + 2)
    x = rand_limbs(2000 + log_n, 1 << log_n)
    f = ctx.ntt(x, log_n)
    assert (f == CO.ntt(x, log_n, threads=0)).all()
    assert (ctx.ntt(f, log_n, inverse=True) == x).all()


def test_the_two_kernels_agree_on_inputs_that_stretch_the_lazy_bounds(built, monkeypatch):
    """The 30-bit kernel keeps values lazily reduced through a pass (sums double, differences carry a bias of 2^12 r) and
    closes a forward transform with fr30_reduce_lazy, one quotient estimate from the top limb (fr30.hpp).  Inputs built
    to push those values as far as they go -- all r - 1, alternating r - 1 / 0 / 1 at every stride, half-and-half blocks,
    a single r - 1 in a field of zeros, ramps -- must give the words the 8 x 32 kernel gives, forward, coset-forward and
    inverse, at every pass structure (one to three passes, the two-pass 2^20 form, odd and even stage counts)."""
    import typlonk_amd

    rm1 = np.array(O.fr_to_mont_limbs(O.R - 1), dtype=np.uint64)
    one = np.array(O.fr_to_mont_limbs(1), dtype=np.uint64)
    g = np.array(O.fr_to_mont_limbs(7), dtype=np.uint64)

    def patterns(n):
        idx = np.arange(n)
        out = [np.tile(rm1, (n, 1))]
        for stride in (1, 2, 4, max(1, n // 4), max(1, n // 2)):
            for other in (np.zeros(4, dtype=np.uint64), one):
                x = np.tile(rm1, (n, 1))
                x[(idx // stride) % 2 == 1] = other
                out.append(x)
        x = np.zeros((n, 4), dtype=np.uint64)
        x[n // 3] = rm1
        out.append(x)
        ramp = np.zeros((n, 4), dtype=np.uint64)
        ramp[:, 0] = idx.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15 & 0x7FFFFFFFFFFFFFFF) % np.uint64(1 << 62)
        ramp[:, 3] = np.uint64(0x3FFFFFFFFFFFFFFF)          # just below 2^254
        out.append(ramp)
        return out

    res = {}
    for mode in (0, 2):
        monkeypatch.setenv("TYPLONK_NTT_FR30", str(mode))
        c2 = typlonk_amd.Context(0)
        try:
            for log_n in (3, 8, 10, 11, 13, 16, 17, 19, 20, 21):
                for k, x in enumerate(patterns(1 << log_n)):
                    res[(mode, log_n, k, "f")] = c2.ntt(x, log_n)
                    res[(mode, log_n, k, "c")] = c2.ntt(x, log_n, coset=g)
                    res[(mode, log_n, k, "i")] = c2.ntt(x, log_n, inverse=True)
        finally:
            c2.close()
    for (mode, log_n, k, d), v in res.items():
        if mode == 0:
            assert (v == res[(2, log_n, k, d)]).all(), (log_n, k, d)
    # and one of them against the CPU restatement, so that the agreement is not two kernels sharing a mistake
    from oracle import coracle as CO

    for log_n in (10, 16):
        x = patterns(1 << log_n)[0]
        assert (res[(2, log_n, 0, "f")] == CO.ntt(x, log_n)).all()
