"""GPU parity of the MSM path (typlonk_msm_g1*) against the CPU oracle, through the C ABI.
Small sizes: the oracle's reference-faithful naive MSM (kzg/src/lib.rs:41-54).  Larger sizes: the
reference's own test identity commit(p) == [p(s)]G (kzg/src/lib.rs:102-105)."""
import numpy as np
import pytest

from helpers import O, fr_pack, g1_pack, g1_unpack_one

pytestmark = pytest.mark.gpu

_srs_cache = {}


def srs_points(s, length):
    key = (s, length)
    if key not in _srs_cache:
        _srs_cache[key] = O.srs_from_secret_fast(s, length)
    return _srs_cache[key]


def load_srs(ctx, s, length):
    pts = srs_points(s, length)
    xy, inf = g1_pack(pts)
    return ctx.srs_load(xy, inf), pts


def commit_identity(coeffs, s):
    return O.g1_mul(O.G1, O.poly_eval(coeffs, s))


def test_kat1_commit_17G(ctx):
    """kzg/src/lib.rs:95-109: s = 2, p = 1 + 2X + 3X^2 -> 17*G (literal coordinates, golden)."""
    sid, pts = load_srs(ctx, 2, 13)
    out, inf = ctx.msm(sid, fr_pack([1, 2, 3]))
    got = g1_unpack_one(out, inf)
    assert got == (
        0x1098F178F84FC753A76BB63709E9BE91EEC3FF5F7F3A5F4836F34FE8A1A6D6C5578D8FD820573CEF3A01E2BFEF3EAF3A,
        0x0EA923110B733B531006075F796CC9368F2477FE26020F465468EFBB380CE1F8EEBAF5C770F31D320F9BD378DC758436,
    )
    # open at z = 1: q = 3X + 5 -> 11 * G
    q, y = O.poly_div_linear([1, 2, 3], 1)
    assert y == 6 and q == [5, 3]
    out, inf = ctx.msm(sid, fr_pack(q))
    assert g1_unpack_one(out, inf) == O.g1_mul(O.G1, 11)
    ctx.srs_free(sid)


@pytest.mark.parametrize("m", [0, 1, 2, 3, 8, 33, 100])
def test_small_vs_naive_oracle(ctx, m):
    s = 0x0123456789ABCDEF0123456789ABCDEF
    sid, pts = load_srs(ctx, s, 103)
    coeffs = O.random_frs(0x5EED + m, m)
    out, inf = ctx.msm(sid, fr_pack(coeffs) if m else np.zeros((0, 4), dtype=np.uint64), m)
    got = g1_unpack_one(out, inf)
    assert got == O.msm_naive(coeffs, pts)
    if m == 0:
        # identity encoding: x = 0, y = Montgomery one, inf = 1 (ark-ec GroupAffine::zero())
        assert inf == 1 and not out[:6].any() and [int(x) for x in out[6:]] == O.fq_to_mont_limbs(1)
    ctx.srs_free(sid)


@pytest.mark.parametrize("m", [1000, 4093, (1 << 14) - 1, 1 << 14])
def test_commit_identity(ctx, m):
    sid, pts = load_srs(ctx, 2, (1 << 14) + 3)
    coeffs = O.random_frs(0x5EED + m, m)
    out, inf = ctx.msm(sid, fr_pack(coeffs))
    assert g1_unpack_one(out, inf) == commit_identity(coeffs, 2)
    ctx.srs_free(sid)


@pytest.mark.parametrize("name", ["zeros", "ones", "r_minus_1", "single", "alternating", "short", "repeated"])
def test_adversarial_scalars(ctx, name):
    m = 1000
    sid, pts = load_srs(ctx, 2, 1003)
    rep = O.random_frs(42, 1)[0]
    coeffs = {
        "zeros": [0] * m,
        "ones": [1] * m,
        "r_minus_1": [O.R - 1] * m,
        "single": [0] * 500 + [rep] + [0] * 499,
        "alternating": [0 if i % 2 else O.R - 1 for i in range(m)],
        "short": [x & 0xFFFF for x in O.random_frs(43, m)],
        "repeated": [rep] * m,
    }[name]
    out, inf = ctx.msm(sid, fr_pack(coeffs))
    assert g1_unpack_one(out, inf) == commit_identity(coeffs, 2)
    ctx.srs_free(sid)


def test_srs_with_identity_points(ctx):
    """s = 0: SRS = [G, inf, inf, ...] (kzg/src/srs.rs:15-24 with s^i = 0)."""
    pts = [O.G1] + [None] * 40
    xy, inf = g1_pack(pts)
    sid = ctx.srs_load(xy, inf)
    coeffs = O.random_frs(5, 41)
    out, oinf = ctx.msm(sid, fr_pack(coeffs))
    assert g1_unpack_one(out, oinf) == O.g1_mul(O.G1, coeffs[0])
    ctx.srs_free(sid)


def test_srs_all_equal_points(ctx):
    """s = 1: every base is G, so buckets hit the P + P doubling branch of the mixed add."""
    pts = [O.G1] * 300
    xy, inf = g1_pack(pts)
    sid = ctx.srs_load(xy, None)
    coeffs = O.random_frs(6, 300)
    out, oinf = ctx.msm(sid, fr_pack(coeffs))
    assert g1_unpack_one(out, oinf) == O.g1_mul(O.G1, sum(coeffs) % O.R)
    # and P + (-P): scalars k, r-k on the same point cancel
    coeffs = [5, O.R - 5, 7, O.R - 7]
    out, oinf = ctx.msm(sid, fr_pack(coeffs))
    assert g1_unpack_one(out, oinf) is None and oinf == 1
    ctx.srs_free(sid)


def test_scalar_mul_homomorphism(ctx):
    """kzg/src/lib.rs:160-171: commit(9 p) == 9 commit(p)."""
    sid, pts = load_srs(ctx, 2, 103)
    p = O.random_frs(9, 8)
    out1, i1 = ctx.msm(sid, fr_pack([9 * c % O.R for c in p]))
    out2, i2 = ctx.msm(sid, fr_pack(p))
    assert g1_unpack_one(out1, i1) == O.g1_mul(g1_unpack_one(out2, i2), 9)
    ctx.srs_free(sid)


def test_length_error_matches_reference_assert(ctx):
    from typlonk_amd.capi import TyplonkError, ERR_LENGTH

    sid, pts = load_srs(ctx, 2, 13)
    with pytest.raises(TyplonkError) as e:
        ctx.msm(sid, fr_pack([1] * 14))
    assert e.value.code == ERR_LENGTH
    ctx.srs_free(sid)


def test_device_resident_intt_then_msm(ctx):
    """interpolate(evals) -> commit without leaving HBM (plonk/src/builder.rs:85-86 pattern)."""
    log_n, n = 10, 1 << 10
    sid, pts = load_srs(ctx, 2, n + 3)
    evals = O.random_frs(77, n)
    buf = ctx.alloc(n)
    buf.upload(fr_pack(evals))
    ctx.ntt_dev(buf, log_n, inverse=True)
    out, inf = ctx.msm_dev(sid, buf, 0, n)
    coeffs = O.ntt(evals, log_n, inverse=True)
    assert g1_unpack_one(out, inf) == commit_identity(coeffs, 2)
    buf.free()
    ctx.srs_free(sid)


def test_srs_generate_matches_reference_faithful_generator(ctx):
    """typlonk_srs_generate == Srs::from_secret (kzg/src/srs.rs:15-34) restated by the C oracle,
    including a shard that starts at a non-zero power."""
    from oracle import coracle as CO

    s = 0x0123456789ABCDEF0123456789ABCDEF
    s_limbs = np.array(O.fr_to_mont_limbs(s), dtype=np.uint64)
    ref_xy, ref_inf = CO.srs_from_secret(s_limbs, 40)
    sid = ctx.srs_generate(s_limbs, 40)
    xy, inf = ctx.srs_download(sid)
    assert (xy == ref_xy).all() and (inf == ref_inf).all()
    ctx.srs_free(sid)
    sid = ctx.srs_generate(s_limbs, 15, start=25)
    xy, inf = ctx.srs_download(sid)
    assert (xy == ref_xy[25:]).all()
    ctx.srs_free(sid)
    # s = 0 -> [G, inf, inf, ...]
    sid = ctx.srs_generate(np.zeros(4, dtype=np.uint64), 4)
    xy, inf = ctx.srs_download(sid)
    assert list(inf) == [0, 1, 1, 1] and g1_unpack_one(xy[0], 0) == O.G1
    assert not xy[1, :6].any() and [int(x) for x in xy[1, 6:]] == O.fq_to_mont_limbs(1)
    ctx.srs_free(sid)


@pytest.mark.parametrize("log_m,delta", [(16, 0), (16, -1), (20, 0), (20, -3), (20, -1), (22, 0), (22, -3)])
def test_commit_identity_large(ctx, log_m, delta):
    """BASELINE configs 2-3 sizes (and the n-1 / n-3 lengths prove() uses): commit(p) == [p(s)]G,
    s = 2, SRS generated on the device, checked with the C oracle's Horner + scalar mul."""
    from oracle import coracle as CO

    n = 1 << log_m
    m = n + delta
    s_limbs = np.array(O.fr_to_mont_limbs(2), dtype=np.uint64)
    sid = ctx.srs_generate(s_limbs, n + 3)
    rng = np.random.default_rng(log_m)
    sc = rng.integers(0, 1 << 63, size=(m, 4), dtype=np.uint64) * 2 + rng.integers(0, 2, size=(m, 4), dtype=np.uint64)
    sc[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
    out, inf = ctx.msm(sid, sc)
    exp, einf = CO.g1_mul_generator(CO.poly_eval(sc, s_limbs))
    assert (out == exp).all() and inf == einf
    if log_m >= 20:
        # the same with fixed-base tables (13 pre-shifted copies of the SRS): the configuration bench.py and the prover use,
        # stand-alone (chunked) and inside a batch (not chunked)
        ctx.srs_precompute(sid, 20)
        out, inf = ctx.msm(sid, sc)
        assert (out == exp).all() and inf == einf
        buf = ctx.alloc(m)
        buf.upload(sc)
        res = ctx.msm_batch_devptr(sid, [buf.devptr, buf.devptr, buf.devptr], [m, m - 1, m])
        assert (res[0][0] == exp).all() and res[0][1] == einf and (res[2][0] == exp).all()
        exp1, einf1 = CO.g1_mul_generator(CO.poly_eval(sc[:m - 1], s_limbs))
        assert (res[1][0] == exp1).all() and res[1][1] == einf1
        buf.free()
    ctx.srs_free(sid)


def test_golden_msm_fixtures(ctx):
    from helpers import hex_pt, load_golden

    for case in load_golden("msm.json"):
        s, n = int(case["secret"], 16), case["srs_len"]
        sc = [int(x, 16) for x in case["scalars"]]
        sid = ctx.srs_generate(np.array(O.fr_to_mont_limbs(s), dtype=np.uint64), n)
        out, inf = ctx.msm(sid, fr_pack(sc) if sc else np.zeros((0, 4), dtype=np.uint64), len(sc))
        assert g1_unpack_one(out, inf) == hex_pt(case["expected"])
        ctx.srs_free(sid)


def test_batch_matches_individual_msms(ctx):
    """typlonk_msm_g1_batch_devptr (two MSMs in flight) == the same MSMs issued one by one,
    for the mixed lengths prove() uses (n, n-1, n-3) plus the empty polynomial."""
    n = 1 << 12
    s_limbs = np.array(O.fr_to_mont_limbs(2), dtype=np.uint64)
    sid = ctx.srs_generate(s_limbs, n + 3)
    rng = np.random.default_rng(77)
    bufs, ms = [], [n, n - 1, 0, n - 3, n, 17, n - 1]
    for m in ms:
        sc = rng.integers(0, 1 << 63, size=(max(m, 1), 4), dtype=np.uint64)
        sc[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
        b = ctx.alloc(max(m, 1))
        b.upload(sc)
        bufs.append((b, sc))
    single = [ctx.msm_devptr(sid, b.devptr, m) for (b, _), m in zip(bufs, ms)]
    batch = ctx.msm_batch_devptr(sid, [b.devptr for b, _ in bufs], ms)
    for (sx, si), (bx, bi) in zip(single, batch):
        assert (sx == bx).all() and si == bi
    assert batch[2][1] == 1
    # and against the oracle identity for one of them
    from oracle import coracle as CO
    exp, einf = CO.g1_mul_generator(CO.poly_eval(bufs[1][1][: n - 1], s_limbs))
    assert (batch[1][0] == exp).all() and batch[1][1] == einf
    for b, _ in bufs:
        b.free()
    ctx.srs_free(sid)


@pytest.mark.parametrize("chain,inflight", [("1", "3"), ("0", "4"), ("1", "1"), ("0", "2"), ("1", "4")])
def test_batch_scheduling_variants_give_identical_points(built, chain, inflight, monkeypatch):
    """How a batch is scheduled -- accumulations chained one after the other across the lanes or free-running
    (TYPLONK_MSM_CHAIN), one to four MSMs in flight (TYPLONK_MSM_INFLIGHT) -- must not change a bit: nine table-mode
    MSMs at 2^16 (prove()'s lengths n, n - 1, n - 3, one empty) against commit(p) == [p(s)]G (kzg/src/lib.rs:102-105),
    twice in a row on the same context (lanes and the chain event are reused)"""
    import typlonk_amd
    from oracle import coracle as CO

    monkeypatch.setenv("TYPLONK_MSM_CHAIN", chain)
    monkeypatch.setenv("TYPLONK_MSM_INFLIGHT", inflight)
    c2 = typlonk_amd.Context(0)
    try:
        n = 1 << 16
        s_limbs = np.array(O.fr_to_mont_limbs(0x5EED), dtype=np.uint64)
        sid = c2.srs_generate(s_limbs, n + 3)
        c2.srs_precompute(sid, 0)
        rng = np.random.default_rng(int(chain) * 10 + int(inflight))
        ms = [n, n - 1, n - 1, 0, n - 3, n, n - 1, 4097, n]
        bufs, exp = [], []
        for m in ms:
            sc = rng.integers(0, 1 << 63, size=(max(m, 1), 4), dtype=np.uint64) * 2
            sc[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
            b = c2.alloc(max(m, 1))
            b.upload(sc)
            bufs.append(b)
            exp.append(CO.g1_mul_generator(CO.poly_eval(sc[:m], s_limbs)) if m else (None, 1))
        for _ in range(2):
            got = c2.msm_batch_devptr(sid, [b.devptr for b in bufs], ms)
            for (gx, gi), (ex, ei), m in zip(got, exp, ms):
                assert gi == ei and (m == 0 or (gx == ex).all()), (chain, inflight, m)
            # a stand-alone MSM between two batches (does not take part in the chain)
            one = c2.msm_devptr(sid, bufs[0].devptr, n)
            assert (one[0] == exp[0][0]).all() and one[1] == exp[0][1]
        for b in bufs:
            b.free()
    finally:
        c2.close()


@pytest.mark.parametrize("name", ["ones", "repeated", "alternating", "short", "two_values"])
def test_adversarial_scalars_at_2_16_use_heavy_bucket_tasks(ctx, name):
    """SURVEY section 8d adversarial sets at a size where a single bucket receives up to m entries: the
    heavy-bucket task splitting must give the exact result (and finish quickly)."""
    import time
    from oracle import coracle as CO

    m = 1 << 16
    s_limbs = np.array(O.fr_to_mont_limbs(2), dtype=np.uint64)
    sid = ctx.srs_generate(s_limbs, m + 3)
    rep = O.random_frs(4242, 2)
    vals = {
        "ones": lambda i: 1,
        "repeated": lambda i: rep[0],
        "alternating": lambda i: 0 if i % 2 else O.R - 1,
        "short": lambda i: (i * 2654435761) & 0xFF,
        "two_values": lambda i: rep[i % 2],
    }[name]
    uniq = {}
    sc = np.empty((m, 4), dtype=np.uint64)
    for i in range(m):
        v = vals(i)
        if v not in uniq:
            uniq[v] = O.fr_to_mont_limbs(v)
        sc[i] = uniq[v]
    t0 = time.time()
    out, inf = ctx.msm(sid, sc)
    dt = time.time() - t0
    exp, einf = CO.g1_mul_generator(CO.poly_eval(sc, s_limbs))
    assert (out == exp).all() and inf == einf
    assert dt < 5.0, f"adversarial MSM took {dt:.1f} s"
    ctx.srs_free(sid)


@pytest.mark.parametrize("c", [16, 17, 18, 19, 20])
def test_fixed_base_tables_give_identical_results(ctx, c):
    """typlonk_srs_precompute: the table-mode MSM (shared bucket set, no window recombination) returns
    exactly the same points as the plain path, for full and shorter lengths, incl. adversarial sets"""
    from oracle import coracle as CO

    n = 1 << 13
    s_limbs = np.array(O.fr_to_mont_limbs(0x0123456789ABCDEF0123456789ABCDEF), dtype=np.uint64)
    sid_plain = ctx.srs_generate(s_limbs, n + 3)
    sid_tab = ctx.srs_generate(s_limbs, n + 3)
    ctx.srs_precompute(sid_tab, c)
    xy, inf = ctx.srs_download(sid_tab)                 # table 0 is still the SRS itself
    xy0, inf0 = ctx.srs_download(sid_plain)
    assert (xy == xy0).all() and (inf == inf0).all()
    rng = np.random.default_rng(c)
    for m in (n, n - 1, n - 3, n // 2 + 5, 100):        # 100 < len/4 falls back to the plain path on table 0
        sc = rng.integers(0, 1 << 63, size=(m, 4), dtype=np.uint64) * 2 + rng.integers(0, 2, size=(m, 4), dtype=np.uint64)
        sc[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
        a, ai = ctx.msm(sid_plain, sc)
        b, bi = ctx.msm(sid_tab, sc)
        assert (a == b).all() and ai == bi
    one = np.tile(np.array(O.fr_to_mont_limbs(1), dtype=np.uint64), (n, 1))
    rm1 = np.tile(np.array(O.fr_to_mont_limbs(O.R - 1), dtype=np.uint64), (n, 1))
    for sc in (one, rm1, np.zeros((n, 4), dtype=np.uint64)):
        a, ai = ctx.msm(sid_plain, sc)
        b, bi = ctx.msm(sid_tab, sc)
        assert (a == b).all() and ai == bi
    # against the oracle once (reference-faithful per-term MSM on a slice)
    sc = rng.integers(0, 1 << 62, size=(2048, 4), dtype=np.uint64)
    sc[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
    sid_small = ctx.srs_generate(s_limbs, 2048)
    ctx.srs_precompute(sid_small, c)
    b, bi = ctx.msm(sid_small, sc)
    r, ri = CO.msm_reference(sc, xy[:2048], inf[:2048])
    assert (b == r).all() and bi == ri
    for sid in (sid_plain, sid_tab, sid_small):
        ctx.srs_free(sid)


def test_device_inversion_divsteps_equals_fermat_ladder(ctx):
    """the SIMT inversion the kernels use (Bernstein-Yang divsteps, fq30.hpp) against the Fermat ladder a^(p-2) on the
    device: 2^20 random and edge residues (0, 1, 2, p-1, p-2, one-limb values), lazily reduced up to 8p, digit for digit,
    and x * x^-1 = 1; no call runs more 30-divstep rounds than the proven bound"""
    bad, rounds = ctx.selftest_fq_inv(1 << 20, seed=0x5EED)
    assert bad == 0
    assert 1 <= rounds <= 37
    bad, rounds = ctx.selftest_fq_inv(1 << 16, seed=7)
    assert bad == 0 and rounds <= 37


@pytest.mark.parametrize("c", [15, 20])
def test_fixed_base_tables_with_identity_bases(ctx, c):
    """the table walk shares ONE inversion per base across its column of entries (srs_gen.hip): identity bases in the
    SRS -- s = 0 gives (G, inf, inf, ...), and a loaded vector with identities in the middle -- must stay identities in
    every table and must not disturb their neighbours"""
    from oracle import coracle as CO

    n = 1 << 14
    rng = np.random.default_rng(100 + c)
    sc = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    sc[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
    # s = 0
    z = ctx.srs_generate(np.zeros(4, dtype=np.uint64), n)
    zt = ctx.srs_generate(np.zeros(4, dtype=np.uint64), n)
    ctx.srs_precompute(zt, c)
    a, ai = ctx.msm(z, sc)
    b, bi = ctx.msm(zt, sc)
    assert (a == b).all() and ai == bi
    # identities sprinkled into a real SRS (every 7th base, the first and the last)
    s_limbs = np.array(O.fr_to_mont_limbs(0xABCDEF0123456789), dtype=np.uint64)
    g = ctx.srs_generate(s_limbs, n)
    xy, inf = ctx.srs_download(g)
    inf = np.ascontiguousarray(inf).copy()
    xy = np.ascontiguousarray(xy).copy()
    holes = np.zeros(n, dtype=bool)
    holes[::7] = True
    holes[-1] = True
    inf[holes] = 1
    xy[holes] = 0
    xy[holes, 6:] = np.array(O.fq_to_mont_limbs(1), dtype=np.uint64)   # ark-ec identity: (0, 1, inf)
    plain = ctx.srs_load(xy, inf)
    tab = ctx.srs_load(xy, inf)
    ctx.srs_precompute(tab, c)
    a, ai = ctx.msm(plain, sc)
    b, bi = ctx.msm(tab, sc)
    assert (a == b).all() and ai == bi
    r, ri = CO.msm_reference(sc[:1024], xy[:1024], inf[:1024])
    small = ctx.srs_load(xy[:1024], inf[:1024])
    got, gi = ctx.msm(small, sc[:1024])
    assert (got == r).all() and gi == ri
    for sid in (z, zt, g, plain, tab, small):
        ctx.srs_free(sid)


def test_fixed_base_tables_commit_identity_2_18(ctx):
    """table mode (c = 20, the shape bench.py uses) at 2^18 terms: commit(p) == [p(s)]G, plus the batch
    entry point and the n-1 / n-3 lengths of prove()"""
    from oracle import coracle as CO

    n = 1 << 18
    s_limbs = np.array(O.fr_to_mont_limbs(2), dtype=np.uint64)
    sid = ctx.srs_generate(s_limbs, n + 3)
    ctx.srs_precompute(sid, 20)
    rng = np.random.default_rng(2018)
    sc = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * 2 + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    sc[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
    buf = ctx.alloc(n)
    buf.upload(sc)
    res = ctx.msm_batch_devptr(sid, [buf.devptr] * 3, [n, n - 1, n - 3])
    for (out, inf), m in zip(res, (n, n - 1, n - 3)):
        exp, einf = CO.g1_mul_generator(CO.poly_eval(sc[:m], s_limbs))
        assert (out == exp).all() and inf == einf
    buf.free()
    ctx.srs_free(sid)


def _mixed_scalars(rng, m):
    """a random blend of the scalar classes the reference can meet: uniform, zero, one, r - 1, tiny, one repeated value"""
    sc = rng.integers(0, 1 << 63, size=(m, 4), dtype=np.uint64) * 2 + rng.integers(0, 2, size=(m, 4), dtype=np.uint64)
    sc[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)          # < 2^254 < r: every limb pattern is a valid Montgomery residue
    if m == 0:
        return sc
    cls = rng.integers(0, 6, size=m)
    const = {1: 0, 2: O.fr_to_mont_limbs(1), 3: O.fr_to_mont_limbs(O.R - 1), 5: O.fr_to_mont_limbs(int(rng.integers(2, 1 << 40)))}
    for k, v in const.items():
        sc[cls == k] = np.array(v if v else [0, 0, 0, 0], dtype=np.uint64)
    tiny = np.nonzero(cls == 4)[0]
    for i in tiny:
        sc[i] = np.array(O.fr_to_mont_limbs(int(rng.integers(0, 1 << 16))), dtype=np.uint64)
    return sc


@pytest.mark.parametrize("seed", range(6))
def test_random_shapes_against_the_bucket_method_oracle(ctx, seed):
    """random (SRS length, m, shard split, tables on/off, scalar mix): every variant of the device path equals the
    CPU bucket-method MSM (oracle_msm_pippenger, itself checked against the reference-faithful path)"""
    from oracle import coracle as CO

    rng = np.random.default_rng(1000 + seed)
    length = int(rng.integers(1, 40000)) if seed % 2 else int(rng.integers(1 << 15, 1 << 16))
    s_limbs = np.array(O.fr_to_mont_limbs(int(rng.integers(2, 1 << 62))), dtype=np.uint64)
    sid = ctx.srs_generate(s_limbs, length)
    xy, inf = ctx.srs_download(sid)
    tab = ctx.srs_generate(s_limbs, length)
    ctx.srs_precompute(tab, 16 + seed % 5)
    # the same SRS as two shards of unequal size
    cut = int(rng.integers(0, length + 1))
    lo_id = ctx.srs_generate(s_limbs, cut, start=0) if cut else None
    hi_id = ctx.srs_generate(s_limbs, length - cut, start=cut) if cut < length else None
    if lo_id is not None:
        ctx.srs_set_shard(lo_id, 0, length)
    if hi_id is not None:
        ctx.srs_set_shard(hi_id, cut, length)
    from typlonk_amd.capi import g1_sum_host
    for m in sorted({0, 1, length, int(rng.integers(0, length + 1)), max(0, cut - 1), min(length, cut + 1)}):
        sc = _mixed_scalars(rng, m)
        exp, einf, _, _ = CO.msm_pippenger(sc, xy, inf, c=11)
        for handle in (sid, tab):
            got, ginf = ctx.msm(handle, sc)
            assert (got == exp).all() and ginf == einf, (length, m, handle == tab)
        parts = [ctx.msm(h, sc) for h in (lo_id, hi_id) if h is not None]
        fxy, finf = g1_sum_host(np.stack([p[0] for p in parts]), np.array([p[1] for p in parts], dtype=np.uint8))
        assert (fxy == exp).all() and finf == einf, (length, m, cut)
    for h in (sid, tab, lo_id, hi_id):
        if h is not None:
            ctx.srs_free(h)


def test_shard_argument_errors(ctx):
    """typlonk_srs_set_shard: the slice must fit into total_len; the MSM length is checked against the TOTAL length
    (the reference's assert, kzg/src/lib.rs:43), not the local one"""
    from typlonk_amd.capi import TyplonkError, ERR_INVALID_ARG, ERR_LENGTH, ERR_RANGE

    s_limbs = np.array(O.fr_to_mont_limbs(5), dtype=np.uint64)
    sid = ctx.srs_generate(s_limbs, 10, start=20)
    with pytest.raises(TyplonkError) as e:
        ctx.srs_set_shard(sid, 25, 30)          # 25 + 10 > 30
    assert e.value.code == ERR_RANGE
    with pytest.raises(TyplonkError) as e:
        ctx.srs_set_shard(9999, 0, 10)
    assert e.value.code == ERR_INVALID_ARG
    ctx.srs_set_shard(sid, 20, 40)
    sc = np.tile(np.array(O.fr_to_mont_limbs(3), dtype=np.uint64), (41, 1))
    with pytest.raises(TyplonkError) as e:
        ctx.msm(sid, sc)                         # 41 > total_len
    assert e.value.code == ERR_LENGTH
    out, inf = ctx.msm(sid, sc[:15])             # m below the shard's first index: the empty sum
    assert inf == 1
    out, inf = ctx.msm(sid, sc[:40])             # 3 * sum_{i=20}^{29} 5^i G
    exp = O.g1_mul(O.G1, 3 * sum(pow(5, i, O.R) for i in range(20, 30)) % O.R)
    assert g1_unpack_one(out, inf) == exp
    ctx.srs_free(sid)


@pytest.mark.parametrize("chunks", [1, 2, 3, 5, 8])
def test_chunked_pipeline_gives_identical_results(built, chunks, monkeypatch):
    """A stand-alone MSM is cut into chunks of terms that add into the same buckets (sort of chunk k + 1 overlapping
    the accumulation of chunk k, msm_host.hip msm_enqueue).  Every chunk count -- forced through TYPLONK_MSM_CHUNKS --
    gives the bit-identical point, with and without fixed-base tables, for uniform and adversarial scalars (heavy
    buckets cross chunk boundaries), full and ragged lengths, and on an SRS shard."""
    import typlonk_amd
    from oracle import coracle as CO

    monkeypatch.setenv("TYPLONK_MSM_CHUNKS", str(chunks))
    c2 = typlonk_amd.Context(0)
    try:
        length = (1 << 16) + 77
        secret = 0xABCDEF0123
        s_limbs = np.array(O.fr_to_mont_limbs(secret), dtype=np.uint64)
        plain = c2.srs_generate(s_limbs, length)
        tab = c2.srs_generate(s_limbs, length)
        c2.srs_precompute(tab, 20)
        shard = c2.srs_generate(s_limbs, length - 1000, start=1000)
        c2.srs_set_shard(shard, 1000, length)
        rng = np.random.default_rng(4242 + chunks)
        for m in (length, length - 3, 40000, 4097 * chunks + 1):
            for kind in ("uniform", "mixed", "ones"):
                if kind == "uniform":
                    sc = rng.integers(0, 1 << 63, size=(m, 4), dtype=np.uint64) * 2
                    sc[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
                elif kind == "mixed":
                    sc = _mixed_scalars(rng, m)
                else:
                    sc = np.tile(np.array(O.fr_to_mont_limbs(1), dtype=np.uint64), (m, 1))
                ps = CO.poly_eval(sc, s_limbs)                      # commit(p) == [p(s)]G, kzg/src/lib.rs:102-105
                exp_xy, exp_inf = CO.g1_mul_generator(ps)
                for h in (plain, tab):
                    got, ginf = c2.msm(h, sc)
                    assert (got == exp_xy).all() and ginf == exp_inf, (chunks, m, kind, h == tab)
                # the shard's partial sum = the full sum minus the first 1000 terms
                part, pinf = c2.msm(shard, sc)
                head, hinf = c2.msm(plain, sc[:1000])
                from typlonk_amd.capi import g1_sum_host
                fxy, finf = g1_sum_host(np.stack([part, head]), np.array([pinf, hinf], dtype=np.uint8))
                assert (fxy == exp_xy).all() and finf == exp_inf, (chunks, m, kind, "shard")
    finally:
        c2.close()


@pytest.mark.parametrize("scatter,l1,extra", [
    ("staged", "512", {}), ("staged", "256", {}), ("direct", "512", {}),
    # an unequal first chunk of a stand-alone MSM and the two-launch reduction with two wavefronts per SIMD on 2^19 buckets
    # (measured and not adopted: profiles/r06_ab_first_chunk_and_rc2.txt; the switches stay, so they stay tested)
    ("staged", "512", {"TYPLONK_MSM_FIRST_PCT": "30", "TYPLONK_MSM_REDUCE": "rc2", "TYPLONK_MSM_RC2_LOGW": "11"}),
])
def test_every_form_of_the_bucket_sort_gives_the_same_point(built, scatter, l1, extra, monkeypatch):
    """Round 6 rebuilt the bucket sort: level 1 stages its runs in the LDS (TYPLONK_MSM_SCATTER=direct keeps the rounds 1-5 form,
    also the fallback for shapes whose staging area does not fit), 256 or 512 threads per level-1 workgroup, level 2 assembles
    a segment's output in the LDS unless the segment is longer than its staging array.  Every form, at the table-mode sizes
    the prover uses -- 2^19 + 5 terms (a chunk: 2048 segments), 2^20 - 3 (a queued MSM: 4096 segments), 2^20 + 2^19 + 1 (three
    chunks), an 8-way shard's 2^17 (c = 17) -- must give commit(p) == [p(s)]G (kzg/src/lib.rs:102-105), for uniform scalars
    and for the sets that overflow a segment: all scalars equal (ONE bucket per window takes everything: a segment of 2^20
    entries, far beyond the level-2 staging array, and heavy-bucket tasks), two values, tiny scalars, alternating 0 / r - 1."""
    import typlonk_amd
    from oracle import coracle as CO

    monkeypatch.setenv("TYPLONK_MSM_SCATTER", scatter)
    monkeypatch.setenv("TYPLONK_MSM_L1_THREADS", l1)
    for k, v in extra.items():
        monkeypatch.setenv(k, v)
    c2 = typlonk_amd.Context(0)
    try:
        length = (1 << 20) + (1 << 19) + 4
        s_limbs = np.array(O.fr_to_mont_limbs(0x1357_9BDF_2468), dtype=np.uint64)
        tab = c2.srs_generate(s_limbs, length)
        c2.srs_precompute(tab, 20)
        small = c2.srs_generate(s_limbs, (1 << 17) + 9)
        c2.srs_precompute(small, 0)                                   # the library's choice for a shard: c = 17
        rng = np.random.default_rng(int(l1) + len(scatter) + len(extra))
        rep = [np.array(O.fr_to_mont_limbs(v), dtype=np.uint64) for v in O.random_frs(99, 2)]
        rm1 = np.array(O.fr_to_mont_limbs(O.R - 1), dtype=np.uint64)

        def scalars(kind, m):
            if kind == "uniform":
                sc = rng.integers(0, 1 << 63, size=(m, 4), dtype=np.uint64) * 2
                sc[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
                return sc
            if kind == "equal":
                return np.tile(rep[0], (m, 1))
            if kind == "two_values":
                sc = np.tile(rep[0], (m, 1))
                sc[1::2] = rep[1]
                return sc
            if kind == "tiny":
                sc = np.zeros((m, 4), dtype=np.uint64)
                small_vals = {v: np.array(O.fr_to_mont_limbs(v), dtype=np.uint64) for v in range(16)}
                for v in range(16):
                    sc[v::16] = small_vals[v]
                return sc
            sc = np.zeros((m, 4), dtype=np.uint64)                    # alternating 0 / r - 1
            sc[0::2] = rm1
            return sc

        cases = [(tab, (1 << 19) + 5, ("uniform", "equal", "tiny")), (tab, (1 << 20) - 3, ("uniform", "two_values", "alternating")),
                 (tab, length - 3, ("uniform", "equal")), (small, 1 << 17, ("uniform", "equal", "alternating"))]
        for sid, m, kinds in cases:
            for kind in kinds:
                sc = scalars(kind, m)
                exp_xy, exp_inf = CO.g1_mul_generator(CO.poly_eval(sc, s_limbs))
                got, ginf = c2.msm(sid, sc)
                assert (got == exp_xy).all() and ginf == exp_inf, (scatter, l1, m, kind)
                if m <= (1 << 20) and kind != "tiny":                  # the same MSM queued (batch form: one launch, 4096 segments)
                    buf = c2.alloc(m)
                    buf.upload(sc)
                    outs = c2.msm_batch_devptr(sid, [buf.devptr, buf.devptr], [m, m - 1])
                    assert (np.asarray(outs[0][0]) == exp_xy).all() and outs[0][1] == exp_inf, (scatter, l1, m, kind, "queued")
                    buf.free()
    finally:
        c2.close()


@pytest.mark.slow
def test_more_than_2_23_terms_takes_the_counting_sort_fallback(ctx):
    """The segmented sort indexes terms with 23 bits; a longer MSM falls back to the global counting sort (msm_host.hip
    msm_enqueue -- the path has no switch of its own since round 4, so this is the test that reaches it): 2^23 + 5 terms
    over [s^i]G against commit(p) == [p(s)]G (kzg/src/lib.rs:102-105), with the ragged lengths m - 1 and m - 3, and a
    chunk-sized control below the limit through the ordinary path over the same SRS."""
    import torch
    from conftest import need_resources
    from oracle import coracle as CO

    need_resources(host_gib=3, hbm_gib=6)
    m = (1 << 23) + 5
    s_limbs = np.array(O.fr_to_mont_limbs(2), dtype=np.uint64)
    sid = ctx.srs_generate(s_limbs, m)
    rng = np.random.default_rng(823)
    sc = rng.integers(0, 1 << 63, size=(m, 4), dtype=np.uint64) * 2
    sc[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
    buf = ctx.alloc(m)
    buf.upload(sc)
    for mm in (m, m - 1, m - 3, (1 << 23) - 1):
        exp_xy, exp_inf = CO.g1_mul_generator(CO.poly_eval(sc[:mm], s_limbs))
        got, ginf = ctx.msm_devptr(sid, buf.devptr, mm)
        assert (got == exp_xy).all() and ginf == exp_inf, mm
    buf.free()
    ctx.srs_free(sid)
    torch.cuda.empty_cache()


def test_one_context_per_thread_two_threads_at_once(built):
    """The threading contract of include/typlonk.h ("one host thread per context") as the Rust layer uses it
    (integration/rust/kzg_hip: Backend::shared keeps one context per thread and device): two threads, each with its OWN
    context on the same GPU, generate an SRS, build tables, commit (stand-alone MSMs of 2^17 + 5 and 2^20 + 1 terms, a batch of
    four) and transform (single and grouped) AT THE SAME TIME -- ctypes releases the interpreter lock around every call -- and
    every result equals the one a single thread computed before."""
    import threading
    import torch
    import typlonk_amd

    dev = torch.device("cuda", 0)
    secret = np.array(O.fr_to_mont_limbs(0x51DE_CAFE), dtype=np.uint64)
    length = (1 << 20) + 4
    rng = np.random.default_rng(77)

    def uniform(m):
        sc = rng.integers(0, 1 << 63, size=(m, 4), dtype=np.uint64) * 2
        sc[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
        return sc

    work = [{"msm": [uniform((1 << 17) + 5), uniform((1 << 20) + 1)], "batch": [uniform(1 << 16) for _ in range(4)],
             "ntt": uniform(1 << 18), "group": [uniform(1 << 14) for _ in range(3)]} for _ in range(2)]

    def run(w, out):
        c = typlonk_amd.Context(0)
        try:
            sid = c.srs_generate(secret, length)
            c.srs_precompute(sid, 20)
            res = {"msm": [], "rounds": 0}
            for _ in range(3):                                         # several rounds, so that the threads really overlap
                res["msm"] = [c.msm(sid, s) for s in w["msm"]]
                vecs = [torch.from_numpy(np.ascontiguousarray(s).view(np.int64)).to(dev) for s in w["batch"]]
                torch.cuda.synchronize()
                res["batch"] = c.msm_batch_devptr(sid, [v.data_ptr() for v in vecs], [len(s) for s in w["batch"]])
                res["ntt"] = c.ntt(w["ntt"], 18, inverse=True)
                grp = [torch.from_numpy(np.ascontiguousarray(s).view(np.int64)).to(dev) for s in w["group"]]
                torch.cuda.synchronize()
                c.ntt_batch_devptr([g.data_ptr() for g in grp], 14)
                torch.cuda.synchronize()
                res["group"] = [g.cpu().numpy().view(np.uint64) for g in grp]
                res["rounds"] += 1
            out.update(res)
        except BaseException as e:   # noqa: BLE001 -- reported by the asserting thread
            out["error"] = repr(e)
        finally:
            c.close()

    alone = [{}, {}]
    for w, o in zip(work, alone):
        run(w, o)
    together = [{}, {}]
    threads = [threading.Thread(target=run, args=(w, o)) for w, o in zip(work, together)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    for a, b in zip(alone, together):
        assert "error" not in a and "error" not in b, (a.get("error"), b.get("error"))
        assert b["rounds"] == 3
        for (axy, ainf), (bxy, binf) in zip(a["msm"] + list(a["batch"]), b["msm"] + list(b["batch"])):
            assert int(ainf) == int(binf) == 0 and (np.asarray(axy) == np.asarray(bxy)).all()
        assert (a["ntt"] == b["ntt"]).all()
        assert all((x == y).all() for x, y in zip(a["group"], b["group"]))


def _fr_add_mod_limbs(a, b):
    """(a + b) mod r on (m, 4) little-endian u64 limb arrays, vectorised (both inputs < r)"""
    r_limbs = [np.uint64((O.R >> (64 * i)) & 0xFFFFFFFFFFFFFFFF) for i in range(4)]
    s = np.empty_like(a)
    carry = np.zeros(a.shape[0], dtype=np.uint64)
    for i in range(4):
        t = a[:, i] + b[:, i]
        c1 = t < a[:, i]
        t2 = t + carry
        c2 = t2 < t
        s[:, i] = t2
        carry = (c1 | c2).astype(np.uint64)
    ge = np.ones(a.shape[0], dtype=bool)          # s >= r, decided from the top limb down
    decided = np.zeros(a.shape[0], dtype=bool)
    for i in (3, 2, 1, 0):
        gt, lt = s[:, i] > r_limbs[i], s[:, i] < r_limbs[i]
        ge = np.where(~decided & lt, False, ge)
        decided |= gt | lt
    d = np.empty_like(s)
    borrow = np.zeros(a.shape[0], dtype=np.uint64)
    for i in range(4):
        t = s[:, i] - r_limbs[i]
        b1 = s[:, i] < r_limbs[i]
        t2 = t - borrow
        b2 = t < borrow
        d[:, i] = t2
        borrow = (b1 | b2).astype(np.uint64)
    return np.where(ge[:, None], d, s)


@pytest.mark.parametrize("log_m", [20, 22])
def test_msm_is_linear_in_the_scalars_at_full_size(ctx, log_m):
    """A size-independent property at BASELINE's full sizes (2^20: config 3, 2^22: config 5): commit(a) + commit(b) ==
    commit(a + b) -- over three different entry points, so that it also ties them together: a through typlonk_msm_g1 (scalars in
    host memory, copied chunk by chunk), b through typlonk_msm_g1_devptr, a + b (and a again) through one
    typlonk_msm_g1_batch_devptr call.  The sum of the two commitments is folded on the host (typlonk_g1_sum_host)."""
    import torch
    from typlonk_amd.capi import g1_sum_host

    m = (1 << log_m) + 1
    rng = np.random.default_rng(1000 + log_m)

    def uniform():
        x = rng.integers(0, 1 << 63, size=(m, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(m, 4), dtype=np.uint64)
        x[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)       # < 2^254 < r
        return x

    a, b = uniform(), uniform()
    # the helper against big integers on a sample (the edge of the reduction included: force one sum above r)
    a[0], b[0] = np.array(O.fr_to_mont_limbs(5), dtype=np.uint64), np.array(O.fr_to_mont_limbs(O.R - 3), dtype=np.uint64)
    ab = _fr_add_mod_limbs(a, b)
    val = lambda l: sum(int(l[i]) << (64 * i) for i in range(4))   # noqa: E731
    for i in (0, 1, 2, m // 2, m - 1):
        assert val(ab[i]) == (val(a[i]) + val(b[i])) % O.R
    sid = ctx.srs_generate(np.array(O.fr_to_mont_limbs(0xABCDE12345), dtype=np.uint64), m + 2)
    ctx.srs_precompute(sid, 20)
    dev = torch.device("cuda", 0)
    to_dev = lambda x: torch.from_numpy(np.ascontiguousarray(x).view(np.int64)).to(dev)   # noqa: E731
    da, db, dab = to_dev(a), to_dev(b), to_dev(ab)
    torch.cuda.synchronize()
    ca = ctx.msm(sid, a)
    cb = ctx.msm_devptr(sid, db.data_ptr(), m)
    cab, ca2 = ctx.msm_batch_devptr(sid, [dab.data_ptr(), da.data_ptr()], [m, m])
    assert int(ca[1]) == int(cb[1]) == int(cab[1]) == 0
    assert (np.asarray(ca2[0]) == np.asarray(ca[0])).all()
    sxy, sinf = g1_sum_host(np.stack([np.asarray(ca[0]), np.asarray(cb[0])]), np.zeros(2, dtype=np.uint8))
    assert sinf == 0 and (sxy == np.asarray(cab[0])).all()
    ctx.srs_free(sid)
