"""GPU tests of the library's behaviour around the hot path: stream ordering of raw device pointers, the bounded
coset-table cache, set-up calls that must not break later MSMs.  Bit-exact comparisons with the CPU oracle."""
import numpy as np
import pytest

from helpers import O, fr_pack, fr_unpack, g1_unpack_one

pytestmark = pytest.mark.gpu


def _limbs(x):
    return np.array(O.fr_to_mont_limbs(x), dtype=np.uint64)


def test_devptr_calls_are_ordered_after_the_default_stream(ctx):
    """typlonk.h, "STREAM ORDERING": the context's own stream is ordered after the legacy default stream, where torch
    runs.  A long chain of torch kernels rewrites the vector and the library is called WITHOUT a synchronisation in
    between: it must see the final contents (NTT and MSM), exactly as after an explicit synchronize."""
    import torch

    dev = torch.device("cuda", 0)
    log_n, n = 16, 1 << 16
    sid = ctx.srs_generate(_limbs(0x5151), n)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    base = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, device=dev, generator=g)
    base[:, 3] &= 0x0FFFFFFFFFFFFFFF
    filler = torch.ones((4096, 4096), device=dev)
    torch.cuda.synchronize()
    for rep in range(3):
        y = base.clone()
        for k in range(20):
            filler = filler @ filler * 0 + 1            # keeps the default stream busy for a while
            y[:, 0] += 0x1234567 * (k + rep + 1)          # ... while the input is still being rewritten
            y[:, 3] &= 0x0FFFFFFFFFFFFFFF
        got_msm = ctx.msm_devptr(sid, y.data_ptr(), n)   # no synchronize: ordering comes from the stream semantics
        z = y.clone()
        ctx.ntt_devptr(z.data_ptr(), log_n)
        ctx.sync()
        torch.cuda.synchronize()
        want = y.cpu().numpy().view(np.uint64)
        ref_msm = ctx.msm(sid, want)                       # host path: upload of the finished vector
        assert (got_msm[0] == ref_msm[0]).all() and got_msm[1] == ref_msm[1]
        assert (z.cpu().numpy().view(np.uint64) == ctx.ntt(want, log_n)).all()
    ctx.srs_free(sid)


def test_set_stream_binds_a_torch_side_stream(ctx):
    """a producer on a NON-default stream: the context is bound to it (typlonk_set_stream) and reads in its order"""
    import torch

    dev = torch.device("cuda", 0)
    log_n, n = 14, 1 << 14
    side = torch.cuda.Stream(device=dev)
    v = O.random_frs(0xABCD, n)
    host = torch.from_numpy(fr_pack(v).view(np.int64))
    try:
        ctx.set_stream(side.cuda_stream)
        with torch.cuda.stream(side):
            d = host.to(dev, non_blocking=False)
            for _ in range(10):
                d = d.clone()
            ctx.ntt_devptr(d.data_ptr(), log_n)
        ctx.sync()
        side.synchronize()
        assert fr_unpack(d.cpu().numpy().view(np.uint64)) == O.ntt(v, log_n)
    finally:
        ctx.set_stream(None)


def test_coset_table_cache_is_bounded_and_stays_correct(ctx):
    """coset tables are keyed by the caller's shift: more distinct shifts than the cache holds (8 groups) are served
    correctly -- evicted groups are rebuilt -- forward and inverse, and the quotient's generator keeps working"""
    log_n, n = 10, 1 << 10
    v = O.random_frs(0xC0C0, n)
    packed = fr_pack(v)
    shifts = [7, 5, 11, 13, 17, 19, 23, 29, 31, 37, 41, 7, 5]
    for s in shifts:
        g = _limbs(s)
        assert fr_unpack(ctx.ntt(packed, log_n, coset=g)) == O.ntt(v, log_n, coset=s)
    for s in (43, 7):
        g = _limbs(s)
        assert fr_unpack(ctx.ntt(packed, log_n, inverse=True, coset=g)) == O.ntt(v, log_n, inverse=True, coset=s)


def test_precompute_never_breaks_a_valid_msm(ctx):
    """for len in (2^22, 2^23] with 20-bit windows a full-length MSM has no table-mode sort shape.  The
    set-up call is speed-only: such an MSM takes the plain path over the same SRS, shorter ones use the tables, and
    every result equals commit(p) == [p(s)]G (kzg/src/lib.rs:102-105)"""
    from oracle import coracle as CO

    n = (1 << 22) + 5
    secret = 3
    sid = ctx.srs_generate(_limbs(secret), n)
    ctx.srs_precompute(sid, 20)
    for m in (n, 1 << 22, (1 << 21) + 7, 1000):
        rng = np.random.default_rng(m)
        sc = rng.integers(0, 1 << 62, size=(m, 4), dtype=np.uint64)
        sc[:, 3] &= np.uint64(0x0FFFFFFFFFFFFFFF)
        out, oinf = ctx.msm(sid, sc)
        ps = CO.poly_eval(sc, _limbs(secret))
        exp_xy, exp_inf = CO.g1_mul_generator(ps)
        assert (out == exp_xy).all() and oinf == exp_inf, m
    ctx.srs_free(sid)


def test_msm_results_identical_with_and_without_profiling(ctx):
    n = 1 << 12
    sid = ctx.srs_generate(_limbs(0x77), n)
    rng = np.random.default_rng(5)
    sc = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    sc[:, 3] &= np.uint64(0x0FFFFFFFFFFFFFFF)
    a = ctx.msm(sid, sc)
    ctx.set_profiling(True)
    b = ctx.msm(sid, sc)
    names = [nm for nm, _ in ctx.profile()]
    ctx.set_profiling(False)
    assert (a[0] == b[0]).all() and a[1] == b[1]
    assert any(nm.startswith("msm_accum") for nm in names)
    assert g1_unpack_one(a[0], a[1]) is not None
    ctx.srs_free(sid)


def test_heavy_bucket_shapes_stay_within_reach_of_the_uniform_case(built):
    """Performance guard, loose on purpose (profiles/r03_heavy_tasks_ab.txt): scalar sets that pile entries on a few
    buckets -- 64 distinct values, or tables whose top window is 2 bits wide -- used to cost 55x / 25x a uniform MSM of
    the same size (one thread per 512-entry task, all folds in one workgroup); with a wavefront per task they cost
    1.2-2x.  Fails only if the cliff comes back (> 12x)."""
    import time

    import torch

    import typlonk_amd
    from bench import fr_mont_limbs, synthetic_scalars

    ctx = typlonk_amd.Context(0)
    try:
        m = 1 << 16
        dev = torch.device("cuda", 0)
        uni = synthetic_scalars(m, 5, dev)
        rep = uni.clone()
        rep[:] = uni[torch.arange(m, device=dev) >> 10 << 10]      # 64 distinct scalars, 1024 copies each

        def ms(sid, sc):
            for _ in range(3):
                ctx.msm_devptr(sid, sc.data_ptr(), m)
            t0 = time.perf_counter()
            for _ in range(10):
                ctx.msm_devptr(sid, sc.data_ptr(), m)
            return (time.perf_counter() - t0) / 10 * 1e3

        auto = ctx.srs_generate(fr_mont_limbs(2), m + 3)
        ctx.srs_precompute(auto, 0)
        thin = ctx.srs_generate(fr_mont_limbs(2), m + 3)
        ctx.srs_precompute(thin, 18)
        base = ms(auto, uni)
        assert ms(auto, rep) < 12 * base
        assert ms(thin, uni) < 12 * base
        # and the points are the same whichever path sums them
        a = ctx.msm_devptr(auto, rep.data_ptr(), m)
        b = ctx.msm_devptr(thin, rep.data_ptr(), m)
        assert (a[0] == b[0]).all() and a[1] == b[1]
    finally:
        ctx.close()


def test_auto_precompute_leaves_a_short_srs_alone(ctx):
    """typlonk_srs_precompute(window_bits = 0) on an SRS shorter than TYPLONK_TABLES_AUTO_MIN_LEN points builds nothing
    (2^16 buckets for a hundred terms would be a cliff) and returns OK; MSMs keep the plain path and the oracle's result,
    and an explicit window can still be asked for afterwards (it would be refused had tables been built)."""
    from typlonk_amd.capi import TyplonkError

    n = 100
    srs = O.srs_from_secret_fast(3, n)
    from helpers import g1_pack

    xy, inf = g1_pack(srs)
    sid = ctx.srs_load(xy, inf)
    ctx.srs_precompute(sid, 0)
    coeffs = O.random_frs(0xA070, n)
    out, oinf = ctx.msm(sid, fr_pack(coeffs))
    assert g1_unpack_one(out, oinf) == O.g1_mul(O.G1, O.poly_eval(coeffs, 3))
    ctx.srs_precompute(sid, 0)              # still nothing built: not "tables already built"
    big = ctx.srs_generate(_limbs(3), 1 << 14)
    ctx.srs_precompute(big, 0)              # at the threshold the tables are built ...
    with pytest.raises(TyplonkError):
        ctx.srs_precompute(big, 0)          # ... and a second build is refused
    ctx.srs_free(sid)
    ctx.srs_free(big)


def test_automatic_window_and_lane_policy_keep_the_oracles_points(built, monkeypatch):
    """typlonk_srs_precompute(0) picks the window by SRS length (15 below 2^16 points, 17 below 2^19, else 20: typlonk.h)
    and a batch picks its lanes by size (free-running below 2^17 terms, four of them below 2^17 points; chained above:
    host.hpp).  Either side of each switch the points are the CPU oracle's, and a batch equals its single calls, equals the
    same batch under the other policy."""
    import torch
    import typlonk_amd
    from oracle import coracle as CO

    dev = torch.device("cuda", 0)
    results = {}
    for policy in ("default", "chained"):
        rng = np.random.default_rng(0x515E)   # the same vectors under both policies
        if policy == "chained":
            monkeypatch.setenv("TYPLONK_MSM_CHAIN", "1")
            monkeypatch.setenv("TYPLONK_MSM_INFLIGHT", "3")
        c2 = typlonk_amd.Context(0)
        try:
            for length in ((1 << 14) + 1, (1 << 16) - 1, 1 << 16, (1 << 17) - 1, (1 << 17) + 3):
                sid = c2.srs_generate(_limbs(5), length)
                c2.srs_precompute(sid, 0)
                vecs = []
                for _ in range(5):
                    x = rng.integers(0, 1 << 63, size=(length, 4), dtype=np.uint64)
                    x[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
                    vecs.append(x)
                dv = [torch.from_numpy(v.view(np.int64)).to(dev) for v in vecs]
                torch.cuda.synchronize()
                ms = [length, length - 1, length // 2, length - 3, length]
                batch = c2.msm_batch_devptr(sid, [t.data_ptr() for t in dv], ms)
                singles = [c2.msm_devptr(sid, t.data_ptr(), m) for t, m in zip(dv, ms)]
                for b, s1 in zip(batch, singles):
                    assert b[1] == s1[1] and (np.asarray(b[0]) == np.asarray(s1[0])).all()
                results[(policy, length)] = [(np.asarray(b[0]).copy(), b[1]) for b in batch]
                if policy == "default":
                    xy, inf = c2.srs_download(sid)
                    want = CO.msm_pippenger(vecs[0], xy, inf)[:2]
                    assert batch[0][1] == want[1] and (np.asarray(batch[0][0]) == want[0]).all()
                c2.srs_free(sid)
        finally:
            c2.close()
    for (policy, length), pts in results.items():
        if policy == "default":
            for a, b in zip(pts, results[("chained", length)]):
                assert a[1] == b[1] and (a[0] == b[0]).all()
