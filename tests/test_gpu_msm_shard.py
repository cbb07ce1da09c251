"""GPU parity of the shard-sized MSM shapes (an 8-way index shard of a 2^20-term commitment, BASELINE config 4;
KzgScheme::evaluate_in_s on a slice, kzg/src/lib.rs:41-54): several lanes per bucket in the accumulation
(msm_accum_ml_kernel), centred scalars with 15 windows of 17 bits, the two-launch row/column reduction.  Every variant
must give the bit-identical canonical affine point: checked against the CPU bucket-method oracle, the reference's own
commit(p) == [p(s)]G identity (kzg/src/lib.rs:102-105) and each other."""
import numpy as np
import pytest

from helpers import O

pytestmark = pytest.mark.gpu

R = O.R


def _limbs(v):
    return np.array(O.fr_to_mont_limbs(v % R), dtype=np.uint64)


def _uniform(rng, m):
    sc = rng.integers(0, 1 << 63, size=(m, 4), dtype=np.uint64) * 2 + rng.integers(0, 2, size=(m, 4), dtype=np.uint64)
    sc[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)   # < 2^254 < r: any limb pattern is a valid Montgomery residue
    return sc


def _edge_scalars(rng, m):
    """the values that matter for centred digits and signed windows: 0, 1, r - 1, (r - 1)/2, (r + 1)/2, 2^254 - 1 ...,
    digit patterns at the window boundaries of c = 15, 17 and 20, mixed with uniform ones"""
    sc = _uniform(rng, m)
    half = (R - 1) // 2
    special = [0, 1, R - 1, half, half + 1, half - 1, R - 2, 2, (1 << 254) - 1, 1 << 253, (1 << 238) - 1, 1 << 238,
               (1 << 16) - 1, 1 << 16, (1 << 16) + 1, (1 << 17) - 1, (1 << 255) % R, R - (1 << 16), R - (1 << 238),
               sum(1 << (17 * j + 16) for j in range(15)) % R, sum((1 << 17) - 1 << (17 * j) for j in range(0, 15, 2)) % R]
    idx = rng.permutation(m)[: min(m, 6 * len(special))]
    for k, i in enumerate(idx):
        sc[i] = _limbs(special[k % len(special)])
    return sc


@pytest.mark.parametrize("lanes", [1, 2, 4, 8, 16])
@pytest.mark.parametrize("reduce_mode", ["", "rc4"])
def test_lanes_per_bucket_and_reductions_give_identical_points(built, lanes, reduce_mode, monkeypatch):
    """TYPLONK_MSM_LANES forces the lanes-per-bucket count of the accumulation (default: chosen from the bucket load),
    TYPLONK_MSM_REDUCE=rc4 the four-launch reduction: every combination equals the CPU bucket method, plain and with
    tables of every window, on uniform, edge-value and heavy-bucket scalar sets, full and ragged lengths, on a shard"""
    import typlonk_amd
    from oracle import coracle as CO
    from typlonk_amd.capi import g1_sum_host

    monkeypatch.setenv("TYPLONK_MSM_LANES", str(lanes))
    if reduce_mode:
        monkeypatch.setenv("TYPLONK_MSM_REDUCE", reduce_mode)
    c2 = typlonk_amd.Context(0)
    try:
        length = (1 << 13) + 3
        s_limbs = _limbs(0x0123456789ABCDEF0123456789ABCDEF)
        plain = c2.srs_generate(s_limbs, length)
        xy, inf = c2.srs_download(plain)
        handles = {"plain": plain}
        for c in (14, 15, 17, 20):
            h = c2.srs_generate(s_limbs, length)
            c2.srs_precompute(h, c)
            handles[f"tables{c}"] = h
        cut = 3001
        shard = c2.srs_generate(s_limbs, length - cut, start=cut)
        c2.srs_set_shard(shard, cut, length)
        c2.srs_precompute(shard, 17)
        rng = np.random.default_rng(77 + lanes)
        for m in (length, length - 3, length // 2 + 1, 2500):
            for kind in ("uniform", "edge", "ones", "rm1"):
                if kind == "uniform":
                    sc = _uniform(rng, m)
                elif kind == "edge":
                    sc = _edge_scalars(rng, m)
                else:
                    sc = np.tile(_limbs(1 if kind == "ones" else R - 1), (m, 1))
                exp, einf, _, _ = CO.msm_pippenger(sc, xy, inf, c=11)
                for name, h in handles.items():
                    got, ginf = c2.msm(h, sc)
                    assert (got == exp).all() and ginf == einf, (lanes, reduce_mode, m, kind, name)
                part, pinf = c2.msm(shard, sc)
                head, hinf = c2.msm(plain, sc[:cut])
                fxy, finf = g1_sum_host(np.stack([part, head]), np.array([pinf, hinf], dtype=np.uint8))
                assert (fxy == exp).all() and finf == einf, (lanes, reduce_mode, m, kind, "shard")
    finally:
        c2.close()


@pytest.mark.parametrize("c", [15, 17])
def test_centred_tables_equal_the_reference_path_on_edge_scalars(ctx, c):
    """c = 17 / 15 tables use centred scalars (k > (r-1)/2 -> r - k with the signs flipped; 15 / 17 windows): the
    reference-faithful per-term MSM of the oracle on a slice, every edge value included"""
    from oracle import coracle as CO

    n = 2048
    s_limbs = _limbs(7)
    sid = ctx.srs_generate(s_limbs, n)
    xy, inf = ctx.srs_download(sid)
    ctx.srs_precompute(sid, c)
    rng = np.random.default_rng(c)
    sc = _edge_scalars(rng, n)
    got, ginf = ctx.msm(sid, sc)
    ref, rinf = CO.msm_reference(sc, xy, inf)
    assert (got == ref).all() and ginf == rinf
    ctx.srs_free(sid)


@pytest.mark.parametrize("world,log_n", [(8, 20), (4, 20), (8, 22)])
def test_index_shards_at_config_sizes_fold_to_the_commit_identity(ctx, world, log_n):
    """BASELINE config 4 / 5 shapes on one GPU: the 2^log_n-term commitment cut into `world` index shards, each with the
    auto-chosen tables (typlonk_srs_precompute(0): c = 17 centred below 2^19 points), multi-lane accumulation; the folded
    partial sums equal [p(s)]G (kzg/src/lib.rs:102-105) and the unsharded table-mode MSM"""
    from oracle import coracle as CO
    from typlonk_amd.capi import g1_sum_host
    from typlonk_amd.dist import shard_bounds

    n = 1 << log_n
    total = n + 3
    s_limbs = _limbs(2)
    rng = np.random.default_rng(log_n * 10 + world)
    sc = _edge_scalars(rng, n)
    buf = ctx.alloc(n)
    buf.upload(sc)
    exp, einf = CO.g1_mul_generator(CO.poly_eval(sc, s_limbs))
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(total, world, r)
        sid = ctx.srs_generate(s_limbs, hi - lo, start=lo)
        ctx.srs_set_shard(sid, lo, total)
        ctx.srs_precompute(sid, 0)
        parts.append(ctx.msm_devptr(sid, buf.devptr, n))
        # ragged prover lengths on the last shards: n - 1 and n - 3 (proof.rs opening / t_hi lengths)
        if r == world - 1:
            tail = [ctx.msm_devptr(sid, buf.devptr, n - d) for d in (1, 3)]
        ctx.srs_free(sid)
    fxy, finf = g1_sum_host(np.stack([p[0] for p in parts]), np.array([p[1] for p in parts], dtype=np.uint8))
    assert (fxy == exp).all() and finf == einf
    for d, (txy, tinf) in zip((1, 3), tail):
        e2, e2inf = CO.g1_mul_generator(CO.poly_eval(sc[: n - d], s_limbs))
        pts = parts[:-1] + [(txy, tinf)]
        gxy, ginf = g1_sum_host(np.stack([p[0] for p in pts]), np.array([p[1] for p in pts], dtype=np.uint8))
        assert (gxy == e2).all() and ginf == e2inf, d
    buf.free()
