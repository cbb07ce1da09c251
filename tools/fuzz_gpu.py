"""Time-boxed differential fuzz of the MSM and NTT entry points against the C restatement (oracle/coracle.py).

Not part of the test suite (it runs for FUZZ_SECONDS, default 600): random SRS lengths, table choices, term counts and
scalar distributions chosen to hit the corners of the bucket method (zeros, one value everywhere, a handful of distinct
values, r - 1, small scalars, single-window scalars), batches against single calls, index shards against the whole,
and NTTs of random size / direction / coset.  Prints one line per failure and a summary; exit code 1 on any mismatch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, typlonk_amd
from oracle import coracle as CO
from oracle import bls12_381 as O

SECONDS = float(os.environ.get("FUZZ_SECONDS", "600"))
SEED = int(os.environ.get("FUZZ_SEED", "20261002"))
MAX_LOG = int(os.environ.get("FUZZ_MAX_LOG", "17"))
rng = np.random.default_rng(SEED)
dev = torch.device("cuda", 0)
R = O.R
SORT_FORMS = [{}, {}, {"TYPLONK_MSM_SCATTER": "direct"}, {"TYPLONK_MSM_L1_THREADS": "256"}, {"TYPLONK_MSM_L1_THREADS": "512"},
              {"TYPLONK_MSM_SORT_PRIO": "0"}, {"TYPLONK_NTT_BIG": "1"}, {"TYPLONK_NTT_BIG": "2"},
              {"TYPLONK_MSM_FIRST_PCT": "30"}, {"TYPLONK_MSM_REDUCE": "rc2", "TYPLONK_MSM_RC2_LOGW": "11"}]
SORT_KEYS = sorted({k for f in SORT_FORMS for k in f})


def new_context():
    """a context under one of the forms the library offers for the bucket sort / the 2^20 NTT plan (read at creation)"""
    form = SORT_FORMS[int(rng.integers(0, len(SORT_FORMS)))]
    for k in SORT_KEYS:
        os.environ.pop(k, None)
    os.environ.update(form)
    return typlonk_amd.Context(0), form


ctx, form = new_context()


def mont(v):
    return np.array(O.fr_to_mont_limbs(int(v) % R), dtype=np.uint64)


def uniform(m):
    x = rng.integers(0, 1 << 63, size=(m, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(m, 4), dtype=np.uint64)
    x[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
    return x


def scalars(m):
    """(name, (m, 4) Montgomery words)"""
    kind = rng.integers(0, 9)
    if kind == 0:
        return "uniform", uniform(m)
    if kind == 1:
        x = uniform(m)
        x[rng.random(m) < 0.9] = 0
        return "sparse", x
    if kind == 2:
        vals = np.stack([mont(int(rng.integers(1, 1 << 62))) for _ in range(int(rng.integers(1, 5)))])
        return "few distinct", vals[rng.integers(0, len(vals), size=m)]
    if kind == 3:
        vals = np.stack([mont(R - 1), mont(1), mont(0), mont(R - 2), mont(2)])
        return "0 / 1 / r-1", vals[rng.integers(0, len(vals), size=m)]
    if kind == 4:
        return "one value", np.tile(uniform(1), (m, 1))
    if kind == 5:   # small standard-form values (Montgomery words of them)
        small = rng.integers(0, 1 << 16, size=m)
        table = {}
        return "small", np.stack([table.setdefault(int(v), mont(int(v))) for v in small])
    if kind == 6:   # one window digit only: v = d << (20 * w)
        w = int(rng.integers(0, 13))
        table = {}
        ds = rng.integers(0, 1 << 8, size=m)
        return f"digit in window {w}", np.stack([table.setdefault(int(d), mont((int(d) << (20 * w)) % R)) for d in ds])
    if kind == 7:   # all digits at the half-window boundary (signed-digit carries everywhere)
        v = sum(1 << (20 * w + 19) for w in range(12))
        vals = np.stack([mont(v), mont(v - 1), mont(v + 1), mont(R - v)])
        return "half-window digits", vals[rng.integers(0, 4, size=m)]
    x = uniform(m)
    x[: m // 2] = x[0]
    return "half repeated", x


def to_dev(x):
    t = torch.from_numpy(np.ascontiguousarray(x).view(np.int64)).to(dev)
    torch.cuda.synchronize()
    return t


def same(a, b):
    return a[1] == b[1] and (a[1] == 1 or (np.asarray(a[0]) == np.asarray(b[0])).all())


stats = {"msm": 0, "batch": 0, "shard": 0, "ntt": 0, "ntt_batch": 0, "contexts": 1, "fail": 0}
t_end = time.time() + SECONDS
rounds = 0
while time.time() < t_end:
    rounds += 1
    if rounds % 8 == 0:
        ctx.close()
        ctx, form = new_context()
        stats["contexts"] += 1
    # ---- one SRS, several MSMs over it
    L = int(rng.integers(1, 1 << int(rng.integers(1, MAX_LOG + 1)))) + 1
    secret = mont(int(rng.integers(2, 1 << 62)))
    sid = ctx.srs_generate(secret, L)
    tables = int(rng.choice([-1, 0, 0, 16, 20]))
    if tables >= 0 and L >= 64:
        try:
            ctx.srs_precompute(sid, tables)
        except typlonk_amd.TyplonkError:
            tables = -1   # window not offered for this length
    xy, inf = ctx.srs_download(sid)
    for _ in range(4):
        m = int(rng.integers(0, L + 1)) if rng.random() < 0.8 else L
        name, s = scalars(max(m, 1))
        s = s[:m]
        got = ctx.msm(sid, s, m) if m else ctx.msm(sid, np.zeros((1, 4), dtype=np.uint64), 0)
        want = CO.msm_pippenger(s, xy[: max(m, 1)], inf[: max(m, 1)])[:2] if m else (None, 1)
        stats["msm"] += 1
        if not same(got, want):
            stats["fail"] += 1
            print(f"FAIL msm L={L} m={m} tables={tables} scalars={name} form={form} seed={SEED}", flush=True)
    # ---- a batch against single calls
    if L >= 8:
        k = int(rng.integers(1, 10))
        vecs = [to_dev(scalars(L)[1]) for _ in range(k)]
        ms = [int(rng.integers(1, L + 1)) for _ in range(k)]
        singles = [ctx.msm_devptr(sid, v.data_ptr(), m) for v, m in zip(vecs, ms)]
        batch = ctx.msm_batch_devptr(sid, [v.data_ptr() for v in vecs], ms)
        stats["batch"] += 1
        if not all(same(a, b) for a, b in zip(singles, batch)):
            stats["fail"] += 1
            print(f"FAIL batch L={L} k={k} ms={ms} tables={tables} form={form} seed={SEED}", flush=True)
    ctx.srs_free(sid)
    # ---- index shards of one SRS against the whole
    if L >= 16:
        world = int(rng.choice([2, 3, 8]))
        name, s = scalars(L)
        m = int(rng.integers(1, L + 1))
        full_sid = ctx.srs_generate(secret, L)
        whole = ctx.msm(full_sid, s, m)
        ctx.srs_free(full_sid)
        parts = []
        shard_tables = int(rng.choice([-1, 0, 0, 15, 17]))
        sd = to_dev(s)
        for r in range(world):
            lo, hi = L * r // world, L * (r + 1) // world
            if hi == lo:
                continue
            ps = ctx.srs_generate(secret, hi - lo, start=lo)
            ctx.srs_set_shard(ps, lo, L)
            if shard_tables >= 0 and hi - lo >= 64:
                try:
                    ctx.srs_precompute(ps, shard_tables)   # 0 = the window the library picks for this shard's length
                except typlonk_amd.TyplonkError:
                    pass
            parts.append(ctx.msm_devptr(ps, sd.data_ptr(), m))
            ctx.srs_free(ps)
        acc = O.INF
        for p_xy, p_inf in parts:   # folded with the oracle's affine arithmetic, not the product's host fold
            acc = O.g1_add(acc, O.g1_from_limbs([int(v) for v in p_xy], p_inf))
        stats["shard"] += 1
        if acc != O.g1_from_limbs([int(v) for v in whole[0]], whole[1]):
            stats["fail"] += 1
            print(f"FAIL shard L={L} m={m} world={world} tables={shard_tables} scalars={name} seed={SEED}", flush=True)
    # ---- NTTs
    for _ in range(3):
        log_n = int(rng.integers(1, MAX_LOG + 2))
        inverse = bool(rng.integers(0, 2))
        coset = mont(int(rng.integers(2, 1 << 62))) if rng.random() < 0.5 else None
        _, x = scalars(1 << log_n)
        got = ctx.ntt(x, log_n, inverse=inverse, coset=coset)
        want = CO.ntt(x, log_n, inverse=inverse, coset=coset, threads=8)
        stats["ntt"] += 1
        if not (got == want).all():
            stats["fail"] += 1
            print(f"FAIL ntt log_n={log_n} inverse={inverse} coset={coset is not None} form={form} seed={SEED}", flush=True)
    # ---- a group of transforms in one call against single calls
    log_n = int(rng.integers(1, MAX_LOG + 1))
    k = int(rng.integers(1, 12))
    inverse = bool(rng.integers(0, 2))
    coset = mont(int(rng.integers(2, 1 << 62))) if rng.random() < 0.5 else None
    vecs = [to_dev(scalars(1 << log_n)[1]) for _ in range(k)]
    singles = [v.clone() for v in vecs]
    for v in singles:
        ctx.ntt_devptr(v.data_ptr(), log_n, inverse=inverse, coset=coset)
    ctx.ntt_batch_devptr([v.data_ptr() for v in vecs], log_n, inverse=inverse, coset=coset)
    torch.cuda.synchronize()
    stats["ntt_batch"] += 1
    if not all(torch.equal(a, b) for a, b in zip(vecs, singles)):
        stats["fail"] += 1
        print(f"FAIL ntt_batch log_n={log_n} k={k} inverse={inverse} coset={coset is not None} form={form} seed={SEED}", flush=True)
ctx.close()
print(f"fuzz seed {SEED}, {SECONDS:.0f} s: {stats}")
sys.exit(1 if stats["fail"] else 0)
