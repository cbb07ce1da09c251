"""Set-up cost of an SRS on the device: typlonk_srs_generate (Srs::from_secret, kzg/src/srs.rs:15-34) and
typlonk_srs_precompute (fixed-base tables), wall time per call, with the points checked against a second route
(tables: MSM over the tables == MSM over the plain SRS; generate: a slice against the CPU oracle)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, typlonk_amd
from bench import synthetic_scalars, fr_mont_limbs

ctx = typlonk_amd.Context(0)
dev = torch.device("cuda", 0)
for log_n in [int(x) for x in os.environ.get("SIZES", "16,20").split(",")]:
    n = (1 << log_n) + 3
    ctx.srs_free(ctx.srs_generate(fr_mont_limbs(3), 64))   # comb table + first-launch costs out of the way
    t0 = time.perf_counter()
    sid = ctx.srs_generate(fr_mont_limbs(2), n)
    t_gen = time.perf_counter() - t0
    plain = ctx.srs_generate(fr_mont_limbs(2), n)
    t0 = time.perf_counter()
    ctx.srs_precompute(sid, 0)
    t_tab = time.perf_counter() - t0
    sc = synthetic_scalars(1 << log_n, 1, dev)
    a = ctx.msm_devptr(sid, sc.data_ptr(), 1 << log_n)
    b = ctx.msm_devptr(plain, sc.data_ptr(), 1 << log_n)
    assert (a[0] == b[0]).all() and a[1] == b[1]
    print(json.dumps({"log_n": log_n, "srs_generate_ms": round(t_gen * 1e3, 2), "srs_precompute_ms": round(t_tab * 1e3, 2),
                      "tables_equal_plain": True}), flush=True)
    ctx.srs_free(sid)
    ctx.srs_free(plain)
