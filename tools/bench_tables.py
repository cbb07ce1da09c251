"""MSM time with fixed-base tables (typlonk_srs_precompute) vs the plain path, one process."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, typlonk_amd
from bench import synthetic_scalars, fr_mont_limbs

ctx = typlonk_amd.Context(0)
ctx.set_profiling(True)
dev = torch.device("cuda", 0)
for log_m in [int(x) for x in os.environ.get("SIZES", "17,20").split(",")]:
    m = 1 << log_m
    sc = synthetic_scalars(m, 1, dev)
    ref = None
    for c in [0] + [int(x) for x in os.environ.get("CS", "16,17,18,19,20").split(",")]:
        sid = ctx.srs_generate(fr_mont_limbs(2), m + 3)
        t0 = time.perf_counter()
        if c:
            ctx.srs_precompute(sid, c)
        tb = time.perf_counter() - t0
        for _ in range(3):
            out = ctx.msm_devptr(sid, sc.data_ptr(), m)
        if ref is None:
            ref = out
        assert (out[0] == ref[0]).all() and out[1] == ref[1]
        torch.cuda.synchronize()
        reps, st = 10, {}
        t = time.perf_counter()
        for _ in range(reps):
            ctx.msm_devptr(sid, sc.data_ptr(), m)
            for k, v in ctx.profile():
                st[k] = st.get(k, 0) + v / reps
        dt = (time.perf_counter() - t) / reps * 1e3
        print(json.dumps({"log_m": log_m, "tables_c": c, "ms": round(dt, 3), "build_s": round(tb, 2),
                          "stages": {k: round(v, 3) for k, v in st.items()}}), flush=True)
        ctx.srs_free(sid)
