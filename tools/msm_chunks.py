"""Single-MSM latency vs the number of pipeline chunks (TYPLONK_MSM_CHUNKS), 2^LOG_M terms with fixed-base tables."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, typlonk_amd
from bench import synthetic_scalars, fr_mont_limbs

log_m = int(os.environ.get("LOG_M", "20"))
m = 1 << log_m
dev = torch.device("cuda", 0)
sc = synthetic_scalars(m, 1, dev)
for chunks in [int(x) for x in os.environ.get("CHUNKS", "1,2,3,4,6,8").split(",")]:
    os.environ["TYPLONK_MSM_CHUNKS"] = str(chunks)
    ctx = typlonk_amd.Context(0)
    sid = ctx.srs_generate(fr_mont_limbs(2), m + 3)
    if int(os.environ.get("TABLES", "20")):
        ctx.srs_precompute(sid, int(os.environ.get("TABLES", "20")))
    for _ in range(3):
        ref = ctx.msm_devptr(sid, sc.data_ptr(), m)
    torch.cuda.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        out = ctx.msm_devptr(sid, sc.data_ptr(), m)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps * 1e3
    ctx.set_profiling(True)
    ctx.msm_devptr(sid, sc.data_ptr(), m)
    st = {}
    for n, ms in ctx.profile():
        st[n] = st.get(n, 0.0) + ms
    print(json.dumps({"log_m": log_m, "chunks": chunks, "ms": round(dt, 4), "stages": {k: round(v, 3) for k, v in st.items()},
                      "x0": int(out[0][0])}), flush=True)
    ctx.close()
