"""MSM time vs window bits c at several sizes (one GPU) -- tunes msm_shape() for sharded runs."""
import os, sys, time, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np, torch, typlonk_amd
    from bench import synthetic_scalars, fr_mont_limbs
    log_m = int(sys.argv[2])
    m = 1 << log_m
    ctx = typlonk_amd.Context(0)
    ctx.set_profiling(True)
    sid = ctx.srs_generate(fr_mont_limbs(2), m + 3)
    sc = synthetic_scalars(m, 1, torch.device("cuda", 0))
    for _ in range(3):
        ctx.msm_devptr(sid, sc.data_ptr(), m)
    torch.cuda.synchronize()
    t = time.perf_counter()
    reps = 10
    st = {}
    for _ in range(reps):
        ctx.msm_devptr(sid, sc.data_ptr(), m)
        for k, v in ctx.profile():
            st[k] = st.get(k, 0) + v / reps
    dt = (time.perf_counter() - t) / reps * 1e3
    print(json.dumps({"log_m": log_m, "c": os.environ.get("TYPLONK_MSM_C", "auto"), "ms": round(dt, 3),
                      "stages": {k: round(v, 3) for k, v in st.items()}}))
else:
    sizes = [int(x) for x in os.environ.get("SWEEP_SIZES", "17,18,19,20").split(",")]
    cs = os.environ.get("SWEEP_CS", "auto,11,12,13,14,15,16").split(",")
    for log_m in sizes:
        for c in cs:
            env = dict(os.environ)
            if c != "auto":
                env["TYPLONK_MSM_C"] = c
            r = subprocess.run([sys.executable, __file__, "child", str(log_m)], env=env, capture_output=True, text=True)
            print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
