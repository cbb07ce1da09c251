"""NTT timings per size / direction / coset with the per-pass HIP-event times."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, typlonk_amd
from bench import synthetic_scalars, fr_mont_limbs

ctx = typlonk_amd.Context(0)
ctx.set_profiling(True)
dev = torch.device("cuda", 0)
g = fr_mont_limbs(7)
for log_n in [int(x) for x in os.environ.get("SIZES", "16,20,22,24").split(",")]:
    n = 1 << log_n
    x = synthetic_scalars(n, 3, dev)
    for inverse, coset in ((False, None), (True, None), (False, g), (True, g)):
        for _ in range(3):
            ctx.ntt_devptr(x.data_ptr(), log_n, inverse, coset)
        torch.cuda.synchronize()
        reps, st = 10, {}
        t = time.perf_counter()
        for _ in range(reps):
            ctx.ntt_devptr(x.data_ptr(), log_n, inverse, coset)
            for k, v in ctx.profile():
                st[k] = st.get(k, 0) + v / reps
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / reps * 1e3
        ks = sum(st.values())
        print(json.dumps({"log_n": log_n, "inverse": inverse, "coset": coset is not None, "ms": round(dt, 4),
                          "kernel_ms": round(ks, 4), "GBps": round(64 * n / ks / 1e6, 1),
                          "passes": {k: round(v, 4) for k, v in st.items()}}), flush=True)
