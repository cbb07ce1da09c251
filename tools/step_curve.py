"""Per-step wall time of the first 60 MSMs after set-up (is there a warm-up curve?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, typlonk_amd
from bench import synthetic_scalars, fr_mont_limbs
m = 1 << 20
ctx = typlonk_amd.Context(0)
sid = ctx.srs_generate(fr_mont_limbs(2), m + 3)
ctx.srs_precompute(sid, 20)
sc = synthetic_scalars(m, 1, torch.device("cuda", 0))
if os.environ.get("SLEEP"):
    time.sleep(float(os.environ["SLEEP"]))
if os.environ.get("PREHEAT"):
    # keep the chip busy right up to the first MSM (is the curve a clock ramp?)
    x = synthetic_scalars(1 << 22, 2, torch.device("cuda", 0))
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < float(os.environ["PREHEAT"]):
        ctx.ntt_devptr(x.data_ptr(), 22)
    ctx.sync()
ts = []
for i in range(60):
    t0 = time.perf_counter()
    ctx.msm_devptr(sid, sc.data_ptr(), m)
    ts.append((time.perf_counter() - t0) * 1e3)
print(" ".join(f"{t:.2f}" for t in ts))
