"""Per-step wall time and accumulation-kernel time of the first MSMs of a fresh process (how long the GPU takes to
reach its steady clocks / state).  PRELOAD_MS=<ms> first keeps the GPU busy with SRS-table builds for that long."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, typlonk_amd
from bench import synthetic_scalars, fr_mont_limbs

m = 1 << 20
ctx = typlonk_amd.Context(0)
ctx.set_profiling(True)
sid = ctx.srs_generate(fr_mont_limbs(2), m + 3)
ctx.srs_precompute(sid, 20)
sc = synthetic_scalars(m, 1, torch.device("cuda", 0))
pre = float(os.environ.get("PRELOAD_MS", "0"))
if pre:
    t0 = time.perf_counter()
    s2 = ctx.srs_generate(fr_mont_limbs(3), m + 3)
    while (time.perf_counter() - t0) * 1e3 < pre:
        ctx.srs_precompute(s2, 20)
    ctx.srs_free(s2)
out = []
for i in range(40):
    t0 = time.perf_counter()
    ctx.msm_devptr(sid, sc.data_ptr(), m)
    dt = (time.perf_counter() - t0) * 1e3
    acc = sum(v for k, v in ctx.profile() if k == "msm_accum")
    out.append((dt, acc))
print("RAMP preload_ms=%s " % pre + " ".join(f"{a:.2f}/{b:.2f}" for a, b in out), flush=True)
