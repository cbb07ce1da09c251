import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, typlonk_amd
from bench import synthetic_scalars, fr_mont_limbs
m = 1 << 20
ctx = typlonk_amd.Context(0)
sc = synthetic_scalars(m, 1, torch.device("cuda", 0))
sid = ctx.srs_generate(fr_mont_limbs(2), m + 3)
ctx.srs_precompute(sid, 20)
for prof in (False, True, False, True):
    ctx.set_profiling(prof)
    for _ in range(3): ctx.msm_devptr(sid, sc.data_ptr(), m)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        ctx.msm_devptr(sid, sc.data_ptr(), m)
        if prof: ctx.profile()
    torch.cuda.synchronize()
    print("profiling", prof, round((time.perf_counter() - t0) / 20 * 1e3, 4), "ms per MSM")
