import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, typlonk_amd
from bench import synthetic_scalars, fr_mont_limbs
m = 1 << 20
ctx = typlonk_amd.Context(0)
sid = ctx.srs_generate(fr_mont_limbs(2), m + 3)
ctx.srs_precompute(sid, 20)
sc = synthetic_scalars(m, 1, torch.device("cuda", 0))
ctx.set_profiling(True)
for i in range(14):
    t0 = time.perf_counter()
    ctx.msm_devptr(sid, sc.data_ptr(), m)
    dt = (time.perf_counter() - t0) * 1e3
    st = {}
    for n, ms in ctx.profile(): st[n] = round(st.get(n, 0) + ms, 3)
    print(i, round(dt, 2), st)
