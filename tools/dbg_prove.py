import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import typlonk_amd
from bench import synthetic_scalars, fr_mont_limbs
dev = torch.device("cuda", 0)
ctx = typlonk_amd.Context(0); ctx.set_profiling(True)
log_n = 20; n = 1 << log_n
sid = ctx.srs_generate(fr_mont_limbs(2), n + 3)
polys = [synthetic_scalars(n, 0xB0B + i, dev) for i in range(3)]
torch.cuda.synchronize()
for i in range(6):
    t = time.perf_counter(); ctx.ntt_devptr(polys[i % 3].data_ptr(), log_n, inverse=(i >= 3)); torch.cuda.synchronize()
    print("ntt", i, (time.perf_counter() - t) * 1e3, ctx.profile(), flush=True)
for i, m in enumerate([n, n, n - 3, n - 1, n - 1]):
    t = time.perf_counter(); ctx.msm_devptr(sid, polys[i % 3].data_ptr(), m); torch.cuda.synchronize()
    print("msm", m, (time.perf_counter() - t) * 1e3, ctx.profile(), flush=True)
