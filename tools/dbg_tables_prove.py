import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, typlonk_amd
from typlonk_amd.circuits import SquaringChain, fr_mont_limbs
log_n = 20; n = 1 << log_n
ctx = typlonk_amd.Context(0)
chain = SquaringChain(ctx, log_n)
ch = [fr_mont_limbs(0x1234567 + k) for k in range(4)]
for tables in (0, 20):
    sid = ctx.srs_generate(fr_mont_limbs(2), n + 3)
    if tables: ctx.srs_precompute(sid, tables)
    f = lambda: ctx.prove(sid, chain.circuit, chain.wire_evals, chain.pi_evals, chain.cosets, lambda c: (ch[0], ch[1]), lambda c: (ch[2], ch[3]))
    f(); torch.cuda.synchronize()
    t = time.perf_counter(); f(); torch.cuda.synchronize(); print("tables", tables, "prove ms", (time.perf_counter() - t) * 1e3)
    # individual commitments of the wire polynomials with stage profile
    ctx.set_profiling(True)
    for i in range(3):
        b = ctx.alloc(n)
        b.upload(chain.wire_evals[i].download())
        ctx.ntt_dev(b, log_n, inverse=True)
        ctx.msm_dev(sid, b, 0, n)
        t = time.perf_counter(); ctx.msm_dev(sid, b, 0, n); dt = (time.perf_counter() - t) * 1e3
        print("  wire", i, round(dt, 2), [(k, round(v, 2)) for k, v in ctx.profile()])
        b.free()
    ctx.set_profiling(False)
    ctx.srs_free(sid)
