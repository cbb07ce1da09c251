"""Soak: many proofs / MSMs / coset NTTs in a row; device memory must stay flat and results identical."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, typlonk_amd
from typlonk_amd.circuits import SquaringChain
from bench import fr_mont_limbs, synthetic_scalars

def free_mb():
    return torch.cuda.mem_get_info()[0] / 2**20

ctx = typlonk_amd.Context(0)
ch = [fr_mont_limbs(0x1234567 + k) for k in range(5)]
for log_n, reps in ((16, 150), (20, 30)):
    n = 1 << log_n
    sid = ctx.srs_generate(fr_mont_limbs(2), n + 3)
    ctx.srs_precompute(sid, 20)
    chain = SquaringChain(ctx, log_n)
    ref, mem = None, []
    for i in range(reps):
        kw = {"challenge_v": (lambda e: ch[4])} if i % 2 else {}
        pr = ctx.prove(sid, chain.circuit, chain.wire_evals, chain.pi_evals, chain.cosets, lambda c: (ch[0], ch[1]), lambda c: (ch[2], ch[3]), **kw)
        key = (i % 2, bytes(np.asarray(pr["commit"][0][0])) + bytes(np.asarray(pr["witness"][0][0])))
        if ref is None: ref = {}
        assert ref.setdefault(key[0], key[1]) == key[1], "proof changed between repetitions"
        if i % 10 == 0:
            x = synthetic_scalars(1 << 14, i, torch.device("cuda", 0))
            ctx.ntt_devptr(x.data_ptr(), 14, coset=fr_mont_limbs(3 + i))     # a new coset shift every time: bounded cache
            ctx.msm_devptr(sid, chain.wire_evals[0].devptr, n - (i % 7))
            torch.cuda.synchronize(); mem.append(free_mb())
    print(f"log_n {log_n}: {reps} proofs ok; free device memory MB first/min/last: {mem[1]:.0f} / {min(mem[1:]):.0f} / {mem[-1]:.0f}")
    assert mem[1] - mem[-1] < 64, "device memory keeps shrinking"
    chain.free(); ctx.srs_free(sid)
ctx.close()
print("soak ok")
