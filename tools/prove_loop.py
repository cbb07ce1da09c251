"""A fixed number of real prove() calls on the squaring-chain circuit (for rocprofv3 kernel traces: where a proof's
GPU time goes).  REPS proofs after one warm-up; BATCHED=1 selects the batched-opening shape."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, typlonk_amd
from typlonk_amd.circuits import SquaringChain
from bench import fr_mont_limbs

log_n = int(os.environ.get("LOG_N", "20"))
n = 1 << log_n
ctx = typlonk_amd.Context(0)
sid = ctx.srs_generate(fr_mont_limbs(2), n + 3)
c = int(os.environ.get("TABLES", "20"))
if c:
    ctx.srs_precompute(sid, c)
chain = SquaringChain(ctx, log_n)
ch = [fr_mont_limbs(0x1234567 + k) for k in range(5)]
batched = os.environ.get("BATCHED", "0") == "1"
kw = {"challenge_v": (lambda e: ch[4])} if batched else {}
run = lambda: ctx.prove(sid, chain.circuit, chain.wire_evals, chain.pi_evals, chain.cosets,
                        lambda c_: (ch[0], ch[1]), lambda c_: (ch[2], ch[3]), **kw)
run()
torch.cuda.synchronize()
reps = int(os.environ.get("REPS", "5"))
t0 = time.perf_counter()
gap = float(os.environ.get("GAP_MS", "0")) * 1e-3   # idle time between proofs (separates them in a kernel trace)
for _ in range(reps):
    run()
    if gap:
        torch.cuda.synchronize()
        time.sleep(gap)
torch.cuda.synchronize()
print(f"prove log_n={log_n} batched={batched}: {(time.perf_counter() - t0) / reps * 1e3:.2f} ms per proof", flush=True)
