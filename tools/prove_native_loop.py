"""A fixed number of NATIVE proofs (typlonk_prove: the library's own transcript between the rounds, no Python in between) on the
squaring-chain circuit, for rocprofv3 kernel traces.  HOST=1: typlonk_prove_host (columns in host memory)."""
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
ctx.srs_precompute(sid, 20)
chain = SquaringChain(ctx, log_n)
host = os.environ.get("HOST", "0") == "1"
cols = [b.download() for b in chain.wire_evals] if host else None
run = (lambda: ctx.prove_native_host(sid, chain.circuit, cols, None, chain.cosets)) if host else \
      (lambda: ctx.prove_native(sid, chain.circuit, chain.wire_evals, chain.pi_evals, chain.cosets))
run()
torch.cuda.synchronize()
reps = int(os.environ.get("REPS", "5"))
gap = float(os.environ.get("GAP_MS", "0")) * 1e-3
tot = 0.0
for _ in range(reps):
    t0 = time.perf_counter()
    run()
    tot += time.perf_counter() - t0
    if gap:
        torch.cuda.synchronize()
        time.sleep(gap)
print(f"prove_native log_n={log_n} host={host}: {tot / reps * 1e3:.2f} ms per proof", flush=True)
