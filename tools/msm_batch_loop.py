"""REPS batches of BATCH table-mode MSMs in one typlonk_msm_g1_batch_devptr call (a prover round's commitments), with a
pause between batches so that they separate in a rocprofv3 kernel trace (tools/trace_timeline.py reads the last one)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, typlonk_amd
from bench import synthetic_scalars, fr_mont_limbs

log_m = int(os.environ.get("LOG_M", "20"))
m = 1 << log_m
nb = int(os.environ.get("BATCH", "9"))
ctx = typlonk_amd.Context(0)
sc = synthetic_scalars(m, 1, torch.device("cuda", 0))
sid = ctx.srs_generate(fr_mont_limbs(2), m + 3)
c = int(os.environ.get("TABLES", "20"))
if c:
    ctx.srs_precompute(sid, c)
ptrs, ms = [sc.data_ptr()] * nb, [m - (k % 3) for k in range(nb)]
ctx.msm_batch_devptr(sid, ptrs, ms)
reps = int(os.environ.get("REPS", "5"))
tot = 0.0
for _ in range(reps):
    torch.cuda.synchronize()
    time.sleep(0.01)
    t0 = time.perf_counter()
    ctx.msm_batch_devptr(sid, ptrs, ms)
    tot += time.perf_counter() - t0
print(f"BATCH {nb} x 2^{log_m}: {tot / reps / nb * 1e3:.3f} ms per MSM", flush=True)
