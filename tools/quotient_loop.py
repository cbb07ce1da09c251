"""REPS calls of the quotient (typlonk_quotient_dev with a loaded circuit, public-input polynomial present) on random
inputs -- mean wall time per call, and under rocprofv3 --kernel-trace --stats the per-kernel split (extension NTT passes,
pointwise kernel, inverse coset NTT)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch, typlonk_amd
from bench import fr_mont_limbs, synthetic_scalars

log_n = int(os.environ.get("LOG_N", "20"))
reps = int(os.environ.get("REPS", "20"))
n = 1 << log_n
dev = torch.device("cuda", 0)
ctx = typlonk_amd.Context(0)
bufs = []
for k in range(13):
    t = synthetic_scalars(n, 0xA000 + k, dev)
    b = ctx.alloc(n)
    b.upload(t.cpu().numpy().view(np.uint64).reshape(n, 4))
    bufs.append(b)
wires, zbuf, pibuf, sels, sigs = bufs[0:3], bufs[3], bufs[4], bufs[5:10], bufs[10:13]
cosets = [fr_mont_limbs(1), fr_mont_limbs(7), fr_mont_limbs(13)]
chal = [fr_mont_limbs(0x1234567 + k) for k in range(3)]
cid = ctx.circuit_load(log_n, sels, sigs)
t_out = ctx.alloc(4 * n)
call = lambda: ctx.quotient_dev(log_n, wires, zbuf, None, None, pibuf, chal[0], chal[1], chal[2], cosets, t_out, circuit=cid)
call()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    call()
torch.cuda.synchronize()
print(f"QUOTIENT log_n={log_n}: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per call", flush=True)
