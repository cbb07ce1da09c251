"""typlonk_prove_host (columns in host memory, uploaded beside round 1) against upload-then-typlonk_prove: wall time per proof as
a host caller (the Rust layer's Backend::prove) sees it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, typlonk_amd
from typlonk_amd.circuits import SquaringChain
from bench import fr_mont_limbs

log_n = int(os.environ.get("LOG_N", "20"))
n = 1 << log_n
ctx = typlonk_amd.Context(0)
sid = ctx.srs_generate(fr_mont_limbs(2), n + 3)
ctx.srs_precompute(sid, 20 if log_n >= 19 else 0)
chain = SquaringChain(ctx, log_n)
host_w = [b.download() for b in chain.wire_evals]
host_pi = np.zeros((n, 4), dtype=np.uint64)
bufs = [ctx.alloc(n) for _ in range(4)]

def via_upload():
    for b, h in zip(bufs[:3], host_w):
        b.upload(h)
    bufs[3].upload(host_pi)
    return ctx.prove_native(sid, chain.circuit, bufs[:3], bufs[3], chain.cosets)

def via_host():
    return ctx.prove_native_host(sid, chain.circuit, host_w, host_pi, chain.cosets)

for name, fn in (("upload + typlonk_prove", via_upload), ("typlonk_prove_host", via_host)) * 2:
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    reps = 6
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    print(f"PROVEHOST 2^{log_n} {name}: {(time.perf_counter() - t0) / reps * 1e3:.2f} ms per proof", flush=True)
