"""Time of a 2^LOG_M-term table-mode MSM on scalar sets that pile entries on a few buckets (equal scalars, tiny scalars,
one distinct value per 2^k terms) next to uniform ones -- the heavy-bucket path of msm_sort / msm_accum."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, typlonk_amd
from bench import synthetic_scalars, fr_mont_limbs

log_m = int(os.environ.get("LOG_M", "20"))
m = 1 << log_m
ctx = typlonk_amd.Context(0)
sid = ctx.srs_generate(fr_mont_limbs(2), m + 3)
ctx.srs_precompute(sid, int(os.environ.get("TABLES", "0")))
dev = torch.device("cuda", 0)
uni = synthetic_scalars(m, 1, dev)
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
def const(v):
    return torch.from_numpy(np.tile(fr_mont_limbs(v).view(np.int64), (m, 1))).to(dev)
sets = {"uniform": uni, "all 1": const(1), "all r-1": const(R - 1), "all 0x1234": const(0x1234)}
short = uni.clone(); short[:, 1:] = 0        # Montgomery limbs with three zero words: still < r, a fixed odd spread
sets["64-bit residues"] = short
rep = uni.clone(); rep[:] = uni[torch.arange(m, device=dev) >> 10 << 10]     # one distinct scalar per 1024 terms
sets["1024 copies each"] = rep
only = os.environ.get("ONLY")
for name, sc in sets.items():
    if only and only != name:
        continue
    for _ in range(2):
        ctx.msm_devptr(sid, sc.data_ptr(), m)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.msm_devptr(sid, sc.data_ptr(), m)
    print(f"ADV 2^{log_m} {name:18s} {(time.perf_counter() - t0) / 5 * 1e3:8.3f} ms", flush=True)
