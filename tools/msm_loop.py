"""A fixed number of table-mode MSMs (for rocprofv3 kernel traces)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, typlonk_amd
from bench import synthetic_scalars, fr_mont_limbs

log_m = int(os.environ.get("LOG_M", "20"))
m = 1 << log_m
ctx = typlonk_amd.Context(0)
sc = synthetic_scalars(m, 1, torch.device("cuda", 0))
sid = ctx.srs_generate(fr_mont_limbs(2), m + 3)
c = int(os.environ.get("TABLES", "20"))
if c:
    ctx.srs_precompute(sid, c)
for _ in range(int(os.environ.get("REPS", "20"))):
    ctx.msm_devptr(sid, sc.data_ptr(), m)
