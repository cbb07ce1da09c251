"""Window sweep of the fair-CPU bucket MSM (oracle_msm_pippenger) on this box's host cores: which c bench.py should use."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import coracle as CO
log_m = int(os.environ.get("LOG_M", "20"))
m = 1 << log_m
rng = np.random.default_rng(1)
sc = rng.integers(0, 1 << 62, size=(m, 4), dtype=np.uint64)
xy, inf = CO.srs_pow2_secret(1, 4096)          # a few distinct points, tiled: the timing does not depend on the values
xy = np.tile(xy, (m // 4096, 1)); inf = np.tile(inf, m // 4096)
CO.msm_pippenger(sc[:1024], xy[:1024], inf[:1024], c=8)
for c in [int(x) for x in os.environ.get("CS", "9,10,11,12,13,14,15,16").split(",")]:
    t = time.perf_counter(); _, _, ops, thr = CO.msm_pippenger(sc, xy, inf, c=c); dt = time.perf_counter() - t
    print(f"log_m {log_m} c {c}: {dt*1e3:8.1f} ms  {m/dt/1e6:6.3f} M terms/s  threads {thr}", flush=True)
